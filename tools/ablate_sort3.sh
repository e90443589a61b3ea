#!/bin/bash
# Tile shape of the u32 (depth) sort with its second payload, as build variants (C3DGS_SORT_FLAGS): sort tests + depth_sort stage time at 3M.
# VARIANTS="-DC3DGS_OS_TILE32=11264,-DC3DGS_OS_BLOCK32=1024 ..." bash tools/ablate_sort3.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${VARIANTS}; do
  touch c3dgs_amd/csrc/radix_sort.hip
  if ! C3DGS_SORT_FLAGS="${v//,/ }" python -m c3dgs_amd.build > gpurun_out/ablate_build.log 2>&1; then echo "[$v] build failed"; tail -3 gpurun_out/ablate_build.log; continue; fi
  t=$(python -m pytest tests/test_sort_gpu.py -m gpu -q -x -k "not timeout and not rocprim" 2>&1 | tail -1)
  for i in 1 2; do P=3000000 python tools/stage_times.py "[$v]" 2>/dev/null | grep -o "^\[.*\]\|'depth_sort': [0-9.]*" | tr '\n' ' '; done; echo " $t"
done
touch c3dgs_amd/csrc/radix_sort.hip; python -m c3dgs_amd.build > /dev/null
