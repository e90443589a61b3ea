#!/bin/bash
# Build variants of the u32 (depth) sort tile shape (C3DGS_SORT_FLAGS); per variant: sort tests, then the depth_sort stage time.
# VARIANTS="-DC3DGS_OS_TILE32=4096,-DC3DGS_OS_BLOCK32=256 ..." bash tools/ablate_sort.sh
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${VARIANTS}; do
  touch c3dgs_amd/csrc/radix_sort.hip
  C3DGS_SORT_FLAGS="${v//,/ }" python -m c3dgs_amd.build > /dev/null
  python -m pytest tests/test_sort_gpu.py -m gpu -q -x 2>&1 | tail -1
  for P in 1000000 3000000 6000000; do
    P=$P python tools/stage_times.py "[$v P=$P]" 2>/dev/null | grep -o "^\[.*\]\|'depth_sort': [0-9.]*" | tr '\n' ' '; echo
  done
done
touch c3dgs_amd/csrc/radix_sort.hip; python -m c3dgs_amd.build > /dev/null
