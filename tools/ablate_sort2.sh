#!/bin/bash
# Tile shape of the depth-key onesweep sort as build variants (C3DGS_SORT_FLAGS); per variant: the sort tests, then the bench stage times.
# VARIANTS="-DC3DGS_OS_TILE32=16384,-DC3DGS_OS_BLOCK32=1024 -DC3DGS_OS_TILE32=8192 ..." bash tools/ablate_sort2.sh   (the u16 tile-key sort has a fixed 8192 x 1024 shape)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${VARIANTS}; do
  touch c3dgs_amd/csrc/radix_sort.hip
  if ! C3DGS_SORT_FLAGS="${v//,/ }" python -m c3dgs_amd.build > gpurun_out/ablate_build.log 2>&1; then echo "[$v] build failed"; tail -3 gpurun_out/ablate_build.log; continue; fi
  t=$(python -m pytest tests/test_sort_gpu.py -m gpu -q -x -k "not timeout and not rocprim" 2>&1 | tail -1)
  python bench.py --no-cpu-baseline --no-extras --no-vq --steps 10 > gpurun_out/ablate_sort.json 2> gpurun_out/ablate_sort.err
  echo "[$v] $t"; python tools/bench_summary.py gpurun_out/ablate_sort.json | grep -o "views/s [0-9.]* ms\| sort=[0-9.]*\|depth_sort=[0-9.]*" | tr '\n' ' '; echo
done
touch c3dgs_amd/csrc/radix_sort.hip; python -m c3dgs_amd.build > /dev/null
