#!/bin/bash
# Build variants of backward_preprocess.hip (C3DGS_BWDPRE_FLAGS): backward_preprocess stage time on the 3M bench scene and on the heavy-tailed one.
# VARIANTS="-DC3DGS_BWD_CH=256 -DC3DGS_BWD_LONG_RUN=8 ..." bash tools/ablate_bwdpre.sh   (a variant may hold several flags joined by commas)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${VARIANTS}; do
  touch c3dgs_amd/csrc/backward_preprocess.hip
  if ! C3DGS_BWDPRE_FLAGS="${v//,/ }" python -m c3dgs_amd.build > gpurun_out/ablate_build.log 2>&1; then echo "[$v] build failed"; tail -3 gpurun_out/ablate_build.log; continue; fi
  for i in 1 2; do P=3000000 python tools/stage_times.py "[$v]" 2>/dev/null | grep -o "^\[.*\]\|'backward_preprocess': [0-9.]*" | tr '\n' ' '; done
  STEPS=4 python tools/prof_heavy.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' heavy:', d['stages_ms']['backward_preprocess'])"
done
touch c3dgs_amd/csrc/backward_preprocess.hip; python -m c3dgs_amd.build > /dev/null
