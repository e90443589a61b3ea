#!/bin/bash
# Timing-only ablations of backward_preprocess (results are wrong by construction): which part costs what.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${ABLATIONS:-"" "-DC3DGS_ABLATE_STAGE=1" "-DC3DGS_ABLATE_SH=1" "-DC3DGS_ABLATE_GS=1" "-DC3DGS_ABLATE_SHMATH=1,-DC3DGS_ABLATE_SH=1" "-DC3DGS_ABLATE_STAGE=1,-DC3DGS_ABLATE_SHMATH=1,-DC3DGS_ABLATE_SH=1,-DC3DGS_ABLATE_GS=1"}; do
  touch c3dgs_amd/csrc/backward_preprocess.hip
  C3DGS_BWDPRE_FLAGS="${v//,/ }" python -m c3dgs_amd.build > /dev/null
  python tools/stage_times.py "[$v]" 2>/dev/null | grep -o "^\[.*\]\|'backward_preprocess': [0-9.]*" | tr '\n' ' '; echo
done
touch c3dgs_amd/csrc/backward_preprocess.hip; python -m c3dgs_amd.build > /dev/null
