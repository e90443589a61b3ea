#!/bin/bash
# Timing-only build variants of render.hip (C3DGS_RENDER_FLAGS); prints stage times per variant.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in ${VARIANTS}; do
  touch c3dgs_amd/csrc/render.hip
  C3DGS_RENDER_FLAGS="-fno-slp-vectorize ${v//,/ }" python -m c3dgs_amd.build > /dev/null
  python tools/stage_times.py "[$v]" 2>/dev/null | grep -o "^\[.*\]\|'render_forward': [0-9.]*\|'render_backward': [0-9.]*" | tr '\n' ' '; echo
done
touch c3dgs_amd/csrc/render.hip; python -m c3dgs_amd.build > /dev/null
