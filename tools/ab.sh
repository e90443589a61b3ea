#!/bin/bash
# A/B of library builds on ONE box: bash tools/ab.sh libA.so libB.so ...  (per-kernel averages, two rounds each, interleaved)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for round in 1 2 3; do
  for lib in "$@"; do
    tag=$(basename $lib .so)_$round
    C3DGS_LIB_PATH=$R/$lib bash tools/kstats.sh $tag > gpurun_out/ab_$tag.txt 2>&1
    echo "== $lib round $round"
    grep -E "render_backward|render_forward|preprocess|sum_partials|os_pass|duplicate" gpurun_out/ab_$tag.txt
  done
done
