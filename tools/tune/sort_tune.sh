#!/bin/bash
# Builds and runs tools/tune/sort_tune.hip for a list of onesweep configurations (on the GPU box).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
out=gpurun_out/sort_tune.log; : > $out
build_run() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $1 tools/tune/sort_tune.hip -o /tmp/sort_tune_$$ 2>> gpurun_out/sort_tune.err && timeout -k 5 60 /tmp/sort_tune_$$ | sed "s|^|[$1] |" >> $out
  tail -1 $out
}
build_run ""
CONFIGS=${CONFIGS:-256,16,8,match 512,16,8,match 256,23,8,match 1024,8,8,match 256,16,7,match 512,16,7,match 256,16,8,basic_memoize 512,12,8,match}
for cfg in $CONFIGS; do
  IFS=, read bs ipt bits algo <<< "$cfg"
  build_run "-DCFG_BS=$bs -DCFG_IPT=$ipt -DCFG_BITS=$bits -DCFG_ALGO=$algo"
done
