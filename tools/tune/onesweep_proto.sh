#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out; : > gpurun_out/onesweep.log
for cfg in ${CONFIGS:-512,8 256,16 512,16 1024,8}; do
  IFS=, read bs ipt <<< "$cfg"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOS_BLOCK=$bs -DOS_IPT=$ipt tools/tune/onesweep_proto.hip -o /tmp/os_$$ 2> gpurun_out/onesweep.err && timeout -k 5 120 /tmp/os_$$ >> gpurun_out/onesweep.log 2>&1
  echo "exit $?" >> gpurun_out/onesweep.log
done
cat gpurun_out/onesweep.log
