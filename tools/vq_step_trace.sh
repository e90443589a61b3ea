#!/bin/bash
# Kernel launches per Lloyd step: rocprofv3 --kernel-trace --stats of tools/vq_step_trace.py, calls divided by the step count.
#   bash tools/vq_step_trace.sh [out.txt]          (SLICE=32768 for a rank's slice of a sharded batch)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=${1:-gpurun_out/vq_step_trace.txt}
: > "$OUT"
for SL in 262144 32768; do
  rm -rf gpurun_out/vqtrace; mkdir -p gpurun_out/vqtrace
  SLICE=$SL STEPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vqtrace -o k -- python3 tools/vq_step_trace.py > gpurun_out/vqtrace/log.txt 2>&1
  SL=$SL python3 - >> "$OUT" <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/vqtrace/**/*kernel_stats.csv", recursive=True)[0]
log = [l for l in open("gpurun_out/vqtrace/log.txt").read().splitlines() if "ms_per_step" in l][-1]
steps = 45                                   # 40 timed + 5 warm-up
print(f"== slice of {os.environ['SL']} points per 2^18-point batch: {log}")
tot = 0.0
for r in csv.DictReader(open(f)):
    calls = int(r["Calls"])
    if calls < steps:                        # set-up kernels (uniform_init, generator, fills of the first step)
        continue
    per = calls / steps
    tot += per
    print(f"{per:6.2f} launches/step  {float(r['AverageNs'])/1e3:8.1f} us avg  {r['Name'][:100]}")
print(f"{tot:6.2f} kernel launches per Lloyd step in total (the raw draws are read from page-locked host memory by draws_to_indices_kernel: no copy)")
PY
done
cat "$OUT"
