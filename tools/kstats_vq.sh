#!/bin/bash
# Per-kernel average durations of the VQ Lloyd steps (rocprofv3 --kernel-trace --stats). bash tools/kstats_vq.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf gpurun_out/kstats_vq; mkdir -p gpurun_out/kstats_vq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_vq -o k -- python3 tools/time_vq.py > gpurun_out/kstats_vq/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kstats_vq/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:9.1f}  {r['Name'][:80]}")
PY
