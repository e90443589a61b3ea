#!/bin/bash
# Per-kernel durations of a few nearest-codeword searches of the bench's colour shape (rocprofv3 --kernel-trace --stats of
# tools/prof_vq.py). bash tools/kstats_prof_vq.sh [tag]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:-x}
rm -rf gpurun_out/kstats_pvq_$TAG; mkdir -p gpurun_out/kstats_pvq_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_pvq_$TAG -o k -- python3 tools/prof_vq.py > gpurun_out/kstats_pvq_$TAG/log.txt 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/kstats_pvq_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:9.1f}  {r['Name'][:90]}")
PY
