#!/bin/bash
# per-kernel average times of a few fwd+bwd passes of the bench workload (rocprofv3 --kernel-trace --stats). bash tools/kstats.sh [tag]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=${1:-k}
rm -rf gpurun_out/kstats_$TAG
ITERS=${ITERS:-6} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$TAG -o k -- python3 tools/prof_raster.py > gpurun_out/kstats_$TAG.log 2>&1
python3 - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/kstats_{tag}/k_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{float(r['AverageNs'])/1e6:8.4f} ms x{r['Calls']:>4}  {r['Name'][:90]}")
PY
