#!/bin/bash
# Per-kernel average durations of a few fwd+bwd passes (rocprofv3 --kernel-trace --stats). bash tools/kstats.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf gpurun_out/kstats; mkdir -p gpurun_out/kstats
ITERS=8 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -o k -- python3 tools/prof_raster.py > gpurun_out/kstats/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kstats/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4}  {r['Name'][:90]}")
PY
