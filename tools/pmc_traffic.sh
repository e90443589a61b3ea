#!/bin/bash
# HBM traffic of the raster kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only (no other trace domains).
# Run on the GPU box through gpurun:  bash tools/pmc_traffic.sh
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_traffic
rm -rf "$OUT"; mkdir -p "$OUT"
ITERS=3 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 tools/prof_raster.py > "$OUT/fetch.log" 2>&1
ITERS=3 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 tools/prof_raster.py > "$OUT/write.log" 2>&1
python3 tools/pmc_summarize.py "$OUT/fetch/f_counter_collection.csv" "$OUT/write/w_counter_collection.csv" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
