#!/bin/bash
# HBM traffic of the raster kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only (no other trace domains).
# The profiled command IS bench.py (its headline step, extras off). Run on the GPU box through gpurun:  bash tools/pmc_traffic.sh
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/pmc_traffic
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 3 --warmup 1 --no-vq --no-cpu-baseline --no-extras > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 3 --warmup 1 --no-vq --no-cpu-baseline --no-extras > "$OUT/write.log" 2>&1
python3 tools/pmc_summarize.py "$OUT/fetch/f_counter_collection.csv" "$OUT/write/w_counter_collection.csv" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
