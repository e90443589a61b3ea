#!/bin/bash
# Round evidence in one gpurun call: the PMC traffic passes first (bench.py reads profiles/traffic_latest.json for
# roofline.traffic, so it must come from the SAME build), then the default bench line, then the same command under
# rocprofv3 --kernel-trace --stats. Usage (on the GPU box): bash tools/evidence.sh <tag>; copy gpurun_out/*<tag>* into profiles/.
set -e
TAG=${1:-r01x}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out
bash tools/pmc_traffic.sh > gpurun_out/pmc_${TAG}.log 2>&1
cp gpurun_out/pmc_traffic/traffic.json gpurun_out/traffic_${TAG}.json
cp gpurun_out/pmc_traffic/traffic.json profiles/traffic_latest.json
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}_prof.json 2> gpurun_out/bench_${TAG}_prof.err
tail -c 600 gpurun_out/bench_${TAG}.json
