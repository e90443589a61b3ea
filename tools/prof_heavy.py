"""The heavy-tailed robustness scene of bench.py (bench_heavy_tail) alone, for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/heavy -o h -- python3 tools/prof_heavy.py"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench, c3dgs_amd
from c3dgs_amd import _lib
dev = torch.device("cuda", 0)
res = bench.bench_heavy_tail(c3dgs_amd, _lib, dev, int(os.environ.get("STEPS", 4)))
print(json.dumps(res))
