"""Driver for a rocprofv3 kernel trace of the Lloyd loop: STEPS steps of vq_features on config 4's colour shape (one GPU), or
on a rank's slice of it (SLICE=32768: what one of 8 ranks runs, minus the all-reduce).  See tools/vq_step_trace.sh."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from c3dgs_amd import vq as vqm

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
N, D, K, B = 5_400_000, 48, 4096, 2 ** 18
feats = torch.randn(N, D, device=dev, generator=g) * 0.1
imp = torch.rand(N, device=dev, generator=g).pow(4)
steps, n_slice = int(os.environ.get("STEPS", 40)), int(os.environ.get("SLICE", B))
torch.manual_seed(11)
ms = bench.vq_slice_step(vqm, feats, imp, K, B, n_slice, steps, dev)
print(f"steps={steps} (+5 warm-up steps) slice={n_slice} ms_per_step={ms:.4f}")
