#!/usr/bin/env python3
"""Assignment-kernel time for the per-rank slices of a sharded colour batch (2^18 points over 1/2/4/8 ranks; K=4096, D=48).
C3DGS_VQ_SPLIT=0 / 1 forces the plain / codebook-split kernel (default: split below 512 four-wave workgroups)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import c3dgs_amd
from c3dgs_amd import _lib

g = torch.Generator(device="cuda").manual_seed(1)
K, D = 4096, 48
cb = torch.randn(K, D, device="cuda", generator=g) * 0.1
for ranks in (1, 2, 4, 8):
    N = 2 ** 18 // ranks
    x = torch.randn(N, D, device="cuda", generator=g) * 0.1
    for _ in range(3):
        c3dgs_amd.weightedDistance(x, cb)
    _lib.profile_enable(True, only="weighted_distance")
    _lib.profile_read()
    for _ in range(20):
        c3dgs_amd.weightedDistance(x, cb)
    torch.cuda.synchronize()
    ms, n = _lib.profile_read()["weighted_distance"]
    _lib.profile_enable(False)
    tf = 2.0 * N * K * D / (ms / n * 1e-3) / 1e12
    print(f"split_env={os.environ.get('C3DGS_VQ_SPLIT', 'auto')} ranks={ranks} N={N}: {ms / n * 1e3:.1f} us  {tf:.1f} TFLOP/s", flush=True)
