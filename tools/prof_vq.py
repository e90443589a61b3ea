"""Small driver for rocprofv3 runs: a few nearest-codeword searches of the bench's colour shape (2^18 x 4096 x 48)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import c3dgs_amd
g = torch.Generator(device="cuda").manual_seed(1)
K, D, N = 4096, 48, 2 ** 18
cb = torch.randn(K, D, device="cuda", generator=g) * 0.1
x = torch.randn(N, D, device="cuda", generator=g) * 0.1
for _ in range(int(os.environ.get("ITERS", 5))):
    c3dgs_amd.weightedDistance(x, cb)
torch.cuda.synchronize()
print("done")
