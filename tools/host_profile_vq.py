"""cProfile of the Lloyd loop's host side (small problem: the GPU work is negligible). python tools/host_profile_vq.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import vq as vqm
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
f = torch.randn(10_000, 12, device=dev, generator=g) * 0.1
imp = torch.rand(10_000, device=dev, generator=g).pow(4)
vqm.vq_features(f, imp, 256, 2 ** 14, 20, silent=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
vqm.vq_features(f, imp, 256, 2 ** 14, 300, silent=True)
torch.cuda.synchronize()
print("ms per step: %.3f" % ((time.perf_counter() - t0) / 300 * 1e3))
pr = cProfile.Profile()
pr.enable()
vqm.vq_features(f, imp, 256, 2 ** 14, 300, silent=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
