"""Where one WHOLE vq_features call of the config-4 colour shape spends its wall time (set-up, 100 Lloyd steps, final
assignment): time.perf_counter marks around the phases + cProfile of the second call. python tools/host_profile_vq_call.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import vq as vqm
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
N, D, K, B, steps = 5_400_000, 48, 4096, 2 ** 18, 100
f = torch.randn(N, D, device=dev, generator=g) * 0.1
imp = torch.rand(N, device=dev, generator=g).pow(4)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = {}
    vqm.vq_features(f, imp, K, B, steps, silent=True, stats=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"call {rep}: {dt*1e3:.1f} ms total; lloyd {st['lloyd_seconds']*1e3:.1f} ms; final assignment {st['final_assignment_seconds']*1e3:.1f} ms", flush=True)
pr = cProfile.Profile(); pr.enable()
vqm.vq_features(f, imp, K, B, steps, silent=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
