import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import vq as vqm, _lib
dev = torch.device("cuda", 0)
def run(N, D, K, B, steps, sn):
    g = torch.Generator(device=dev).manual_seed(3)
    f = torch.randn(N, D, device=dev, generator=g) * 0.1
    if sn: f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, device=dev, generator=g).pow(4)
    vqm.vq_features(f, imp, K, B, 3, scale_normalize=sn, silent=True)
    torch.cuda.synchronize(); _lib.profile_enable(True); _lib.profile_read()
    t0 = time.perf_counter()
    vqm.vq_features(f, imp, K, B, steps, scale_normalize=sn, silent=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = _lib.profile_read(); _lib.profile_enable(False)
    print(f"N={N} D={D} K={K} B={B} steps={steps}: {dt:.3f} s ({1e3*dt/steps:.3f} ms/step)", {k: round(v[0]/v[1], 4) for k, v in st.items()})
run(5_400_000, 48, 4096, 2**18, 100, False)
run(4_500_000, 6, 2048, 2**20, 200, True)
run(10_000, 12, 256, 2**14, 100, False)
