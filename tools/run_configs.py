"""Measures the BASELINE.json configs on ONE GPU and prints markdown rows for BASELINE.md section 4.
(bench.py is the contract line; this is the wider table.)"""
import math, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import c3dgs_amd
from c3dgs_amd import rasterizer as rz, vq as vqm, _lib
from tests import synth

dev = torch.device("cuda", 0)
E = torch.Tensor([])


def timeit(fn, n, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def raster(P, indexed, backward, n=20):
    W, H, focal = 1920, 1080, 1200.0
    intr, ev = synth.camera(W, H, focal)
    sc = synth.scene(P, W, H, focal)
    src = synth.index_scene(sc) if indexed else sc
    t = {k: v.to(dev) for k, v in src.items()}
    view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
    bg = torch.zeros(3, device=dev)
    dL = synth.grad_image(W, H).to(dev)
    out = {}

    def fwd():
        if indexed:
            o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
        else:
            o = rz._C.rasterize_gaussians(bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, False, False, True)
        out["o"] = o
        return o

    def both():
        o = fwd()
        if indexed:
            rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
        else:
            rz._C.rasterize_gaussians_backward(bg, t["means3D"], o[2], E, t["scales"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False)

    dt = timeit(both if backward else fwd, n)
    o = out["o"]
    V = int((o[2] > 0).sum().item())
    return dt, P, V, o[0]


def vq(N, D, K, B, steps, scale_normalize):
    g = torch.Generator(device=dev).manual_seed(3)
    f = torch.randn(N, D, device=dev, generator=g) * 0.1
    if scale_normalize:
        f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, device=dev, generator=g).pow(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb, idx = vqm.vq_features(f, imp, K, B, steps, scale_normalize=scale_normalize, silent=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


rows = []
# config 1
t1 = vq(10_000, 12, 256, 2 ** 14, 100, False)
rows.append(f"| 1 VQ plumbing: N=10k, D=12, K=256, batch 2^14, 100 steps | 1 | {t1:.3f} s | — | — | — | — | oracle-equal (tests/test_vq_gpu.py) |")
for name, P, indexed, bwd in [("2 raster fwd 1M (non-indexed)", 1_000_000, False, False), ("2 raster fwd 1M (indexed)", 1_000_000, True, False),
                              ("3 raster fwd+bwd 3M (indexed, QAT)", 3_000_000, True, True), ("3' raster fwd+bwd 3M (non-indexed)", 3_000_000, False, True),
                              ("raster fwd+bwd 1M (indexed)", 1_000_000, True, True), ("raster fwd+bwd 6M (indexed)", 6_000_000, True, True)]:
    dt, P_, V, R = raster(P, indexed, bwd)
    rows.append(f"| {name} | 1 | {1/dt:.1f} views/s ({dt*1e3:.2f} ms) | {P_} / {V} / {R} | — | — | — | bit-exact keys, PSNR>=80 dB, grads<=1e-4 (tests) |")
    torch.cuda.empty_cache()
tc = vq(5_400_000, 48, 4096, 2 ** 18, 100, False)
tg = vq(4_500_000, 6, 2048, 2 ** 20, 800, True)
rows.append(f"| 4 VQ 6M: colour N=5.4M D=48 K=4096 B=2^18 x100 | 1 | {tc:.2f} s | — | 2NKD/step = 103 GFLOP | see bench vq | — | oracle-equal |")
rows.append(f"| 4 VQ 6M: covariance N=4.5M D=6 K=2048 B=2^20 x800 (scale_normalize) | 1 | {tg:.2f} s | — | — | — | — | oracle-equal |")
print("\n".join(rows))
