"""Times the fused L1+SSIM loss (fwd+bwd, 1080p) against the same math written with torch ops (five grouped 11x11
conv2d + elementwise, as utils/loss_utils.py does) on the same GPU."""
import os, sys, time, math
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from c3dgs_amd import loss as L, _lib
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
gt = torch.rand(3, 1080, 1920, generator=g).to(dev)
img0 = (gt + 0.1 * torch.randn(3, 1080, 1920, generator=g).to(dev)).clamp(0, 1)

def torch_loss(img, gt, lam=0.2):
    g1 = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], device=dev)
    g1 = g1 / g1.sum()
    w = (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous()
    mu1, mu2 = F.conv2d(img, w, padding=5, groups=3), F.conv2d(gt, w, padding=5, groups=3)
    s1 = F.conv2d(img * img, w, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(gt * gt, w, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(img * gt, w, padding=5, groups=3) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - lam) * (img - gt).abs().mean() + lam * (1 - m.mean())

def run(fn, n=20):
    for _ in range(3):
        x = img0.clone().requires_grad_(); fn(x, gt).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x = img0.clone().requires_grad_(); v = fn(x, gt); v.backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(v), x.grad

t_f, v_f, g_f = run(lambda a, b: L.l1_ssim_loss(a, b, 0.2))
t_t, v_t, g_t = run(torch_loss)
_lib.profile_enable(True); _lib.profile_read()
run(lambda a, b: L.l1_ssim_loss(a, b, 0.2), 10)
st = _lib.profile_read()
print(f"fused: {t_f:.3f} ms/iter (value {v_f:.7f})   torch ops: {t_t:.3f} ms/iter (value {v_t:.7f})   grad rel diff {(g_f-g_t).abs().max().item()/g_t.abs().max().item():.2e}")
print({k: round(v[0] / v[1], 4) for k, v in st.items()})
