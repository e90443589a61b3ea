"""Workload statistics of the bench scene (how much of the sorted lists the blend kernels really visit)."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, c3dgs_amd
from c3dgs_amd import rasterizer as rz, _lib
from tests import gpu_util
dev = torch.device("cuda", 0)
P, W, H = int(os.environ.get("P", 3_000_000)), 1920, 1080
intr, ev, t, dL, ix = bench.build_workload(P, W, H, 1200.0, dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
E = torch.Tensor([])
o = rz._C.rasterize_gaussians_indexed(torch.zeros(3, device=dev), t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
fw = dict(num_rendered=o[0], color=o[1], radii=o[2], geom=o[3], binning=o[4], img=o[5], W=W, H=H)
u = gpu_util.unpack(fw)
R = o[0]; T = u["ranges"].shape[0]
il = _lib.ImageLayout(); _lib.lib().c3dgs_get_image_layout(W, H, C.byref(il))
tile_used = o[5][il.tile_used: il.tile_used + 4 * T].view(torch.int32).cpu().numpy()
n = (u["ranges"][:, 1] - u["ranges"][:, 0]).astype(np.int64)
nc = u["n_contrib"].astype(np.int64)
print("P", P, "V", int((u["radii"] > 0).sum()), "R", R, "R/P", R / P)
print("per-tile list length: mean", n.mean(), "max", n.max(), "p50", np.median(n))
print("tile_used: sum", tile_used.sum(), "frac of R", tile_used.sum() / R, "mean", tile_used.mean())
print("n_contrib per pixel: mean", nc.mean(), "p50", np.median(nc), "max", nc.max())
print("final_T: mean", u["final_T"].mean(), "frac saturated(<1e-3)", (u["final_T"] < 1e-3).mean())
rad = u["radii"][u["radii"] > 0]
print("radius px: mean", rad.mean(), "p50", np.median(rad), "p90", np.percentile(rad, 90), "max", rad.max())
tt = u["tiles_touched"][u["tiles_touched"] > 0]
print("tiles_touched: mean", tt.mean(), "p50", np.median(tt), "p99", np.percentile(tt, 99), "max", tt.max())
op = u["conic_opacity"][u["radii"] > 0, 3]
print("opacity mean", op.mean(), "p50", np.median(op))
