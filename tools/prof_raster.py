"""Small driver for rocprofv3 runs: a few fwd+bwd passes of the bench workload, nothing else."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import bench, c3dgs_amd
from c3dgs_amd import rasterizer as rz
dev = torch.device("cuda", 0)
P, W, H = int(os.environ.get("P", 3_000_000)), 1920, 1080
N = int(os.environ.get("ITERS", 3))
intr, ev, t, dL, ix = bench.build_workload(P, W, H, 1200.0, dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
E = torch.Tensor([])
bg = torch.zeros(3, device=dev)
if os.environ.get("NONINDEXED"):      # the reference's plain path: per-Gaussian SH / scale / rotation rows, [P, M, 3] gradient
    from tests import synth
    sc = {k: v.to(dev) for k, v in synth.scene(P, W, H, 1200.0).items()}
    for _ in range(N):
        o = rz._C.rasterize_gaussians(bg, sc["means3D"], E, sc["opacities"], sc["scales"], sc["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, sc["shs"], 3, campos, False, False, True)
        g = rz._C.rasterize_gaussians_backward(bg, sc["means3D"], o[2], E, sc["scales"], sc["rotations"], 1.0, E, view, proj, tfx, tfy, dL, sc["shs"], 3, campos, o[3], o[0], o[4], o[5], False)
    torch.cuda.synchronize()
    print("done", o[0])
    sys.exit(0)
for _ in range(N):
    o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
    g = rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
torch.cuda.synchronize()
print("done", o[0])
