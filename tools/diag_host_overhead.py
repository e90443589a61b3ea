import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, cProfile, pstats
import bench
import c3dgs_amd
from c3dgs_amd import _lib, rasterizer as rz
dev = torch.device("cuda",0)
P,W,H=3_000_000,1920,1080
intr, ev, t, dL, ix = bench.build_workload(P,W,H,1200.0,dev)
rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=intr, extrinsic_vector=ev.to(dev), bg=torch.zeros(3, device=dev), scale_modifier=1.0, sh_degree=3, prefiltered=False, debug=False, clamp_color=True)
evd = ev.to(dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, evd, dev)
E = torch.Tensor([])
def fwd():
    return rz._C.rasterize_gaussians_indexed(rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
def bwd(o):
    return rz._C.rasterize_gaussians_backward_indexed(rs.bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
for _ in range(3):
    o = fwd(); g = bwd(o)
torch.cuda.synchronize()
def T(f, n=10):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3, r
tf, o = T(fwd); print("fwd ms", tf)
tb, g = T(lambda: bwd(o)); print("bwd ms", tb)
tc, _ = T(lambda: rz.camera_matrices(intr, evd, dev)); print("camera_matrices ms", tc)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    o = fwd(); g = bwd(o)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
