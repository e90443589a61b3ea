"""Host-side cost of one raster step (tiny scene, GPU time negligible): cProfile top entries. python tools/host_profile.py"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch, bench, c3dgs_amd
dev = torch.device("cuda", 0)
intr, ev, t, dL, ix = bench.build_workload(2000, 64, 64, 40.0, dev)
rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=intr, extrinsic_vector=ev.to(dev), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                             sh_degree=3, prefiltered=False, debug=False, clamp_color=True)
rast = c3dgs_amd.GaussianRasterizerIndexed(rs, optimize_camera=True)
leaves = {k: t[k].clone().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "scale_factors", "rotations")}
means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
evd = ev.to(dev)


def step():
    for v in leaves.values():
        v.grad = None
    rast.markVisible(leaves["means3D"], extrinsic_vector=evd)
    color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], sh_indices=t["sh_indices"],
                        g_indices=t["g_indices"], shs=leaves["shs"], scales=leaves["scales"], scale_factors=leaves["scale_factors"],
                        rotations=leaves["rotations"], extrinsic_vector=evd)
    torch.autograd.backward(color, dL)


for _ in range(50):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
