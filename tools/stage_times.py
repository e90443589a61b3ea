"""Mean stage times (HIP events) of the bench workload through the C ABI. python tools/stage_times.py [tag]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from c3dgs_amd import rasterizer as rz, _lib
dev = torch.device("cuda", 0)
P, W, H = int(os.environ.get("P", 3_000_000)), 1920, 1080
intr, ev, t, dL, ix = bench.build_workload(P, W, H, 1200.0, dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
E = torch.Tensor([])
bg = torch.zeros(3, device=dev)


def one():
    o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
    rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])


for _ in range(6):
    one()
torch.cuda.synchronize()
_lib.profile_enable(True)
_lib.profile_read()
for _ in range(10):
    one()
torch.cuda.synchronize()
st = _lib.profile_read()
print(sys.argv[1] if len(sys.argv) > 1 else "", {k: round(v[0] / v[1], 4) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])})
