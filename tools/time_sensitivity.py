"""Per-camera cost of the sensitivity pass (row N2) on the config-5 scene, with a torch-profiler kernel table.
python tools/time_sensitivity.py [gaussians] [cameras]"""
import math, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import model as gm, sensitivity, _lib
from tests import synth

dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, focal = 1920, 1080, 1200.0


class Camera:
    def __init__(self, yaw):
        intr, _ = synth.camera(W, H, focal)
        h = 0.5 * yaw
        self.intrinsic = intr.to(dev)
        self.extrinsic_vector = torch.tensor([0.0, math.sin(h), 0.0, math.cos(h), 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        self.original_image = None


sc = synth.scene(P, W, H, focal, seed=1234, sh_degree=3)
op = sc["opacities"].clamp(1e-6, 1 - 1e-6)
g = gm.GaussianModel(3, quantization=True, device=dev)
g.set_tensors(xyz=sc["means3D"], features_dc=sc["shs"][:, :1], features_rest=sc["shs"][:, 1:],
              scaling=sc["scales"] / sc["scales"].norm(dim=1, keepdim=True), rotation=sc["rotations"],
              opacity=torch.log(op / (1 - op)), scaling_factor=torch.log(sc["scales"].norm(dim=1, keepdim=True)))
del sc
pipe, bg = gm.PipelineParams(), torch.zeros(3, device=dev)
cams = [Camera(y) for y in torch.linspace(-0.25, 0.25, NC).tolist()]
with torch.no_grad():
    for c in cams:
        c.original_image = g.render(c, pipe, bg)["render"].detach().clone()
for use_gt in (True, False):
    sensitivity.calc_importance_experimental(g, cams[:2], pipe, use_gt=use_gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sensitivity.calc_importance_experimental(g, cams, pipe, use_gt=use_gt)
    torch.cuda.synchronize()
    print(f"use_gt={use_gt}: {(time.perf_counter() - t0) / NC * 1e3:.2f} ms per camera", flush=True)
_lib.profile_enable(True); _lib.profile_read()
sensitivity.calc_importance_experimental(g, cams, pipe, use_gt=True)
torch.cuda.synchronize()
st = _lib.profile_read(); _lib.profile_enable(False)
print({k: round(v[0] / NC, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}, "ms per camera in library stages; sum",
      round(sum(v[0] for v in st.values()) / NC, 3), flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    sensitivity.calc_importance_experimental(g, cams[:4], pipe, use_gt=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
