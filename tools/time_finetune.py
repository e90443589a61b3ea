"""Host- vs GPU-bound check of the fine-tuning loop on a small indexed model: the plain loop, pipeline.finetune, and the
GPU time of one iteration (library stage sum). python tools/time_finetune.py [gaussians] [iterations]"""
import math, os, random, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import model as gm, loss as lossm, optim, pipeline, _lib
from tests import synth

dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 340_000
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 300
W, H, focal = 1920, 1080, 1200.0
sc = synth.scene(P, W, H, focal, seed=1234, sh_degree=3, scale_median=0.03)
raw = synth.raw_params(synth.index_scene(sc))


def fresh():
    g = gm.GaussianModel(3, quantization=True, device=dev)
    g.set_tensors(**raw)
    g.spatial_lr_scale = 1.0
    return g


class Cam:
    def __init__(self, yaw):
        intr, _ = synth.camera(W, H, focal)
        self.intrinsic = intr.to(dev)
        h = 0.5 * yaw
        self.extrinsic_vector = torch.tensor([0.0, math.sin(h), 0.0, math.cos(h), 0.0, 0.0, 0.0], device=dev)


cams = [Cam(y) for y in torch.linspace(-0.2, 0.2, 8).tolist()]
pipe, bg = gm.PipelineParams(), torch.zeros(3, device=dev)
g0 = fresh()
with torch.no_grad():
    for c in cams:
        c.original_image = g0.render(c, pipe, bg)["render"].clone()


def plain(g, n):
    g.training_setup(pipeline.OptimizationParams())
    for it in range(n):
        cam = cams[it % len(cams)]
        lossm.l1_ssim_loss(g.render(cam, pipe, bg)["render"], cam.original_image, 0.2).backward()
        g.optimizer.step()
        g.optimizer.zero_grad(set_to_none=True)


class Scene:
    loaded_iter = 0

    def getTrainCameras(self):
        return cams


if os.environ.get("COLD"):          # what do the first iterations of a cold process cost?
    g = fresh()
    Scene.gaussians = g
    marks = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipeline.finetune(Scene(), pipeline._Dataset(), pipeline.OptimizationParams(), pipeline.CompressionParams(finetune_iterations=100), pipe,
                      log=lambda it, v: marks.append((it, round(time.perf_counter() - t0, 4))))
    torch.cuda.synchronize()
    print("cold pipeline.finetune: seconds at iteration", marks, flush=True)
for name in ("plain", "pipeline", "plain", "pipeline"):
    g = fresh()
    Scene.gaussians = g
    plain(g, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if name == "plain":
        plain(g, IT)
    else:
        pipeline.finetune(Scene(), pipeline._Dataset(), pipeline.OptimizationParams(), pipeline.CompressionParams(finetune_iterations=IT), pipe)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) / IT * 1e3:.3f} ms per iteration", flush=True)
g = fresh()
plain(g, 10)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_read()
plain(g, 20)
torch.cuda.synchronize()
st = _lib.profile_read(); _lib.profile_enable(False)
print("library stages per iteration: %.3f ms" % (sum(v[0] for v in st.values()) / 20),
      {k: round(v[0] / 20, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])[:8]})
