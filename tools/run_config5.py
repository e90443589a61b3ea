"""BASELINE.json config 5 on ONE GPU: the composed compression pipeline of compress.py:202-300 on a synthetic scene --
sensitivity pass (row N2) -> sensitivity-weighted VQ of colour and covariance (V4, extract_rot_scale of N4) -> QAT
fine-tuning of the indexed model (rows N1 + raster + N3, Adam) -> Morton-sorted npz payload (N4) -> PSNR against the
uncompressed renders. Everything runs through c3dgs_amd's public API.

    python tools/run_config5.py [--gaussians 6000000] [--cameras 32] [--finetune 300] [--width 1920 --height 1080]

The reference runs 5000 fine-tuning iterations; the per-iteration time measured over --finetune iterations is
extrapolated to 5000 in the report (the fine-tuning itself is a fixed-cost loop of identical steps)."""
import argparse, json, math, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch
import c3dgs_amd
from c3dgs_amd import loss as lossm, model as gm, optim, sensitivity, vq as vqm
from tests import synth

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=6_000_000)
ap.add_argument("--cameras", type=int, default=32)
ap.add_argument("--finetune", type=int, default=300)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--color-steps", type=int, default=100)
ap.add_argument("--gaussian-steps", type=int, default=800)
ap.add_argument("--scene", default="opaque", choices=["opaque", "translucent"],
                help="opaque: synth-v1 as benched (opacity sigmoid(N(-1, 1.5^2))): the 32 cameras never blend 94 %% of it, the reference's "
                     "zero-importance prune (compression/vq.py:205-211) keeps 0.34M of 6M. translucent: opacity sigmoid(N(-3, 1)): "
                     "rays go deep, most Gaussians earn a gradient and survive, so the QAT stage runs at config-3 scale")
ap.add_argument("--oracle-psnr", action="store_true",
                help="also render camera 0 of the UNCOMPRESSED scene with the CPU oracle (oracle/, test infrastructure) and report the "
                     "PSNR of the HIP render of the uncompressed and of the compressed model against it")
ap.add_argument("--out", default="gpurun_out/config5.json")
args = ap.parse_args()
dev = torch.device("cuda", 0)
W, H, focal, P = args.width, args.height, 1200.0 * args.width / 1920.0, args.gaussians
t_all = time.time()
timings = {}


def sync():
    torch.cuda.synchronize()
    return time.time()


class Camera:
    """Yaw sweep: the camera stays at the world origin and turns about the y axis (extrinsic = quaternion + zero shift)."""

    def __init__(self, yaw):
        intr, _ = synth.camera(W, H, focal)
        h = 0.5 * yaw
        self.intrinsic = intr.to(dev)
        self.extrinsic_vector = torch.tensor([0.0, math.sin(h), 0.0, math.cos(h), 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        self.original_image = None


sc = synth.scene(P, W, H, focal, seed=1234, sh_degree=3)
if args.scene == "translucent":
    sc["opacities"] = torch.sigmoid(torch.randn(P, 1, generator=torch.Generator().manual_seed(77)) - 3.0).float()
oracle_img = None
if args.oracle_psnr:                                # camera 0 of the uncompressed scene through the CPU oracle (checker only)
    from oracle import oracle as orc
    intr0, _ = synth.camera(W, H, focal)
    h0 = 0.5 * float(torch.linspace(-0.25, 0.25, args.cameras)[0])
    ev0 = torch.tensor([0.0, math.sin(h0), 0.0, math.cos(h0), 0.0, 0.0, 0.0], dtype=torch.float32)
    t0 = time.time()
    st = orc.rasterize_forward(bg=torch.zeros(3).numpy(), means3D=sc["means3D"].numpy(), opacities=sc["opacities"].numpy(),
                               shs=sc["shs"].numpy(), scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy(), degree=3,
                               scale_modifier=1.0, prefiltered=False, clamp_color=True, **orc.camera(intr0.numpy(), ev0.numpy()))
    oracle_img = torch.from_numpy(st.out_color).to(dev)
    timings["oracle_render_camera0_s"] = time.time() - t0
    del st
op = sc["opacities"].clamp(1e-6, 1 - 1e-6)
gaussians = gm.GaussianModel(3, quantization=True, device=dev)
gaussians.set_tensors(xyz=sc["means3D"], features_dc=sc["shs"][:, :1], features_rest=sc["shs"][:, 1:],
                      scaling=sc["scales"] / sc["scales"].norm(dim=1, keepdim=True), rotation=sc["rotations"],
                      opacity=torch.log(op / (1 - op)), scaling_factor=torch.log(sc["scales"].norm(dim=1, keepdim=True)))
del sc
pipe = gm.PipelineParams()
bg = torch.zeros(3, device=dev)
cams = [Camera(yaw) for yaw in torch.linspace(-0.25, 0.25, args.cameras).tolist()]
with torch.no_grad():
    for c in cams:                                  # "ground truth" = the uncompressed model's own renders
        c.original_image = gaussians.render(c, pipe, bg)["render"].detach().clone()
print(f"scene: {args.scene}, {P} Gaussians, {len(cams)} cameras at {W}x{H}", flush=True)
psnr_oracle = {}
if oracle_img is not None:      # the uncompressed model's own render (with its FakeQuantize observers) against the oracle's
    psnr_oracle["uncompressed_hip_vs_oracle_dB"] = float(-10 * torch.log10(((cams[0].original_image - oracle_img) ** 2).mean()))

# ---- sensitivity (compress.py:218)
t0 = sync()
color_importance, gaussian_sensitivity = sensitivity.calc_importance_experimental(gaussians, cams, pipe, use_gt=True)
timings["sensitivity_calculation"] = sync() - t0
print("sensitivity", round(timings["sensitivity_calculation"], 2), "s", flush=True)

# ---- clustering (compress.py:223-257)
t0 = sync()
with torch.no_grad():
    color_comp = vqm.CompressionSettings(codebook_size=2 ** 12, importance_prune=0.0, importance_include=None,
                                         importance_include_relative=0.9, steps=args.color_steps, decay=0.8, batch_size=2 ** 18)
    gauss_comp = vqm.CompressionSettings(codebook_size=2 ** 12, importance_prune=None, importance_include=None,
                                         importance_include_relative=0.75, steps=args.gaussian_steps, decay=0.8,
                                         batch_size=2 ** 20)
    vqm.compress_gaussians(gaussians, color_importance.amax(-1), gaussian_sensitivity.amax(-1), color_comp, gauss_comp,
                           color_compress_non_dir=True, prune_threshold=0.0, silent=True)
timings["clustering"] = sync() - t0
survivors = int(gaussians._xyz.shape[0])
del color_importance, gaussian_sensitivity
torch.cuda.empty_cache()
print("clustering", round(timings["clustering"], 2), "s; codebooks", tuple(gaussians._features_dc.shape), tuple(gaussians._scaling.shape),
      flush=True)


def psnr_all():
    with torch.no_grad():
        vals = []
        for c in cams[:: max(1, len(cams) // 8)]:
            img = gaussians.render(c, pipe, bg)["render"]
            vals.append(float(-10 * torch.log10(((img - c.original_image) ** 2).mean())))
    return sum(vals) / len(vals)


psnr_vq = psnr_all()
print("PSNR after VQ (before fine-tuning)", round(psnr_vq, 2), "dB", flush=True)

# ---- QAT fine-tuning: the reference's loop and optimizer set-up (finetune.py:10-66, gaussian_model.py:292-322) as
# mirrored by c3dgs_amd.pipeline.finetune / GaussianModel.training_setup (fused Adam, one launch for all seven tensors)
import random
from c3dgs_amd import pipeline
random.seed(0)
gaussians.spatial_lr_scale = 1.0


class _Scene:
    loaded_iter = 0

    def getTrainCameras(self):
        return cams


_Scene.gaussians = gaussians
marks = []
t0 = sync()
pipeline.finetune(_Scene(), pipeline._Dataset(), pipeline.OptimizationParams(),
                  pipeline.CompressionParams(finetune_iterations=args.finetune), pipe,
                  log=lambda it, ema: marks.append((it, time.time() - t0)))      # called every 10 iterations, after a host read
ft = sync() - t0
# the first iterations of a process pay one-off costs (code objects, allocator growth, optimizer state: ~0.4 s); the
# steady per-iteration time is the slope after them
steady = [(i, t) for i, t in marks if i >= 20]
per_it = (steady[-1][1] - steady[0][1]) / max(steady[-1][0] - steady[0][0], 1) if len(steady) > 1 else ft / max(args.finetune, 1)
timings["finetune_measured"] = ft
timings["finetune_iterations"] = args.finetune
timings["finetune_ms_per_iteration"] = 1e3 * per_it
timings["finetune_first_iterations_overhead_s"] = ft - per_it * args.finetune
timings["finetune_5000_extrapolated"] = (ft - per_it * args.finetune) + 5000 * per_it
psnr_ft = psnr_all()
print("fine-tuning", round(ft, 2), "s for", args.finetune, "iterations; PSNR", round(psnr_ft, 2), "dB", flush=True)

# ---- encode (compress.py:281-286)
t0 = sync()
path = os.path.join(tempfile.mkdtemp(), "point_cloud.npz")
gaussians.save_npz(path, sort_morton=True)
timings["encode"] = time.time() - t0
size_mb = os.path.getsize(path) / 1024 ** 2
raw_mb = P * (3 + 48 + 3 + 4 + 1) * 4 / 1024 ** 2
if oracle_img is not None:
    with torch.no_grad():
        img0 = gaussians.render(cams[0], pipe, bg)["render"]
    psnr_oracle["compressed_hip_vs_oracle_uncompressed_dB"] = float(-10 * torch.log10(((img0 - oracle_img) ** 2).mean()))
res = {"scene": args.scene, "gaussians": P, "survivors_of_the_importance_prune": survivors, "cameras": len(cams), "resolution": [W, H],
       "timings_s": timings, "psnr_vs_oracle_camera0": psnr_oracle,
       "total_with_5000_iterations_s": timings["sensitivity_calculation"] + timings["clustering"] + timings["finetune_5000_extrapolated"] + timings["encode"],
       "payload_MiB": size_mb, "uncompressed_fp32_MiB": raw_mb, "compression_ratio": raw_mb / size_mb,
       "psnr_vs_uncompressed_after_vq_dB": psnr_vq, "psnr_vs_uncompressed_after_finetune_dB": psnr_ft,
       "wall_s": time.time() - t_all}
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
json.dump(res, open(args.out, "w"), indent=1)
print(json.dumps(res))
