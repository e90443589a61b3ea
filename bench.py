#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): raster views/sec, forward+backward, 1920x1080.  The N=1 workload is
BASELINE.json configs[2]: 3M synthetic Gaussians (synth-v1), SH degree 3, the indexed-camera rasterizer the QAT loop
(finetune.py) uses, clamp_color=True.  A step = mark_visible + forward + backward of ONE view through the package's
autograd Function, inputs resident in HBM.  The raster path does not shard a single view ("replicas
only"), so with --gpus N every rank renders its own replica and `value` is the whole-job views/s (weak scaling).

Extra objects on the same JSON line:
  roofline      dominant kernel: algorithmic bytes per launch (SURVEY.md 8(d) formula) / live HIP-event duration
  cpu_baseline  the oracle (CPU restatement, OpenMP) timed on this box's host cores on a bounded sample
  stages        per-stage mean ms from the same HIP events (all kernels of the view)
  vq            sensitivity-weighted VQ Lloyd steps/s on config 4's colour shape, sharded over the N ranks (RCCL)
  (N = 1 only)  qat_loop: the step + fused L1/SSIM loss; qat_model: the whole QAT view / iteration from raw parameters (fused
                glue + fused Adam next to the reference's torch glue + torch Adam); postvq_index_layout: the headline step with
                the codebook indices laid out as compression/vq.py's join_features produces them
  device_allocs_in_timed_region   hipMalloc calls of torch's caching allocator during the K timed steps (expected 0)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # f32-in MFMA dense peak


def alg_bytes(stage, P, V, R, T, N, M, bit, indexed=True):
    """Algorithmic HBM bytes of one launch of `stage` -- the per-term formula of SURVEY.md section 8(d)."""
    G_in = 48 if indexed else 28
    G_out = 4 if indexed else 28
    f = {
        "mark_visible": P * (12 + 1),
        "preprocess": P * (12 + 4 + 4) + V * (4 + 12 * M + G_in) + V * (4 + 8 + 24 + 16 + 12 + 3),
        "scan": 8 * P,
        "duplicate_with_keys": 4 * P + 16 * V + 12 * R,
        "sort": (24 * math.ceil((32 + bit) / 8) + 8) * R,
        "identify_ranges": 8 * R + 8 * T,
        "render_forward": 8 * T + 40 * R + 20 * N,
        "zero_partials": 36 * R,
        "render_backward": 8 * T + 40 * R + 8 * N + 12 * N + 72 * R,
        "backward_preprocess": V * (12 + 4 + 24 + 16 + 24 + 12) + V * (12 + 12 + 12 * M + 3 + G_in + 24 + 12)
                               + V * (12 + 12 * M + G_out),
    }
    return float(f[stage])


def higher_msb(n):
    msb, step = 16, 16
    while step > 1:
        step //= 2
        msb = msb + step if (n >> msb) else msb - step
    return msb + 1 if (n >> msb) else msb


def build_workload(P, W, H, focal, device):
    from tests import synth
    intr, ev = synth.camera(W, H, focal)
    sc = synth.scene(P, W, H, focal, seed=1234, sh_degree=3)
    ix = synth.index_scene(sc)
    t = {k: v.to(device) for k, v in ix.items()}
    dL = synth.grad_image(W, H).to(device)
    return intr, ev, t, dL, ix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=3_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vq", action="store_true")
    ap.add_argument("--vq-steps", type=int, default=10)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import c3dgs_amd
    from c3dgs_amd import _lib
    _lib.lib()

    P, W, H, focal = args.gaussians, args.width, args.height, 1200.0
    intr, ev, t, dL, ix_cpu = build_workload(P, W, H, focal, dev)
    rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=intr, extrinsic_vector=ev.to(dev), bg=torch.zeros(3, device=dev),
                                                 scale_modifier=1.0, sh_degree=3, prefiltered=False, debug=False,
                                                 clamp_color=True)
    rast = c3dgs_amd.GaussianRasterizerIndexed(rs, optimize_camera=True)
    leaves = {k: t[k].clone().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "scale_factors", "rotations")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    evd = ev.to(dev)
    state = {}

    def step():
        for v in leaves.values():
            v.grad = None
        means2D.grad = None
        visible = rast.markVisible(leaves["means3D"], extrinsic_vector=evd)
        color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                            sh_indices=t["sh_indices"], g_indices=t["g_indices"], shs=leaves["shs"], scales=leaves["scales"],
                            scale_factors=leaves["scale_factors"], rotations=leaves["rotations"], extrinsic_vector=evd)
        torch.autograd.backward(color, dL)
        state["radii"], state["visible"] = radii, visible

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: run until the step time has settled (torch's caching allocator reaches its steady-state
    # block set after ~4 steps; a freshly acquired box can also take a while to ramp clocks / page the libraries in)
    prev = None
    for k in range(40):
        torch.cuda.synchronize()
        tp = time.perf_counter()
        step()
        torch.cuda.synchronize()
        cur = time.perf_counter() - tp
        if k >= 5 and prev is not None and abs(cur - prev) <= 0.05 * min(cur, prev):
            break
        prev = cur
    # every stage bracketed by HIP events: per-stage means and the dominant kernel (untimed pass; an event pair costs
    # ~10 us of queue time, x11 stages, which does not belong in `value`)
    _lib.profile_enable(True)
    _lib.profile_read()
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    stages = _lib.profile_read()
    _lib.profile_enable(False)
    raster_names = ("preprocess", "scan", "duplicate_with_keys", "sort", "identify_ranges", "render_forward", "zero_partials",
                    "render_backward", "backward_preprocess", "mark_visible", "depth_sort")
    dom = max((k for k in stages if k in raster_names), key=lambda k: stages[k][0] / max(stages[k][1], 1))
    barrier()
    # timed region: exactly K steps; only the dominant kernel carries an event pair (roofline.achieved is its live
    # average over these same K launches)
    _lib.profile_enable(True, only=dom)
    _lib.profile_read()
    n_alloc = torch.cuda.memory_stats(dev)["num_device_alloc"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_alloc = torch.cuda.memory_stats(dev)["num_device_alloc"] - n_alloc
    dom_live = _lib.profile_read()
    _lib.profile_enable(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    barrier()

    # workload statistics for the roofline
    radii = state["radii"]
    V = int((radii > 0).sum().item())
    from c3dgs_amd import rasterizer as rz
    with torch.no_grad():
        view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, evd, dev)
        R = rz._C.rasterize_gaussians_indexed(
            rs.bg, t["means3D"], torch.Tensor([]), t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0,
            torch.Tensor([]), view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False,
            True)[0]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    N = W * H
    bit = higher_msb(T)
    stage_ms = {k: v[0] / max(v[1], 1) for k, v in stages.items()}
    stage_ms[dom] = dom_live[dom][0] / max(dom_live[dom][1], 1)     # the timed region's own measurement
    raster_stages = [k for k in stage_ms if k in raster_names and k != "depth_sort"]
    traffic = None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (tools/pmc_traffic.sh)
        if P == 3_000_000 and (W, H) == (1920, 1080):
            with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
    except Exception:
        traffic = None
    dom_bytes = alg_bytes(dom, P, V, R, T, N, 16, bit, indexed=True)
    dom_gbs = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
    view_bytes = sum(alg_bytes(k, P, V, R, T, N, 16, bit, True) for k in raster_stages)

    out = {
        "metric": "raster views/sec fwd+bwd @1920x1080",
        "value": world * args.steps / elapsed,
        "unit": "views/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: synth-v1 3M Gaussians, SH deg 3, indexed-camera rasterizer "
                               "(QAT inner loop), mark_visible+fwd+bwd, 1 view/step/GPU",
                   "gaussians": P, "visible": V, "tile_instances": R, "width": W, "height": H,
                   "sh_codebook": int(t["shs"].shape[0]), "gaussian_codebook": int(t["scales"].shape[0]),
                   "parallelism": f"replicas x{world}"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom_gbs / HBM_PEAK_GBS, "traffic": traffic, "alg_bytes_per_launch": dom_bytes,
                     "avg_launch_ms": stage_ms[dom]},
        "stages_ms": {k: round(v, 4) for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1])},
        # informational: kernels of one step (untimed profiling pass) vs the timed step. A ratio far above 1 means the
        # GPU sat idle waiting for the host during the timed region (seen once on a heavily loaded box: profiles/README.md)
        "device_allocs_in_timed_region": n_alloc,     # hipMalloc calls of the caching allocator (expected: 0)
        "step_over_kernel_time": (1e3 * elapsed / args.steps) / max(sum(v for k, v in stage_ms.items() if k in raster_names), 1e-9),
        "view_alg_bytes": view_bytes,
        "view_hbm_frac": view_bytes * (args.steps / elapsed) / 1e9 / HBM_PEAK_GBS,
    }

    # ---- the same step with the fused L1+SSIM loss producing dL/dimage (QAT inner loop of finetune.py:40-49 without the
    # optimizer / FakeQuantize glue): extra, not the headline
    # (single-GPU extras: with N replicas they would only repeat the headline's weak scaling, behind more barriers)
    try:
        if world > 1:
            raise _SkipExtra()
        from c3dgs_amd import loss as lossm
        gt = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(5))

        def qat_step():
            for v in leaves.values():
                v.grad = None
            rast.markVisible(leaves["means3D"], extrinsic_vector=evd)
            color, _ = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                            sh_indices=t["sh_indices"], g_indices=t["g_indices"], shs=leaves["shs"], scales=leaves["scales"],
                            scale_factors=leaves["scale_factors"], rotations=leaves["rotations"], extrinsic_vector=evd)
            lossm.l1_ssim_loss(color, gt, 0.2).backward()

        for _ in range(3):
            qat_step()
        barrier()
        tq = time.perf_counter()
        for _ in range(args.steps):
            qat_step()
        torch.cuda.synchronize()
        out["qat_loop"] = {"metric": "views/s render + fused L1+SSIM loss + backward", "value": world * args.steps / (time.perf_counter() - tq)}
    except _SkipExtra:
        pass
    except Exception as e:
        out["qat_loop"] = {"error": repr(e)}

    # ---- the headline step again with the index layout compression/vq.py produces (join_features, vq.py:90-103): most
    # Gaussians point into the 4096-row VQ codebook, the kept ones own one row each, in Gaussian order. synth-v1's indices
    # (the headline) are uniformly random over the whole codebook -- the worst case for the codebook gathers.
    try:
        if world > 1:
            raise _SkipExtra()
        out["postvq_index_layout"] = bench_postvq_layout(step, t, P, args.steps, dev, _lib)
    except _SkipExtra:
        pass
    except Exception as e:
        out["postvq_index_layout"] = {"error": repr(e)}

    # ---- the whole QAT view from the RAW parameters -- getters (activations + FakeQuantize observers +
    # [visible] gathers) + raster + fused loss + backward -- through c3dgs_amd.model.GaussianModel.render (fused glue),
    # next to the reference's composition of the same glue from torch ops / torch.ao modules around the same rasterizer
    try:
        if world > 1:
            raise _SkipExtra()
        out["qat_model"] = bench_qat_model(c3dgs_amd, _lib, dev, ix_cpu, intr, evd, W, H, args.steps, barrier, world)
    except _SkipExtra:
        pass
    except Exception as e:
        out["qat_model"] = {"error": repr(e)}

    # ---- VQ (config 4 colour shape), sharded over the ranks with one all-reduce per Lloyd step
    if not args.no_vq:
        try:
            out["vq"] = bench_vq(c3dgs_amd, _lib, dev, rank, world, args.vq_steps)
        except Exception as e:  # keep the headline line alive
            out["vq"] = {"error": repr(e)}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(ix_cpu, intr, ev, W, H, focal)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_postvq_layout(step, t, P, steps, dev, _lib):
    """Same scene, same codebook sizes; only sh_indices / g_indices change (and so which rows the kernels gather)."""
    g = torch.Generator().manual_seed(17)

    def layout(n_rows, n_vq=4096):
        kept = max(min(n_rows - n_vq, P), 0)                      # rows past the VQ codebook: one per kept Gaussian
        idx = torch.randint(0, min(n_vq, n_rows), (P,), generator=g, dtype=torch.int64)
        if kept:
            who = torch.randperm(P, generator=g)[:kept].sort().values
            idx[who] = n_vq + torch.arange(kept, dtype=torch.int64)
        return idx.to(dev)

    saved = t["sh_indices"], t["g_indices"]
    try:
        t["sh_indices"], t["g_indices"] = layout(t["shs"].shape[0]), layout(t["scales"].shape[0])
        prev = None
        for k in range(40):                                        # until the step time has settled (as for the headline)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            step()
            torch.cuda.synchronize()
            cur = time.perf_counter() - tp
            if k >= 5 and prev is not None and abs(cur - prev) <= 0.05 * min(cur, prev):
                break
            prev = cur
        n_alloc = torch.cuda.memory_stats(dev)["num_device_alloc"]
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n_alloc = torch.cuda.memory_stats(dev)["num_device_alloc"] - n_alloc
        _lib.profile_enable(True)
        _lib.profile_read()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        st = _lib.profile_read()
        _lib.profile_enable(False)
    finally:
        t["sh_indices"], t["g_indices"] = saved
    return {"metric": "views/s fwd+bwd, indices as join_features lays them out", "value": steps / el, "ms_per_step": 1e3 * el / steps,
            "device_allocs_in_timed_loop": n_alloc,
            "stages_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])[:6]}}


class _SkipExtra(Exception):
    pass


def bench_qat_model(c3dgs_amd, _lib, dev, ix_cpu, intr, evd, W, H, steps, barrier, world):
    from c3dgs_amd import loss as lossm
    from c3dgs_amd import model as gm
    from tests import synth
    raw = synth.raw_params(ix_cpu)
    m = gm.GaussianModel(3, quantization=True, device=dev).set_tensors(**raw)

    class Cam:
        intrinsic, extrinsic_vector = intr.to(dev), evd
    cam, pipe, bg = Cam(), gm.PipelineParams(), torch.zeros(3, device=dev)
    gt = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(5))

    def fused_step():
        for p in m.parameters():
            p.grad = None
        lossm.l1_ssim_loss(m.render(cam, pipe, bg)["render"], gt, 0.2).backward()

    # the reference's glue (scene/gaussian_model.py:213-267, 766-886): torch ops + torch.ao modules + boolean gathers
    mods = {k: torch.ao.quantization.FakeQuantize(dtype=torch.qint8).to(dev) for k in gm.SLOTS}
    leaves = {k: v.to(dev).clone().requires_grad_(v.is_floating_point()) for k, v in raw.items()}
    rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=cam.intrinsic, extrinsic_vector=evd, bg=bg, scale_modifier=1.0,
                                                 sh_degree=3, prefiltered=False, debug=False, clamp_color=True)
    rast = c3dgs_amd.GaussianRasterizerIndexed(rs, optimize_camera=True)
    nz = torch.nn.functional.normalize

    def torch_glue_step():
        for v in leaves.values():
            v.grad = None
        xyz = gm.FakeQuantizationHalf.apply(leaves["xyz"])
        opacity = mods["opacity"](torch.sigmoid(leaves["opacity"]))
        scales = mods["scaling"](nz(torch.relu(leaves["scaling"])))
        rotations = nz(mods["rotation"](leaves["rotation"]))
        sfac = torch.exp(mods["scaling_factor"](leaves["scaling_factor"]))
        shs = torch.cat((mods["features_dc"](leaves["features_dc"]), mods["features_rest"](leaves["features_rest"])), dim=1)
        screen = torch.zeros_like(xyz, requires_grad=True)
        vis = rast.markVisible(xyz, extrinsic_vector=evd)
        color, _ = rast(means3D=xyz[vis], means2D=screen[vis], shs=shs, sh_indices=leaves["feature_indices"][vis],
                        g_indices=leaves["gaussian_indices"][vis], colors_precomp=None, opacities=opacity[vis], scales=scales,
                        scale_factors=sfac[vis], rotations=rotations, cov3D_precomp=None, extrinsic_vector=evd)
        lossm.l1_ssim_loss(color, gt, 0.2).backward()

    # the complete QAT iteration of finetune.py:29-66 incl. optimizer.step(): per-group learning rates of
    # scene/gaussian_model.py:296-308, eps 1e-15
    from c3dgs_amd import optim as optm
    lrs = (0.00016, 0.0025, 0.0025 / 20.0, 0.005, 0.005, 0.001, 0.05)

    def groups(ps):
        return [{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)]
    opt_fused = optm.Adam(groups(m.parameters()), lr=0.0, eps=1e-15)
    opt_torch = torch.optim.Adam(groups([leaves[k] for k in ("xyz", "features_dc", "features_rest", "scaling", "scaling_factor",
                                                               "rotation", "opacity")]), lr=0.0, eps=1e-15)

    def fused_iteration():
        lossm.l1_ssim_loss(m.render(cam, pipe, bg)["render"], gt, 0.2).backward()
        opt_fused.step()
        opt_fused.zero_grad(set_to_none=True)

    def torch_iteration():
        torch_glue_step()
        opt_torch.step()
        opt_torch.zero_grad(set_to_none=True)

    res = {"metric": "views/s of a whole QAT view from raw parameters (getters + raster + L1/SSIM loss + backward)"}
    for name, fn in (("fused_glue", fused_step), ("torch_glue", torch_glue_step), ("iteration_fused_glue_fused_adam", fused_iteration),
                     ("iteration_torch_glue_torch_adam", torch_iteration)):
        for _ in range(4):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res[name] = {"views_per_s": world * steps / el, "ms_per_view": 1e3 * el / steps}
    _lib.profile_enable(True)                                  # separate, untimed pass for the glue kernels' durations
    _lib.profile_read()
    for _ in range(5):
        fused_step()
    torch.cuda.synchronize()
    st = _lib.profile_read()
    _lib.profile_enable(False)
    res["glue_stages_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if k.startswith("qat_")}
    _lib.profile_enable(True, only="adam_step")
    _lib.profile_read()
    for _ in range(5):
        fused_iteration()
    torch.cuda.synchronize()
    st = _lib.profile_read()
    _lib.profile_enable(False)
    res["adam_step_ms"] = round(st["adam_step"][0] / max(st["adam_step"][1], 1), 4) if "adam_step" in st else None
    res["glue_only_ms"] = {"fused": round(sum(res["glue_stages_ms"].values()), 4)}
    return res


def bench_vq(c3dgs_amd, _lib, dev, rank, world, steps):
    """Config 4 colour codebook: N=5.4M x 48 features, K=4096, batch 2^18 split over the ranks."""
    from c3dgs_amd import vq as vqm
    g = torch.Generator(device=dev).manual_seed(7)
    N, D, K, B = 5_400_000, 48, 4096, 2 ** 18
    feats = torch.randn(N, D, device=dev, generator=g) * 0.1
    imp = torch.rand(N, device=dev, generator=g).pow(4)
    model = vqm.VectorQuantize(D, K, decay=0.8).to(dev)
    model.uniform_init(feats, torch.rand(K, D, device=dev, generator=g))
    cpu_g = torch.Generator().manual_seed(11)
    batches = [torch.randint(0, N, (B,), generator=cpu_g).to(dev) for _ in range(steps + 2)]

    scratch = {}

    def one(b):
        lo, hi = (rank * B) // world, ((rank + 1) * B) // world
        _, S, dsum = model.partial_sums(feats, imp, gather=b[lo:hi].contiguous(), scratch=scratch)
        if world > 1:
            S, dsum = vqm.all_reduce_sums(dist, None, S, dsum)
        model.apply_sums(S)

    for b in batches[:2]:
        one(b)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    _lib.profile_read()
    t0 = time.perf_counter()
    for b in batches[2:]:
        one(b)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = _lib.profile_read()
    _lib.profile_enable(False)
    elt = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
    el = float(elt.item())
    wd_ms = st.get("weighted_distance", (0, 1))[0] / max(st.get("weighted_distance", (0, 1))[1], 1)
    flops = 2.0 * (B / world) * K * D
    return {"metric": "vq_lloyd_steps_per_s", "value": steps / el, "ms_per_step": 1e3 * el / steps, "n_gpus": world,
            "scaling": "strong", "config": {"workload": "config 4 colour: N=5.4M, D=48, K=4096, batch 2^18", "batch": B},
            "assign_kernel_ms": wd_ms,
            "roofline": {"bound": "mfma", "kernel": "weighted_distance", "achieved": flops / (wd_ms * 1e-3) / 1e12 if wd_ms else None,
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / (wd_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if wd_ms else None, "traffic": None}}


def cpu_baseline(ix_cpu, intr, ev, W, H, focal):
    """The oracle (`kind: port`, oracle/c3dgs_oracle.c, OpenMP) on this box's host cores: one fwd+bwd view of the SAME
    workload. A 10 % Gaussian subsample is timed first; if that predicts more than ~45 s for the full view the scaled
    subsample figure is reported instead (the sample field says which)."""
    import numpy as np
    from oracle import oracle as orc
    from tests import synth
    cam = orc.camera(intr.numpy(), ev.numpy())
    P = ix_cpu["means3D"].shape[0]
    dL = synth.grad_image(W, H).numpy()

    def run(n):
        sel = slice(0, n)
        t0 = time.perf_counter()
        st = orc.rasterize_forward(bg=np.zeros(3, np.float32), means3D=ix_cpu["means3D"][sel].numpy(),
                                   opacities=ix_cpu["opacities"][sel].numpy(), shs=ix_cpu["shs"].numpy(),
                                   scales=ix_cpu["scales"].numpy(), rotations=ix_cpu["rotations"].numpy(),
                                   scale_factors=ix_cpu["scale_factors"][sel].numpy(),
                                   sh_indices=ix_cpu["sh_indices"][sel].numpy(), g_indices=ix_cpu["g_indices"][sel].numpy(),
                                   degree=3, clamp_color=True, **cam)
        orc.rasterize_backward(st, dL)
        return time.perf_counter() - t0, st.num_rendered

    sub = max(1, P // 10)
    dt_sub, r_sub = run(sub)
    if dt_sub * (P / sub) <= 45.0:
        dt, r = run(P)
        return {"value": 1.0 / dt, "unit": "views/s", "cores": orc.num_threads(), "kind": "port",
                "sample": f"1 full view fwd+bwd, {P} Gaussians at {W}x{H} ({dt:.2f} s, R={r}); oracle/c3dgs_oracle.c, OpenMP"}
    scale = P / sub
    return {"value": 1.0 / (dt_sub * scale), "unit": "views/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"1 view fwd+bwd of the first {sub} of {P} Gaussians at {W}x{H} ({dt_sub:.2f} s, R={r_sub}), "
                      f"time scaled x{scale:.0f}; oracle/c3dgs_oracle.c, OpenMP"}


if __name__ == "__main__":
    main()
