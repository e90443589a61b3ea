#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING the reference's importable Python in the build container.

    python tests/golden/make_golden.py            (needs /root/reference; never runs on the GPU box)

What it pins; the fixtures hold data only (inputs + expected outputs):
  vq_*.npz       compression/vq.py:vq_features end to end on CPU tensors, with its two missing native deps
                 shimmed (torch_scatter.scatter == index_add_, weightedDistance == exact direct-difference
                 argmin) and its RNG draws (rand_like of uniform_init, randint batches) captured as data
  sh.npz         utils/sh_utils.py:eval_sh                       -> K2's SH->RGB (before +0.5 / clamp)
  cov3d.npz      utils/general_utils.py:build_covariance_from_scaling_rotation -> K2's cov3D
  sh_bwd.npz     torch.autograd THROUGH eval_sh on dirs = normalize(pos - campos), as the reference's own python colour path
                 does (scene/gaussian_model.py:828-832): d/dsh and d/dpos -> pins backward.cu:20-139 (SH backward incl. the
                 direction-normalisation Jacobian dnormvdv); positions lie in front of a posed camera so that the same
                 fixture can be rasterized on the GPU
  cov3d_bwd.npz  torch.autograd through build_covariance_from_scaling_rotation: Jacobians d cov6 / d scale and
                 d cov6 / d rotation (the reference's build_rotation normalises q, so for unit q this is the TANGENTIAL part
                 of the kernel's un-normalised-quaternion gradient) + one random upstream -> pins backward.cu:278-341
  vq_d48.npz     vq_features at D = 48 (N=20000, K=512, 8 steps of 4096), seed-only: features / importance come from a seeded
  vq_config0.npz generator, the draws from torch.manual_seed -- and BASELINE.json configs[0] exactly (N=10000, D=12, K=256,
                 100 steps of 2^14, seed 0). Stored: seeds, the uniform_init draw, final codebook, final indices, per-step error
  loss.npz       utils/loss_utils.py: (1-l)*l1_loss + l*(1-ssim) and its autograd gradient (finetune.py:48)
  morton.npz     mortonEncode/splitBy3 (scene/gaussian_model.py:1417-1432) on _sort_morton's quantisation (:999-1003)
  splats.npz     utils/splats.py: extract_rot_scale(to_full_cov(cov6)) and build_covariance of its result (pure torch,
                 importable): the eigendecomposition step of compress_covariance (compression/vq.py:186)
  lr.npz         utils/general_utils.py:get_expon_lr_func (the xyz learning-rate schedule of finetune.py's loop)
  camera.npz     getProjectionMatrix / quat_to_mat / mat_to_quat (diff_gaussian_rasterization_no_camera/__init__.py:19-52,
                 text executed with `.cuda()` dropped) and the camera set-up of the autograd wrappers built from them
                 (:152-172: tan(FoV/2), H, W, `extrinsic @ getProjectionMatrix(...)`, `extrinsic.inverse()[3, :3]`)
  camgrad.npz    the closed-form grad_params block of _RasterizeGaussiansIndexedCamera.backward
                 (diff_gaussian_rasterization_no_camera/__init__.py:674-844), executed on CPU tensors
"""
import os
import re
import sys
import textwrap
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def shim_and_import_vq():
    sys.path.insert(0, REF)
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, reduce="sum", dim_size=None):
        assert reduce == "sum" and dim == 0
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        return out.index_add_(0, index, src)
    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    wd = types.ModuleType("weighted_distance")
    wdc = types.ModuleType("weighted_distance._C")

    def weightedDistance(x, cb):
        outd, outi = [], []
        for s in range(0, x.shape[0], 4096):
            d = ((x[s:s + 4096, None] - cb[None]) ** 2).sum(-1)
            m, i = d.min(1)
            outd.append(m)
            outi.append(i)
        return torch.cat(outd), torch.cat(outi)
    wdc.weightedDistance = weightedDistance
    wd._C = wdc
    sys.modules["weighted_distance"] = wd
    sys.modules["weighted_distance._C"] = wdc
    import compression.vq as vq
    torch.cuda.synchronize = lambda *a, **k: None        # vq.py:83 calls it unconditionally
    return vq


def gen_vq(vq, name, N, D, K, steps, chunk, scale_normalize, seed):
    g = torch.Generator().manual_seed(seed)
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    if scale_normalize:
        f[:, [0, 3, 5]] = f[:, [0, 3, 5]].abs() + 0.2
    imp = torch.rand(N, generator=g).pow(4).float()
    draws = {"rand": None, "batches": []}
    orig_rand_like, orig_randint = torch.rand_like, torch.randint

    def rand_like(t, *a, **k):
        r = orig_rand_like(t, *a, **k)
        draws["rand"] = r.clone()
        return r

    def randint(*a, **k):
        r = orig_randint(*a, **k)
        draws["batches"].append(r.clone())
        return r
    torch.manual_seed(seed)
    torch.rand_like, torch.randint = rand_like, randint
    try:
        errors = []
        import builtins
        cb, idx = vq.vq_features(f, imp, K, chunk, steps, scale_normalize=scale_normalize, silent=True)
    finally:
        torch.rand_like, torch.randint = orig_rand_like, orig_randint
    np.savez_compressed(os.path.join(OUT, name), features=f.numpy(), importance=imp.numpy(), init_rand=draws["rand"].numpy(),
                        batches=torch.stack(draws["batches"]).numpy().astype(np.int64), codebook=cb.numpy(),
                        indices=idx.numpy().astype(np.int64), scale_normalize=np.array(int(scale_normalize)),
                        K=np.array(K))
    print(name, "codebook", tuple(cb.shape), "steps", len(draws["batches"]))


def gen_sh():
    from utils.sh_utils import eval_sh
    g = torch.Generator().manual_seed(5)
    P = 257
    sh = torch.randn(P, 16, 3, generator=g).float()          # kernel layout [P, M, 3]
    d = torch.randn(P, 3, generator=g)
    d = (d / d.norm(dim=1, keepdim=True)).float()
    out = {}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = eval_sh(deg, sh.transpose(1, 2), d).numpy()   # eval_sh wants [..., 3, M]
    np.savez_compressed(os.path.join(OUT, "sh.npz"), sh=sh.numpy(), dirs=d.numpy(), **out)
    print("sh.npz")


def gen_cov3d():
    from utils.general_utils import build_covariance_from_scaling_rotation
    g = torch.Generator().manual_seed(6)
    P = 300
    s = torch.exp(torch.randn(P, 3, generator=g) * 0.5 - 3.0).float()
    q = torch.randn(P, 4, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).float()
    out = {}
    for mod in (1.0, 1.7):
        out[f"cov_mod{mod}"] = build_covariance_from_scaling_rotation(s, mod, q).numpy()
    np.savez_compressed(os.path.join(OUT, "cov3d.npz"), scales=s.numpy(), rotations=q.numpy(), **out)
    print("cov3d.npz")


def gen_sh_bwd():
    """autograd through utils/sh_utils.py:eval_sh on normalised view directions (the reference's python colour path,
    scene/gaussian_model.py:828-832: dir_pp = xyz - camera_center; dir_pp / dir_pp.norm(); eval_sh)."""
    import math
    from utils.sh_utils import eval_sh
    src = open(os.path.join(REF, "submodules/diff-gaussian-rasterization-no-camera/"
                                 "diff_gaussian_rasterization_no_camera/__init__.py")).read().split("\n")
    ns = dict(torch=torch, math=math)
    exec("\n".join(src[18:52]).replace(".cuda()", ""), ns)            # getProjectionMatrix / quat_to_mat (see gen_camera)
    g = torch.Generator().manual_seed(15)
    P, W, H, focal = 257, 160, 112, 120.0
    q = torch.tensor([0.05, -0.03, 0.02, 0.99])
    ev = torch.cat([q, torch.tensor([0.1, -0.05, 0.2])]).float()
    intr = torch.tensor([[2 * math.atan(W / (2 * focal)), 0, float(W)], [0, 2 * math.atan(H / (2 * focal)), float(H)], [0, 0, 1]],
                        dtype=torch.float32)
    view = ns["quat_to_mat"](ev)                                       # W2C transposed (row-vector convention)
    campos = view.inverse()[3, :3]                                     # __init__.py:176
    # points in camera space in front of the camera, mapped to world space with the inverse pose: p_c = p_w @ view[:3,:3] + view[3,:3]
    z = torch.rand(P, generator=g) * 4 + 2
    pc = torch.stack([(torch.rand(P, generator=g) * 2 - 1) * z * W / (2 * focal) * 0.9,
                      (torch.rand(P, generator=g) * 2 - 1) * z * H / (2 * focal) * 0.9, z], 1).double()
    pos = ((pc - view[3, :3].double()) @ view[:3, :3].double().inverse()).float().contiguous()
    sh = torch.randn(P, 16, 3, generator=g).float() * 0.3
    sh[:, 0] = torch.randn(P, 3, generator=g) * 0.5
    up = torch.randn(P, 3, generator=g).float()
    out = dict(pos=pos.numpy(), campos=campos.numpy(), extrinsic_vector=ev.numpy(), intrinsic=intr.numpy(), sh=sh.numpy(),
               upstream=up.numpy(), W=np.array(W), H=np.array(H))
    for deg in range(4):
        shp = sh.clone().requires_grad_()
        posp = pos.clone().requires_grad_()
        d = posp - campos
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh(deg, shp.transpose(1, 2), d)
        (rgb * up).sum().backward()
        out[f"dsh_deg{deg}"] = shp.grad.numpy()
        out[f"dpos_deg{deg}"] = posp.grad.numpy() if posp.grad is not None else np.zeros((P, 3), np.float32)   # deg 0: no direction
        shp2 = sh.clone().requires_grad_()
        d2 = pos - campos
        eval_sh(deg, shp2.transpose(1, 2), d2 / d2.norm(dim=1, keepdim=True))[:, 0].sum().backward()
        out[f"basis_deg{deg}"] = shp2.grad[:, :, 0].numpy()           # d rgb[p, c] / d sh[p, k, c] = basis_k(dir_p), any c
    np.savez_compressed(os.path.join(OUT, "sh_bwd.npz"), **out)
    print("sh_bwd.npz")


def gen_cov3d_bwd():
    """autograd through utils/general_utils.py:build_covariance_from_scaling_rotation (strip_symmetric(L L^T), L = R(q/|q|) S)."""
    from utils.general_utils import build_covariance_from_scaling_rotation
    g = torch.Generator().manual_seed(16)
    P = 257
    s = torch.exp(torch.randn(P, 3, generator=g) * 0.5 - 3.2).float()
    q = torch.randn(P, 4, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).float()
    up = torch.randn(P, 6, generator=g).float()
    out = dict(scales=s.numpy(), rotations=q.numpy(), upstream=up.numpy())
    for mod in (1.0, 1.7):
        js, jq = np.zeros((P, 6, 3), np.float32), np.zeros((P, 6, 4), np.float32)
        for k in range(6):
            sp, qp = s.clone().requires_grad_(), q.clone().requires_grad_()
            build_covariance_from_scaling_rotation(sp, mod, qp)[:, k].sum().backward()
            js[:, k], jq[:, k] = sp.grad.numpy(), qp.grad.numpy()
        sp, qp = s.clone().requires_grad_(), q.clone().requires_grad_()
        (build_covariance_from_scaling_rotation(sp, mod, qp) * up).sum().backward()
        out.update({f"jac_scale_mod{mod}": js, f"jac_rot_mod{mod}": jq, f"dscale_mod{mod}": sp.grad.numpy(),
                    f"drot_mod{mod}": qp.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "cov3d_bwd.npz"), **out)
    print("cov3d_bwd.npz")


def seeded_vq_inputs(N, D, data_seed):
    """The synthetic features / importance of SURVEY 8(d) config 1, from a seeded CPU generator (regenerated by the tests)."""
    g = torch.Generator().manual_seed(data_seed)
    f = (torch.randn(N, D, generator=g) * 0.1).float()
    imp = torch.rand(N, generator=g).pow(4).float()
    return f, imp


def gen_vq_seeded(vq, name, N, D, K, steps, chunk, data_seed, seed):
    """vq_features driven by torch.manual_seed(seed) only: VectorQuantize.__init__'s kaiming_uniform_ (vq.py:19-21), the
    rand_like of uniform_init (:26) and every randint batch (:69) come from the CPU default generator in that order."""
    f, imp = seeded_vq_inputs(N, D, data_seed)
    draws = {"rand": None}
    orig_rand_like = torch.rand_like

    def rand_like(t, *a, **k):
        r = orig_rand_like(t, *a, **k)
        draws["rand"] = r.clone()
        return r
    errors = []
    import builtins
    torch.manual_seed(seed)
    torch.rand_like = rand_like
    try:
        cb, idx = vq.vq_features(f, imp, K, chunk, steps, silent=True)
    finally:
        torch.rand_like = orig_rand_like
    np.savez_compressed(os.path.join(OUT, name), N=np.array(N), D=np.array(D), K=np.array(K), steps=np.array(steps),
                        chunk=np.array(chunk), data_seed=np.array(data_seed), seed=np.array(seed),
                        init_rand=draws["rand"].numpy(), codebook=cb.numpy(),
                        indices=idx.numpy().astype(np.int16 if K <= 32767 else np.int32),
                        features_checksum=np.array(float(f.double().sum())), importance_checksum=np.array(float(imp.double().sum())))
    print(name, "codebook", tuple(cb.shape))


def gen_camgrad():
    src = open(os.path.join(REF, "submodules/diff-gaussian-rasterization-no-camera/"
                                 "diff_gaussian_rasterization_no_camera/__init__.py")).read().split("\n")
    block = "\n".join(src[673:788])                           # lines 674..788: X,Y,Z ... grad_params[:, 6, 1]
    block = textwrap.dedent(block).replace('device="cuda"', 'device="cpu"')
    g = torch.Generator().manual_seed(9)
    P = 64
    means3D = torch.cat([torch.randn(P, 2, generator=g), torch.rand(P, 1, generator=g) * 5 + 2], 1).float()
    intrinsic = torch.tensor([[1.2, 0, 640.0], [0, 0.9, 480.0], [0, 0, 1]], dtype=torch.float32)
    q = torch.tensor([0.05, -0.03, 0.02, 0.99])
    ev = torch.cat([q / q.norm(), torch.tensor([0.1, -0.05, 0.2])]).float()
    ns = dict(torch=torch, means3D=means3D, extrinsic_vector=ev,
              raster_settings=types.SimpleNamespace(intrinsic=intrinsic))
    exec(block, ns)
    gp = ns["grad_params"]
    du, dv = torch.randn(P, generator=g).float(), torch.randn(P, generator=g).float()
    grad_mat = torch.stack([(gp[:, p, 0] * du + gp[:, p, 1] * dv).sum() for p in range(7)])
    np.savez_compressed(os.path.join(OUT, "camgrad.npz"), means3D=means3D.numpy(), intrinsic=intrinsic.numpy(),
                        extrinsic_vector=ev.numpy(), du=du.numpy(), dv=dv.numpy(), grad_params=gp.numpy(),
                        grad_mat=grad_mat.numpy())
    print("camgrad.npz")


def gen_camera():
    """Lines 19-52 of the rasterizer package's __init__.py are self-contained (math + torch); the package itself cannot be
    imported (its `_C` extension is CUDA-only), so their text is executed here with the `.cuda()` calls dropped. fp32
    element arithmetic on CPU tensors is the same IEEE arithmetic the reference runs on 0-dim CUDA tensors."""
    import math
    src = open(os.path.join(REF, "submodules/diff-gaussian-rasterization-no-camera/"
                                 "diff_gaussian_rasterization_no_camera/__init__.py")).read().split("\n")
    ns = dict(torch=torch, math=math)
    exec("\n".join(src[18:52]).replace(".cuda()", ""), ns)
    g = torch.Generator().manual_seed(31)
    poses, intrs = [torch.tensor([0., 0, 0, 1, 0, 0, 0])], [torch.tensor([[1.35, 0, 1920.], [0, 0.85, 1080.], [0, 0, 1]])]
    for k in range(7):
        q = torch.randn(4, generator=g)
        q = q / q.norm() if k < 5 else q * 0.7                 # the last two: NOT normalised (the reference never normalises)
        poses.append(torch.cat([q, torch.randn(3, generator=g) * (0.3 + k)]).float())
        fx, fy = float(torch.rand(1, generator=g)) * 1.5 + 0.3, float(torch.rand(1, generator=g)) * 1.2 + 0.3
        intrs.append(torch.tensor([[fx, 0, float(100 + 237 * k)], [0, fy, float(50 + 131 * k)], [0, 0, 1]], dtype=torch.float32))
    out = dict(extrinsic_vector=torch.stack(poses).numpy(), intrinsic=torch.stack(intrs).numpy())
    views, Ps, projs, campos, scal, quats = [], [], [], [], [], []
    for ev, intr in zip(poses, intrs):
        extrinsic = ns["quat_to_mat"](ev)
        Pm = ns["getProjectionMatrix"](intr)
        views.append(extrinsic.numpy())
        Ps.append(Pm.numpy())
        projs.append((extrinsic @ Pm).numpy())                                     # __init__.py:171
        campos.append(extrinsic.inverse()[3, :3].numpy())                          # :176
        scal.append([float(math.tan(intr[0, 0] * 0.5)), float(math.tan(intr[1, 1] * 0.5)), int(intr[1, 2]), int(intr[0, 2])])  # :152-155
        quats.append(torch.stack([torch.as_tensor(v) for v in ns["mat_to_quat"](extrinsic.transpose(0, 1))]).numpy())
    out.update(view=np.stack(views), P=np.stack(Ps), proj=np.stack(projs), campos=np.stack(campos),
               scalars=np.array(scal, dtype=np.float64), mat_to_quat=np.stack(quats))
    np.savez_compressed(os.path.join(OUT, "camera.npz"), **out)
    print("camera.npz")


def gen_morton():
    """mortonEncode / splitBy3 (scene/gaussian_model.py:1417-1432) executed on the quantised positions of
    _sort_morton (:999-1003). The module itself is not importable (simple_knn, plyfile), the two functions are
    self-contained, so their text is executed here."""
    src = open(os.path.join(REF, "scene/gaussian_model.py")).read().split("\n")
    ns = dict(torch=torch)
    exec("\n".join(src[1416:1432]), ns)
    g = torch.Generator().manual_seed(4)
    xyz = (torch.randn(5000, 3, generator=g) * torch.tensor([3.0, 0.7, 1.9])).float()
    pp_min = xyz.min(0).values
    pp_diap = xyz.max(0).values - pp_min
    xyz_q = ((2 ** 21 - 1) * (xyz - pp_min) / pp_diap).long()
    codes = ns["mortonEncode"](xyz_q, pp_diap.argsort())
    np.savez_compressed(os.path.join(OUT, "morton.npz"), xyz=xyz.numpy(), codes=codes.numpy().astype(np.int64),
                        axis_order=pp_diap.argsort().numpy().astype(np.int32))
    print("morton.npz")


def gen_splats():
    from utils.splats import build_covariance, extract_rot_scale, to_full_cov
    g = torch.Generator().manual_seed(12)
    n = 600
    s = torch.exp(torch.randn(n, 3, generator=g) * 0.8)
    s = s / s.norm(dim=1, keepdim=True)                                   # normalised scaling, as get_normalized_covariance
    s[:40, 1] = s[:40, 0]                                                 # repeated eigenvalue
    s[40:60, 2] = 1e-5                                                    # nearly flat
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    from utils.general_utils import build_covariance_from_scaling_rotation
    cov6 = build_covariance_from_scaling_rotation(s, 1.0, q).float()
    cov6[60:80] = (cov6[60:80] + cov6[80:100]) * 0.5                      # codebook entries are weighted MEANS of covariances
    cov6[100] = 0.0                                                       # degenerate: all-zero entry
    cov6[101] = torch.tensor([1.0, 0, 0, 1.0, 0, 1.0])                    # identity
    rot, scaling = extract_rot_scale(to_full_cov(cov6))
    rebuilt = build_covariance(rot, scaling)
    np.savez_compressed(os.path.join(OUT, "splats.npz"), cov6=cov6.numpy(), rot=rot.numpy(), scaling=scaling.numpy(),
                        rebuilt=rebuilt.numpy())
    print("splats.npz")


def gen_loss():
    """finetune.py:48 loss and its autograd gradient, utils/loss_utils.py (pure torch, importable)."""
    from utils.loss_utils import l1_loss, ssim
    out = {}
    for tag, (C_, H, W, seed) in {"a": (3, 37, 53, 0), "b": (3, 16, 16, 1), "c": (1, 9, 40, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        gt = torch.rand(C_, H, W, generator=g)
        img = (gt + 0.15 * torch.randn(C_, H, W, generator=g)).clamp(0, 1).requires_grad_()
        lam = 0.2
        s_ = ssim(img, gt)
        l_ = l1_loss(img, gt)
        loss = (1 - lam) * l_ + lam * (1 - s_)
        loss.backward()
        out.update({f"img_{tag}": img.detach().numpy(), f"gt_{tag}": gt.numpy(), f"loss_{tag}": np.array(float(loss.detach())),
                    f"ssim_{tag}": np.array(float(s_.detach())), f"l1_{tag}": np.array(float(l_.detach())),
                    f"grad_{tag}": img.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)
    print("loss.npz")


def gen_lr():
    """utils/general_utils.py:get_expon_lr_func, the xyz learning-rate schedule of the fine-tuning loop (pure numpy)."""
    sys.path.insert(0, REF)
    from utils.general_utils import get_expon_lr_func
    params = np.array([[1.6e-4, 1.6e-6, 0, 0.01, 30000], [3.2e-4, 3.2e-6, 0, 0.01, 100], [1e-2, 1e-4, 100, 0.1, 1000],
                       [0.0, 0.0, 0, 1.0, 1000]], dtype=np.float64)       # lr_init, lr_final, delay_steps, delay_mult, max_steps
    steps = np.array([-1, 0, 1, 10, 25, 50, 100, 999, 1000, 5000, 30000, 10 ** 6], dtype=np.int64)
    values = np.array([[float(get_expon_lr_func(a, b, lr_delay_steps=int(d), lr_delay_mult=m, max_steps=int(n))(int(st)))
                        for st in steps] for a, b, d, m, n in params])
    np.savez_compressed(os.path.join(OUT, "lr.npz"), params=params, steps=steps, values=values)
    print("lr.npz")


if __name__ == "__main__":
    if sys.argv[1:] == ["lr"]:
        gen_lr()
        sys.exit(0)
    if sys.argv[1:] == ["camera"]:
        gen_camera()
        sys.exit(0)
    if sys.argv[1:] == ["bwd"]:
        sys.path.insert(0, REF)
        gen_sh_bwd()
        gen_cov3d_bwd()
        sys.exit(0)
    if sys.argv[1:] == ["vq_seeded"]:
        vq = shim_and_import_vq()
        gen_vq_seeded(vq, "vq_d48.npz", N=20000, D=48, K=512, steps=8, chunk=4096, data_seed=48, seed=3)
        gen_vq_seeded(vq, "vq_config0.npz", N=10000, D=12, K=256, steps=100, chunk=2 ** 14, data_seed=0, seed=0)
        sys.exit(0)
    vq = shim_and_import_vq()
    gen_vq(vq, "vq_color.npz", N=3000, D=12, K=64, steps=12, chunk=1024, scale_normalize=False, seed=0)
    gen_vq(vq, "vq_cov.npz", N=2500, D=6, K=32, steps=10, chunk=512, scale_normalize=True, seed=1)
    gen_vq_seeded(vq, "vq_d48.npz", N=20000, D=48, K=512, steps=8, chunk=4096, data_seed=48, seed=3)
    gen_vq_seeded(vq, "vq_config0.npz", N=10000, D=12, K=256, steps=100, chunk=2 ** 14, data_seed=0, seed=0)
    gen_sh()
    gen_cov3d()
    gen_sh_bwd()
    gen_cov3d_bwd()
    gen_camgrad()
    gen_camera()
    gen_loss()
    gen_morton()
    gen_splats()
    gen_lr()
