"""Deterministic synthetic scenes (`synth-v1`) shared by tests and bench.py.

All tensors are generated with a CPU torch.Generator so that the same seed gives the same scene here
and on the GPU box.  Returned tensors are CPU float32 / int64; callers move them to the device.
"""
import math

import torch


def camera(W=1920, H=1080, focal=1200.0, extrinsic_vector=(0, 0, 0, 1, 0, 0, 0)):
    """intrinsic 3x3 as scene/cameras.py:39-41 encodes it ([0,0]=FoVx rad, [1,1]=FoVy rad, [0,2]=W, [1,2]=H)."""
    fovx = 2.0 * math.atan(W / (2.0 * focal))
    fovy = 2.0 * math.atan(H / (2.0 * focal))
    intrinsic = torch.tensor([[fovx, 0.0, float(W)], [0.0, fovy, float(H)], [0.0, 0.0, 1.0]], dtype=torch.float32)
    ev = torch.tensor(extrinsic_vector, dtype=torch.float32)
    return intrinsic, ev


def scene(P, W=1920, H=1080, focal=1200.0, seed=1234, sh_degree=3, scale_median=0.009, scale_sigma=0.6,
          zmin=2.0, zmax=12.0, behind_fraction=0.0):
    """Gaussians of synth-v1. Returns dict(means3D, scales, rotations, opacities[P,1], shs[P,M,3])."""
    g = torch.Generator().manual_seed(seed)
    tfx, tfy = W / (2.0 * focal), H / (2.0 * focal)
    z = torch.rand(P, generator=g) * (zmax - zmin) + zmin
    u = torch.rand(P, generator=g) * 2 - 1
    v = torch.rand(P, generator=g) * 2 - 1
    x = u * z * tfx * 1.05
    y = v * z * tfy * 1.05
    if behind_fraction > 0:
        flip = torch.rand(P, generator=g) < behind_fraction
        z = torch.where(flip, -z, z)
    means3D = torch.stack([x, y, z], 1).float()
    scales = torch.exp(torch.randn(P, 3, generator=g) * scale_sigma + math.log(scale_median)).float()
    rot = torch.randn(P, 4, generator=g)
    rotations = (rot / rot.norm(dim=1, keepdim=True)).float()
    opacities = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5 - 1.0).float()
    M = (sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g) * 0.05
    shs[:, 0] = torch.randn(P, 3, generator=g) * 0.5
    return dict(means3D=means3D.contiguous(), scales=scales.contiguous(), rotations=rotations.contiguous(),
                opacities=opacities.contiguous(), shs=shs.float().contiguous())


def index_scene(sc, seed=99, shs_extra=4096, gs_extra=4096, sh_frac=0.1, g_frac=0.25):
    """Indexed (post-VQ-like) variant: codebooks shs[SHS,M,3], scales[GS,3] (unit-normalised), rotations[GS,4],
    per-Gaussian sh_indices, g_indices, scale_factors = ||scales||."""
    g = torch.Generator().manual_seed(seed)
    P = sc["means3D"].shape[0]
    SHS = max(1, min(P, int(shs_extra + sh_frac * P)))
    GS = max(1, min(P, int(gs_extra + g_frac * P)))
    sh_sel = torch.randperm(P, generator=g)[:SHS]
    g_sel = torch.randperm(P, generator=g)[:GS]
    sh_indices = torch.randint(0, SHS, (P,), generator=g, dtype=torch.int64)
    g_indices = torch.randint(0, GS, (P,), generator=g, dtype=torch.int64)
    sc_cb = sc["scales"][g_sel]
    sc_cb = sc_cb / sc_cb.norm(dim=1, keepdim=True)
    scale_factors = sc["scales"].norm(dim=1, keepdim=True)
    return dict(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"][sh_sel].contiguous(),
                scales=sc_cb.contiguous(), rotations=sc["rotations"][g_sel].contiguous(),
                scale_factors=scale_factors.contiguous(), sh_indices=sh_indices, g_indices=g_indices)


def grad_image(W, H, seed=4321):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g).float()


def raw_params(ix):
    """Learnable (pre-activation) tensors of an indexed scene, as scene/gaussian_model.py stores them: _opacity =
    logit(opacity), _scaling_factor = log(scale factor), _scaling / _rotation codebooks, _features_dc / _features_rest
    halves of the SH codebook. Feeding them through the getters reproduces `ix` up to fake-quantisation."""
    op = ix["opacities"].clamp(1e-6, 1 - 1e-6)
    return dict(xyz=ix["means3D"].clone(), opacity=torch.log(op / (1 - op)).float(),
                scaling_factor=torch.log(ix["scale_factors"]).float(), scaling=ix["scales"].clone(),
                rotation=ix["rotations"].clone(), features_dc=ix["shs"][:, :1].contiguous(),
                features_rest=ix["shs"][:, 1:].contiguous(), feature_indices=ix["sh_indices"],
                gaussian_indices=ix["g_indices"])
