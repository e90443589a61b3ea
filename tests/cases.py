"""Raster parity cases shared by the CPU (oracle vs float64 autograd) and GPU (HIP vs oracle) tests.
Edge cases follow SURVEY.md section 7's list: P=0, R=0, Gaussians behind the camera, image size not a multiple
of 16, SH degree 0-3, clamp_color on/off, precomputed covariance / colour variants, indexed codebooks,
t-clamp at the frustum edge, scale_modifier != 1, non-zero background."""
import numpy as np
import torch

from oracle import oracle as orc
from tests import synth


def _cam(W, H, focal, ev=(0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2)):
    intr, e = synth.camera(W, H, focal, extrinsic_vector=ev)
    return orc.camera(intr.numpy(), e.numpy()), intr, e


def make_case(name, P=4000, W=200, H=136, focal=125.0, seed=7, scale_median=0.03):
    """-> (inputs dict of CPU tensors/None + flags, cam dict, indexed bool)."""
    deg = 3
    kw = {}
    if name == "tiny":
        P, W, H, focal, scale_median = 48, 48, 32, 40.0, 0.25
        kw = dict(zmin=2, zmax=6)
    if name == "odd_size":
        W, H = 203, 131
    if name == "behind":
        kw["behind_fraction"] = 0.3
    if name.startswith("deg"):
        deg = int(name[3])
    if name == "deep_tile":      # thousands of faint splats stacked over four tiles: many staging rounds per tile
        P, W, H, focal, scale_median = 5000, 32, 32, 30.0, 0.2
        kw = dict(zmin=3, zmax=9)
    if name == "huge_splats":    # every splat's rectangle is clipped by the screen: R = P x (all tiles)
        P, W, H, focal, scale_median = 300, 150, 100, 90.0, 6.0
    if name == "one_tile":       # a 1 x 1 tile grid, narrower than one wave's pixel block
        P, W, H, focal, scale_median = 200, 9, 5, 8.0, 0.3
        kw = dict(zmin=2, zmax=6)
    if name == "p257":           # one Gaussian past a 256-block of the two-level scans
        P = 257
    if name == "p8193":          # one key past an 8192-item tile of the sorts
        P = 8193
    if name == "p12289":         # one key past a 12288-item tile of the depth sort (keys + ids + packed rectangles: two tiles)
        P = 12289
    if name == "p24577":         # three depth-sort tiles, the last holding one item
        P = 24577
    ev = (0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2)
    if name == "equal_depth":    # identity camera + one z: all depth keys tie -> the stable sort must keep id order
        ev = (0, 0, 0, 1, 0, 0, 0)
    cam, intr, ev = _cam(W, H, focal, ev)
    sc = synth.scene(P, W, H, focal, seed=seed, sh_degree=3, scale_median=scale_median, **kw)
    if name == "deep_tile":
        sc["opacities"] = sc["opacities"] * 0.02
    if name == "equal_depth":
        sc["means3D"][:, :2] *= (5.0 / sc["means3D"][:, 2])[:, None]
        sc["means3D"][:, 2] = 5.0
        sc["means3D"][::3, 2] = 5.5
    if name == "all_behind":
        sc["means3D"][:, 2] = -sc["means3D"][:, 2].abs() - 1.0
    if name == "wide_depth":     # depths over five decades (0.05 .. 6000): every digit of the depth sort keys is exercised
        g = torch.Generator().manual_seed(11)
        z = torch.exp(torch.rand(P, generator=g) * (np.log(6000.0) - np.log(0.05)) + np.log(0.05)).float()
        sc["means3D"] = sc["means3D"] * (z / sc["means3D"][:, 2])[:, None]
        sc["scales"] = sc["scales"] * (z / 7.0)[:, None]
    if name == "frustum_edge":   # exercise the 1.3*tan_fov clamp of computeCov2D
        sc["means3D"][:, 0] *= 1.6
        sc["scales"] *= 3.0
    inp = dict(bg=torch.tensor([0.2, 0.4, 0.1]), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"],
               colors_precomp=None, scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None, scale_factors=None,
               sh_indices=None, g_indices=None, degree=deg, scale_modifier=1.0, prefiltered=False, clamp_color=True)
    indexed = False
    if name == "empty":
        for k in ("means3D", "opacities", "shs", "scales", "rotations"):
            inp[k] = inp[k][:0].contiguous()
    if name == "no_clamp":
        inp["clamp_color"] = False
        inp["shs"] = inp["shs"] * 3.0          # make some colours negative
    if name == "clamp_hits":
        inp["shs"] = inp["shs"] * 3.0
    if name == "black_bg":
        inp["bg"] = torch.zeros(3)
    if name == "scale_mod":
        inp["scale_modifier"] = 1.7
    if name == "colors_precomp":
        g = torch.Generator().manual_seed(3)
        inp["shs"] = None
        inp["colors_precomp"] = torch.rand(P, 3, generator=g)
    if name == "cov_precomp":
        # unit-scale covariance from the reference's own helper semantics: Sigma = R S^2 R^T, upper triangle
        from tests.dense_ref import _rot
        Rm = _rot(sc["rotations"].double())
        Lm = Rm * sc["scales"].double()[:, None, :]
        Sg = Lm @ Lm.transpose(1, 2)
        inp["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).float().contiguous()
        inp["scales"] = None
        inp["rotations"] = None
    if name.startswith("indexed"):
        ix = synth.index_scene(sc, shs_extra=64, gs_extra=64)
        inp.update(shs=ix["shs"], scales=ix["scales"], rotations=ix["rotations"], scale_factors=ix["scale_factors"],
                   sh_indices=ix["sh_indices"], g_indices=ix["g_indices"])
        indexed = True
        if name == "indexed_deg1":
            inp["degree"] = 1
        if name == "indexed_scale_mod":
            inp["scale_modifier"] = 1.3
    return inp, cam, indexed


def oracle_forward(inp, cam):
    n = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    return orc.rasterize_forward(bg=n["bg"], means3D=n["means3D"], opacities=n["opacities"], shs=n["shs"],
                                 colors_precomp=n["colors_precomp"], scales=n["scales"], rotations=n["rotations"],
                                 cov3D_precomp=n["cov3D_precomp"], scale_factors=n["scale_factors"],
                                 sh_indices=n["sh_indices"], g_indices=n["g_indices"], degree=n["degree"],
                                 scale_modifier=n["scale_modifier"], prefiltered=n["prefiltered"],
                                 clamp_color=n["clamp_color"], **cam)


FORWARD_CASES = ["tiny", "base", "odd_size", "behind", "all_behind", "empty", "deg0", "deg1", "deg2", "no_clamp",
                 "clamp_hits", "black_bg", "scale_mod", "colors_precomp", "cov_precomp", "frustum_edge", "indexed",
                 "indexed_deg1", "indexed_scale_mod", "wide_depth", "deep_tile", "huge_splats", "one_tile", "p257", "p8193",
                 "p12289", "p24577", "equal_depth"]
