"""-m gpu randomized parity sweep: random scene sizes, image sizes (incl. single-pixel rows / columns and sizes far from
multiples of 16), focal lengths, splat sizes, SH degrees, flags and input variants, HIP vs oracle with the same bars as
tests/test_raster_gpu.py. Seeds are fixed, so a failure names a reproducible case."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc_mod
from tests import cases, fullsize, gpu_util, synth
from tests.test_raster_gpu import GRAD_TOL, _check_forward

pytestmark = pytest.mark.gpu


def _random_case(seed, large=False):
    r = np.random.default_rng(seed)
    if large:                                            # more tiles, deeper lists, several sort tiles
        P = int(r.integers(20000, 70000))
        W, H = int(r.integers(300, 700)), int(r.integers(200, 420))
    else:                                                # edge-heavy: block boundaries, 1-pixel rows / columns
        P = int(r.choice([1, 2, 63, 255, 256, 257, 1000, 4097, int(r.integers(3, 12000))]))
        W = int(r.choice([1, 15, 16, 17, 33, int(r.integers(2, 300))]))
        H = int(r.choice([1, 16, 31, int(r.integers(2, 200))]))
    focal = float(r.uniform(0.4, 1.6) * max(W, H, 8))
    scale = float(np.exp(r.uniform(np.log(0.004), np.log(0.6))))
    deg = int(r.integers(0, 4))
    variant = r.choice(["plain", "indexed", "colors_precomp", "cov_precomp"])
    behind = float(r.choice([0.0, 0.0, 0.3]))
    ev = (float(r.normal(0, 0.05)), float(r.normal(0, 0.05)), float(r.normal(0, 0.05)), 1.0, float(r.normal(0, 0.2)),
          float(r.normal(0, 0.2)), float(r.normal(0, 0.3)))
    n = np.sqrt(sum(v * v for v in ev[:4]))
    ev = tuple(v / n for v in ev[:4]) + ev[4:]
    intr, e = synth.camera(W, H, focal, extrinsic_vector=ev)
    cam = orc_mod.camera(intr.numpy(), e.numpy())
    sc = synth.scene(P, W, H, focal, seed=seed, sh_degree=3, scale_median=scale, zmin=float(r.uniform(0.3, 3)),
                     zmax=float(r.uniform(4, 40)), behind_fraction=behind)
    if r.random() < 0.3:
        sc["opacities"] = sc["opacities"] * 0.05          # faint: long blend lists
    inp = dict(bg=torch.tensor(r.random(3), dtype=torch.float32), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"],
               colors_precomp=None, scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None, scale_factors=None,
               sh_indices=None, g_indices=None, degree=deg, scale_modifier=float(r.choice([1.0, 1.0, 0.6, 1.9])),
               prefiltered=False, clamp_color=bool(r.random() < 0.7))
    indexed = False
    if variant == "indexed":
        ix = synth.index_scene(sc, seed=seed + 1, shs_extra=int(r.integers(1, 40)), gs_extra=int(r.integers(1, 40)))
        inp.update(shs=ix["shs"], scales=ix["scales"], rotations=ix["rotations"], scale_factors=ix["scale_factors"],
                   sh_indices=ix["sh_indices"], g_indices=ix["g_indices"])
        indexed = True
    elif variant == "colors_precomp":
        inp["shs"] = None
        inp["colors_precomp"] = torch.tensor(r.random((P, 3)), dtype=torch.float32)
    elif variant == "cov_precomp":
        from tests.dense_ref import _rot
        Rm = _rot(sc["rotations"].double())
        Lm = Rm * sc["scales"].double()[:, None, :]
        Sg = Lm @ Lm.transpose(1, 2)
        inp["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]],
                                           1).float().contiguous()
        inp["scales"] = inp["rotations"] = None
    return inp, cam, indexed, f"seed {seed}: P={P} {W}x{H} focal={focal:.1f} scale={scale:.3f} deg={deg} {variant}"


@pytest.mark.parametrize("seed", list(range(100, 140)) + list(range(1000, 1010)))
def test_random_case_forward_and_backward(hip, orc, seed):
    inp, cam, indexed, what = _random_case(seed, large=seed >= 1000)
    st = cases.oracle_forward(inp, cam)
    fw = gpu_util.hip_forward(inp, cam, indexed)
    try:
        _check_forward(gpu_util.unpack(fw), st, borderline_ok=seed >= 1000)
    except AssertionError as e:
        raise AssertionError(f"{what}: {e}") from e
    dL = synth.grad_image(cam["W"], cam["H"], seed=seed).numpy()
    ref = orc.rasterize_backward(st, dL)
    got = gpu_util.hip_backward(fw, dL)
    # the 1e-4 bar holds for blend lists up to about a thousand entries; both sides rebuild T by dividing out
    # (1 - alpha) entry by entry in fp32 (backward.cu:505), so the error grows with the depth of the list: the large
    # cases reach 4,500-9,700 entries per tile (screen-filling splats) and are held to 1e-3. Blend decisions that flip on a
    # rounding difference of exp() are handled as tests/fullsize.py: check_grads describes (in a 4,500-deep tile one
    # flipped pixel moved dL_drotations to 1.1e-3 once).
    fullsize.check_grads(st, gpu_util.unpack(fw), got, ref, 1e-3 if seed >= 1000 else GRAD_TOL, what)
