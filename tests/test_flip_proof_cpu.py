"""CPU test of tests/fullsize.py:prove_flips -- the float64 proof that a pixel on which two fp32 evaluations of the blend
disagree is a TRUE borderline (forward.cu:344-360). The second fp32 evaluation is tests/alt_blend.py (render.hip's rounding
of `power`, emulated in numpy); the scene is adversarial: hundreds of Gaussians get the opacity that puts their alpha at one
chosen pixel within half an ulp of 1/255, so flips are certain and every one of them must be explained by in-band decisions.
A corrupted pixel (what a real blend bug would produce) must NOT be explained."""
import numpy as np
import torch

from tests import alt_blend, cases, fullsize, synth
from oracle import oracle as orc


def _adversarial_state():
    W, H, focal = 80, 48, 60.0
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=(0.02, -0.01, 0.03, 0.99, 0.05, -0.02, 0.1))
    cam = orc.camera(intr.numpy(), ev.numpy())
    sc = synth.scene(700, W, H, focal, seed=21, sh_degree=3, scale_median=0.12, zmin=2, zmax=6)
    inp = dict(bg=torch.tensor([0.3, 0.1, 0.2]), means3D=sc["means3D"], opacities=sc["opacities"].clone(), shs=sc["shs"],
               colors_precomp=None, scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None, scale_factors=None,
               sh_indices=None, g_indices=None, degree=3, scale_modifier=1.0, prefiltered=False, clamp_color=True)
    st = cases.oracle_forward(inp, cam)
    A_THR = float(np.float32(1.0) / np.float32(255.0))
    rng = np.random.default_rng(3)
    tuned = 0
    for k in np.nonzero(st.radii > 0)[0]:
        mx, my = st.means2D[k]
        x, y = int(round(float(mx))) + int(rng.integers(-3, 4)), int(round(float(my))) + int(rng.integers(-3, 4))
        if not (0 <= x < W and 0 <= y < H):
            continue
        dx, dy = float(np.float32(mx) - np.float32(x)), float(np.float32(my) - np.float32(y))
        a, b, c = (float(v) for v in st.conic_opacity[k, :3])
        G = np.exp(-0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy)
        op = A_THR / G
        if 0.004 < op < 0.95:
            inp["opacities"][k, 0] = float(np.float32(op))
            tuned += 1
    assert tuned > 300
    return cases.oracle_forward(inp, cam)


def test_every_flip_between_two_fp32_evaluations_is_proven_borderline():
    st = _adversarial_state()
    u = alt_blend.forward(st)
    flipped = fullsize.flipped_pixels(u, st)
    n = int(flipped.sum())
    assert n >= 5, f"the adversarial scene produced only {n} flips: the test would be vacuous"
    proof = fullsize.prove_flips(u, st, flipped)
    assert proof["flips"] == n
    assert proof["outside_band"] == 0 and proof["oracle_outside_band"] == 0, proof
    assert proof["decisions"], "flipped pixels were matched without a single in-band decision"
    assert max(r for _, r in proof["decisions"]) <= 1.0
    # away from the flips the two evaluations agree to fp32 round-off
    ok = ~flipped
    assert np.abs(u["out_color"] - st.out_color)[:, ok].max() <= 2e-5 + 1e-4 * np.abs(st.out_color).max()


def test_the_walk_reproduces_the_oracle_on_ordinary_pixels():
    inp, cam, _ = cases.make_case("tiny")
    st = cases.oracle_forward(inp, cam)
    u = dict(out_color=st.out_color, final_T=st.final_T, n_contrib=st.n_contrib)
    every = np.ones((st.H, st.W), bool)
    proof = fullsize.prove_flips(u, st, every)
    assert proof["outside_band"] == 0 and proof["oracle_outside_band"] == 0


def test_a_wrong_pixel_is_not_explained_away():
    inp, cam, _ = cases.make_case("tiny")
    st = cases.oracle_forward(inp, cam)
    deep = np.argsort(st.n_contrib)[-3:]                     # pixels that blend the most Gaussians
    for kind in ("colour", "n_contrib", "dropped_contribution"):
        u = dict(out_color=st.out_color.copy(), final_T=st.final_T.copy(), n_contrib=st.n_contrib.copy())
        for pix in deep:
            y, x = divmod(int(pix), st.W)
            if kind == "colour":
                u["out_color"][0, y, x] += 3e-3               # < 1/255 of full scale: smaller than one borderline contribution can be
            elif kind == "n_contrib":
                u["n_contrib"][pix] -= 1
            else:
                u["final_T"][pix] *= 1.02
        flipped = np.zeros((st.H, st.W), bool)
        flipped.reshape(-1)[deep] = True
        proof = fullsize.prove_flips(u, st, flipped)
        assert proof["outside_band"] == 3, (kind, proof)
        assert proof["oracle_outside_band"] == 0
