"""CPU: the oracle's forward colours and analytic backward (K10-K12, incl. indexed scatter, background term,
clamping, precomputed variants) against float64 torch.autograd of an independent dense re-derivation."""
import numpy as np
import pytest
import torch

from tests import cases, dense_ref, synth

DD = torch.float64
TOL = 2e-5


def _leaves(inp):
    P = inp["means3D"].shape[0]
    lv = dict(means3D=inp["means3D"].to(DD).requires_grad_(), means2D=torch.zeros(P, 3, dtype=DD, requires_grad=True),
              opacities=inp["opacities"].to(DD).requires_grad_())
    for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "scale_factors"):
        if inp.get(k) is not None:
            lv[k] = inp[k].to(DD).requires_grad_()
    return lv


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", ["tiny", "tiny_indexed", "tiny_cov_precomp", "tiny_colors", "tiny_noclamp", "tiny_deg1"])
def test_oracle_vs_float64_autograd(orc, name):
    base = dict(P=40, W=48, H=32, focal=40.0, scale_median=0.25)
    variant = {"tiny": "base", "tiny_indexed": "indexed", "tiny_cov_precomp": "cov_precomp", "tiny_colors": "colors_precomp",
               "tiny_noclamp": "no_clamp", "tiny_deg1": "deg1"}[name]
    inp, cam, indexed = cases.make_case(variant, seed=3, **base)
    inp["means3D"][:, 2] = inp["means3D"][:, 2] * 0.4 + 1.5        # bring everything close so tiles overlap
    st = cases.oracle_forward(inp, cam)
    assert st.num_rendered > 0
    dL = synth.grad_image(cam["W"], cam["H"]).numpy()
    g = orc.rasterize_backward(st, dL)
    lv = _leaves(inp)
    img = dense_ref.dense_render(st, lv)
    assert np.abs(img.detach().numpy() - st.out_color).max() < 5e-6
    (img * torch.tensor(dL, dtype=DD)).sum().backward()
    assert rel(g["dL_dmeans3D"], lv["means3D"].grad.numpy()) < TOL
    assert rel(g["dL_dmeans2D"], lv["means2D"].grad.numpy()) < TOL
    assert rel(g["dL_dopacity"], lv["opacities"].grad.numpy().reshape(-1, 1)) < TOL
    if "shs" in lv:
        assert rel(g["dL_dsh"], lv["shs"].grad.numpy()) < TOL
    if "colors_precomp" in lv:
        assert rel(g["dL_dcolors"], lv["colors_precomp"].grad.numpy()) < TOL
    if "cov3D_precomp" in lv:
        assert rel(g["dL_dcov3D"], lv["cov3D_precomp"].grad.numpy()) < TOL
    if "scales" in lv:
        assert rel(g["dL_dscales"], lv["scales"].grad.numpy()) < TOL
        assert rel(g["dL_drotations"], lv["rotations"].grad.numpy()) < TOL
    if "scale_factors" in lv:
        assert rel(g["dL_dscale_factors"], lv["scale_factors"].grad.numpy().reshape(-1, 1)) < TOL
