"""Pins oracle/qat.py (the CPU restatement of the QAT getters) against the real thing: the
torch.ao.quantization.FakeQuantize modules and torch ops the reference's GaussianModel composes
(scene/gaussian_model.py:54-77, 109-118, 213-267, 1405-1414), run here on CPU tensors."""
import numpy as np
import pytest
import torch

from oracle import qat


def _torch_getters(quantization=True):
    """The reference's module set (gaussian_model.py:109-134), built from torch itself."""
    FQ = torch.ao.quantization.FakeQuantize
    m = {k: FQ(dtype=torch.qint8) for k in qat.SLOTS}
    if not quantization:
        for k in ("features_dc", "features_rest", "scaling", "scaling_factor", "rotation"):
            m[k].disable_fake_quant()
            m[k].disable_observer()
    return m


def _torch_forward(m, t, half):
    nz = torch.nn.functional.normalize
    o = {}
    o["xyz"] = t["xyz"].half().float() if half else t["xyz"]
    o["opacity"] = m["opacity"](torch.sigmoid(t["opacity"]))
    o["scales_n"] = m["scaling"](nz(torch.relu(t["scaling"])))
    o["scale_factors"] = torch.exp(m["scaling_factor"](t["scaling_factor"]))
    o["rotations"] = nz(m["rotation"](t["rotation"]))
    o["shs"] = torch.cat((m["features_dc"](t["features_dc"]), m["features_rest"](t["features_rest"])), dim=1)
    return o


def _inputs(seed, P=4000, GS=700, SHS=300, M=16, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return {"xyz": r(P, 3) * 3, "opacity": r(P, 1) * 1.5 * scale - 1, "scaling_factor": r(P, 1) * 0.6 * scale - 4.7,
            "scaling": r(GS, 3) * scale, "rotation": r(GS, 4) * scale, "features_dc": r(SHS, 1, 3) * 0.5 * scale,
            "features_rest": r(SHS, M - 1, 3) * 0.05 * scale}


def _close_quant(a, b, step, what, max_flip_frac=2e-3):
    """Equal up to fp32 noise, except for a small fraction of elements that may sit one quantisation step apart
    (an activation that differs in its last bit at a rounding tie)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    tol = 1e-6 * max(1.0, np.abs(b).max())
    flips = d > tol
    assert flips.mean() <= max_flip_frac, f"{what}: {flips.mean():.2e} of the elements differ"
    if flips.any():
        assert d[flips].max() <= step * 1.001 + tol, f"{what}: difference {d[flips].max()} exceeds one step {step}"


@pytest.mark.parametrize("quantization", [True, False])
def test_getters_follow_torch_over_several_observer_steps(quantization):
    m = _torch_getters(quantization)
    g = qat.Getters(quantization)
    for step in range(4):
        t = _inputs(10 + step, scale=1.0 + 0.3 * step)     # moving ranges: exercises the averaging branch
        ref = _torch_forward(m, t, half=quantization)
        o = g.forward(*(t[k].numpy() for k in ("xyz", "opacity", "scaling_factor", "scaling", "rotation", "features_dc",
                                                "features_rest")))
        for k in qat.SLOTS:
            mod, st = m[k], g.st[k]
            obs = mod.activation_post_process
            if quantization or k == "opacity":
                np.testing.assert_allclose(st.min_val, float(obs.min_val), rtol=3e-7, atol=1e-12, err_msg=k)
                np.testing.assert_allclose(st.max_val, float(obs.max_val), rtol=3e-7, atol=1e-12, err_msg=k)
                np.testing.assert_allclose(st.scale, float(mod.scale), rtol=3e-7, err_msg=k)
                assert st.zero_point == int(mod.zero_point), k
        assert np.array_equal(o["xyz"], ref["xyz"].numpy())
        _close_quant(o["opacity"], ref["opacity"].detach().numpy(), g.st["opacity"].scale, "opacity")
        _close_quant(o["scales_n"], ref["scales_n"].detach().numpy(), g.st["scaling"].scale, "scales_n")
        # exp(fq(.)): one step of the argument moves the value by a factor exp(step)
        a, b = np.log(o["scale_factors"]), np.log(ref["scale_factors"].detach().numpy())
        _close_quant(a, b, g.st["scaling_factor"].scale if quantization else 0.0, "log scale_factors")
        np.testing.assert_allclose(o["rotations"], ref["rotations"].detach().numpy(), rtol=0, atol=3e-7)
        _close_quant(o["shs"][:, :1], ref["shs"][:, :1].detach().numpy(), g.st["features_dc"].scale, "shs dc")
        _close_quant(o["shs"][:, 1:], ref["shs"][:, 1:].detach().numpy(), g.st["features_rest"].scale, "shs rest")


def test_identity_modules_are_bit_exact_with_torch():
    """No transcendental in front of the module (rotation, features): the restatement must equal torch bit for bit."""
    fq = torch.ao.quantization.FakeQuantize(dtype=torch.qint8)
    st = qat.FqState()
    for step in range(5):
        x = torch.randn(5000, generator=torch.Generator().manual_seed(step)) * (1 + step)
        ref = fq(x)
        qat.observe(st, x.numpy())
        y, _ = qat.fake_quant(st, x.numpy())
        assert st.scale == np.float32(fq.scale.item()) and st.zero_point == int(fq.zero_point)
        assert np.array_equal(y, ref.numpy()), step


def test_backward_follows_torch_autograd():
    m = _torch_getters(True)
    g = qat.Getters(True)
    t = _inputs(3)
    # saturate some elements so the straight-through mask matters: observe a narrow range first
    narrow = _inputs(3, scale=0.3)
    _torch_forward(m, narrow, True)
    g.forward(*(narrow[k].numpy() for k in ("xyz", "opacity", "scaling_factor", "scaling", "rotation", "features_dc",
                                            "features_rest")))
    tt = {k: v.clone().requires_grad_(k != "xyz") for k, v in t.items()}
    ref = _torch_forward(m, tt, True)
    gen = torch.Generator().manual_seed(77)
    up = {k: torch.randn(ref[k].shape, generator=gen) for k in ("opacity", "scale_factors", "scales_n", "rotations", "shs")}
    sum((ref[k] * up[k]).sum() for k in up).backward()
    o = g.forward(*(t[k].numpy() for k in ("xyz", "opacity", "scaling_factor", "scaling", "rotation", "features_dc",
                                           "features_rest")))
    r = g.backward(o, t["scaling"].numpy(), up["opacity"].numpy(), up["scale_factors"].numpy(), up["scales_n"].numpy(),
                   up["rotations"].numpy(), up["shs"].numpy())
    for k in ("opacity", "scaling_factor", "scaling", "rotation", "features_dc", "features_rest"):
        ref_g = tt[k].grad.numpy()
        ours = r[k]
        # mask flips at rounding ties move single elements; everything else agrees to fp32 accuracy
        bad = np.abs(ours - ref_g) > 2e-5 * max(1.0, np.abs(ref_g).max())
        assert bad.mean() < 2e-3, (k, bad.mean())
        assert (np.abs(ref_g) > 0).any(), k
    # masks really bite in this set-up
    assert (~o["m_rot"]).any() and (~o["m_dc"]).any()


def test_visible_rows_matches_raster_oracle_mark_visible():
    from oracle import oracle as orc
    xyz = (np.random.default_rng(0).standard_normal((3000, 3)) * 2).astype(np.float32)
    cam = orc.camera(np.array([[0.9, 0, 640], [0, 0.6, 360], [0, 0, 1]], np.float32),
                     np.array([0.05, -0.02, 0.03, 0.99, 0.1, -0.2, 0.4], np.float32))
    vis = orc.mark_visible(qat.half_round(xyz), cam["viewmatrix"], cam["projmatrix"])
    assert np.array_equal(qat.visible_rows(qat.half_round(xyz), cam["viewmatrix"]), vis.astype(bool))


def test_quantize_codes_cpu_rounding_is_bit_exact_with_torch_and_device_rounding_differs_only_at_ties():
    fq = torch.ao.quantization.FakeQuantize(dtype=torch.qint8)
    st = qat.FqState()
    x = torch.randn(400000, generator=torch.Generator().manual_seed(2)) * 2
    fq(x)
    qat.observe(st, x.numpy())
    want = torch.quantize_per_tensor(x, fq.scale, fq.zero_point, fq.dtype).int_repr().numpy()
    assert np.array_equal(qat.quantize_codes(st, x.numpy(), device_rounding=False), want)
    dev = qat.quantize_codes(st, x.numpy(), device_rounding=True)
    diff = dev.astype(np.int32) - want
    assert (diff != 0).mean() < 1e-4 and np.abs(diff).max() <= 1
