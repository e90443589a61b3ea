"""CPU: structural properties and edge cases of the oracle itself (it is the checker, so it is checked)."""
import numpy as np
import pytest

from tests import cases, synth


def test_higher_msb(orc):
    assert orc.get_higher_msb(8160) == 13          # 1920x1080: 120 x 68 tiles -> sort bits [0, 45)
    for n, want in [(1, 1), (2, 2), (3, 2), (4, 3), (255, 8), (256, 9), (65535, 16)]:
        assert orc.get_higher_msb(n) == want


@pytest.mark.parametrize("name", ["base", "odd_size", "behind", "indexed", "frustum_edge"])
def test_binning_invariants(orc, name):
    inp, cam, _ = cases.make_case(name)
    st = cases.oracle_forward(inp, cam)
    R = st.num_rendered
    assert R == int(st.tiles_touched.astype(np.int64).sum()) == int(st.point_offsets[-1])
    assert ((st.radii > 0) == (st.tiles_touched > 0)).all()
    order = np.argsort(st.keys_unsorted, kind="stable")           # stable sort == the reference's radix sort
    np.testing.assert_array_equal(st.keys_sorted, st.keys_unsorted[order])
    np.testing.assert_array_equal(st.point_list, st.values_unsorted[order])
    tiles = (st.keys_sorted >> np.uint64(32)).astype(np.int64)
    cnt = np.bincount(tiles, minlength=st.T)
    np.testing.assert_array_equal(st.ranges[:, 1].astype(np.int64) - st.ranges[:, 0].astype(np.int64), cnt)
    depth_bits = (st.keys_sorted & np.uint64(0xffffffff)).astype(np.uint32)
    np.testing.assert_array_equal(depth_bits, st.depths[st.point_list].view(np.uint32))
    assert ((st.final_T >= 0) & (st.final_T <= 1)).all()
    assert (st.n_contrib <= cnt.max()).all()


def test_empty_and_all_behind(orc):
    for name in ("empty", "all_behind"):
        inp, cam, _ = cases.make_case(name)
        st = cases.oracle_forward(inp, cam)
        assert st.num_rendered == 0
        # P == 0: the reference never launches and returns its zero-filled image (rasterize_points.cu:69,82);
        # P > 0 but nothing visible: every pixel is the background colour
        bg = np.zeros(3, np.float32) if name == "empty" else inp["bg"].numpy()
        np.testing.assert_allclose(st.out_color, np.broadcast_to(bg[:, None, None], st.out_color.shape))
        g = orc.rasterize_backward(st, synth.grad_image(cam["W"], cam["H"]).numpy())
        assert all(not np.any(v) for v in g.values())


def test_mark_visible_matches_radii_culling(orc):
    inp, cam, _ = cases.make_case("behind")
    vis = orc.mark_visible(inp["means3D"].numpy(), cam["viewmatrix"], cam["projmatrix"])
    st = cases.oracle_forward(inp, cam)
    assert not (st.radii[~vis] > 0).any()


def test_weighted_distance_reference_semantics(orc):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(200, 7)).astype(np.float32)
    cb = rng.normal(size=(50, 7)).astype(np.float32)
    d, i = orc.weighted_distance(x, cb)
    full = ((x[:, None].astype(np.float64) - cb[None]) ** 2).sum(-1)
    np.testing.assert_array_equal(i, full.argmin(1))
    np.testing.assert_allclose(d, full.min(1), rtol=1e-5)
    cb2 = np.concatenate([cb, cb])                                 # ties: strict '<' keeps the first
    _, i2 = orc.weighted_distance(x, cb2)
    np.testing.assert_array_equal(i2, i)
    with pytest.raises(RuntimeError, match="dimension 2"):
        orc.weighted_distance(x[0], cb)
    with pytest.raises(RuntimeError, match="same number of channels"):
        orc.weighted_distance(x, cb[:, :5])


def test_vq_update_empty_clusters_decay_toward_zero(orc):
    """compression/vq.py:34: an empty cluster's EMA target is 0/(0+eps) = 0 (reproduced on purpose)."""
    x = np.zeros((16, 3), np.float32) + 0.5
    w = np.ones(16, np.float32)
    cb = np.array([[0.5, 0.5, 0.5], [9, 9, 9]], np.float32)
    ent = np.zeros(2, np.float32)
    orc.vq_update(x, w, cb, ent, 0.8, 1e-5)
    np.testing.assert_allclose(cb[1], 0.8 * 9, rtol=1e-6)
    np.testing.assert_allclose(cb[0], 0.8 * 0.5 + 0.2 * (8.0 / (16 + 1e-5)), rtol=1e-6)
    np.testing.assert_allclose(ent, [0.2 * 16, 0.0], rtol=1e-6)


def test_splats_oracle_matches_reference_extract_rot_scale_golden():
    """oracle/splats.py vs the reference's utils/splats.py outputs (tests/golden/splats.npz)."""
    import os
    import numpy as np
    from oracle import splats
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "splats.npz"))
    rot, scaling = splats.extract_rot_scale(splats.to_full_cov(g["cov6"]))
    lam_max = (g["scaling"] ** 2).max(1, keepdims=True)
    np.testing.assert_allclose(scaling ** 2, g["scaling"] ** 2, rtol=0, atol=float(3e-6 * lam_max.max()))
    np.testing.assert_allclose(np.linalg.norm(rot, axis=1), 1.0, atol=1e-6)
    rebuilt = splats.build_covariance(rot, scaling)
    np.testing.assert_allclose(rebuilt, g["rebuilt"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(rebuilt, splats.to_full_cov(g["cov6"]), rtol=0, atol=5e-6)
