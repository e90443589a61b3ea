"""CPU: host-side logic of the drop-in Python surface (no kernels): camera set-up, argument checks, the factored
camera-pose Jacobian against the reference's closed form (golden vector), module aliasing, and the hard failure on
CPU tensors (the package has no CPU path)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

import c3dgs_amd
from c3dgs_amd import rasterizer as rz
from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_oracle_setup(orc):
    intr, ev = synth.camera(640, 480, 500.0, extrinsic_vector=(0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2))
    view, proj, campos, tfx, tfy, H, W = rz.camera_matrices(intr, ev, "cpu")
    cam = orc.camera(intr.numpy(), ev.numpy())
    np.testing.assert_array_equal(view.numpy(), cam["viewmatrix"])
    np.testing.assert_allclose(proj.numpy(), cam["projmatrix"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(campos.numpy(), cam["campos"], rtol=1e-5, atol=1e-6)
    assert (H, W) == (480, 640) and abs(tfx - cam["tan_fovx"]) < 1e-12 and abs(tfy - cam["tan_fovy"]) < 1e-12
    # W2C^T layout: m[0], m[4], m[8], m[12] is row 0 of W2C; translation in the last ROW
    np.testing.assert_allclose(view.numpy()[3, :3], [0.1, -0.05, 0.2], rtol=1e-6)
    # identity pose -> identity matrix
    np.testing.assert_array_equal(rz.quat_to_mat(torch.tensor([0., 0, 0, 1, 0, 0, 0])).numpy(), np.eye(4, dtype=np.float32))
    P = rz.getProjectionMatrix(intr)
    assert P.shape == (4, 4) and P[2, 3] == 1.0 and abs(P[3, 2].item() + 0.010001) < 1e-6   # transposed, znear .01 zfar 100


def test_camera_helpers_vs_reference_functions():
    """The package's own quat_to_mat / getProjectionMatrix / mat_to_quat / camera_matrices against tests/golden/camera.npz
    (the reference's __init__.py:19-52, 152-176 executed by make_golden.py)."""
    d = np.load(os.path.join(G, "camera.npz"), allow_pickle=False)
    for k in range(d["extrinsic_vector"].shape[0]):
        ev, intr = torch.from_numpy(d["extrinsic_vector"][k]), torch.from_numpy(d["intrinsic"][k])
        np.testing.assert_array_equal(rz.quat_to_mat(ev).numpy().view(np.uint32), d["view"][k].view(np.uint32))
        np.testing.assert_array_equal(rz.getProjectionMatrix(intr).numpy().view(np.uint32), d["P"][k].view(np.uint32))
        q = torch.stack([torch.as_tensor(v) for v in rz.mat_to_quat(torch.from_numpy(d["view"][k]).T)])
        np.testing.assert_allclose(q.numpy(), d["mat_to_quat"][k], rtol=1e-6, atol=1e-7)
        view, proj, campos, tfx, tfy, H, W = rz.camera_matrices(intr, ev, "cpu")
        np.testing.assert_array_equal(view.numpy().view(np.uint32), d["view"][k].view(np.uint32))
        np.testing.assert_allclose(proj.numpy(), d["proj"][k], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(campos.numpy(), d["campos"][k], rtol=1e-5, atol=2e-6 * max(np.abs(d["campos"][k]).max(), 1.0))
        assert (tfx, tfy, H, W) == (d["scalars"][k, 0], d["scalars"][k, 1], int(d["scalars"][k, 2]), int(d["scalars"][k, 3]))


def test_mat_to_quat_roundtrip():
    ev = torch.tensor([0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2])
    ev[:4] /= ev[:4].norm()
    m = rz.quat_to_mat(ev).T
    q = torch.stack([torch.as_tensor(v) for v in rz.mat_to_quat(m)])
    np.testing.assert_allclose(q.numpy(), ev.numpy(), atol=1e-6)


def test_camera_pose_jacobian_matches_reference_closed_form():
    d = np.load(os.path.join(G, "camgrad.npz"), allow_pickle=False)
    got = rz.camera_pose_jacobian_sum(torch.from_numpy(d["means3D"]), torch.from_numpy(d["intrinsic"]),
                                      torch.from_numpy(d["extrinsic_vector"]), torch.from_numpy(d["du"]),
                                      torch.from_numpy(d["dv"]))
    np.testing.assert_allclose(got.numpy(), d["grad_mat"], rtol=2e-4, atol=1e-4)


def test_reference_pose_closed_form_is_not_the_projection(orc):
    """VERDICT r1 asked to pin the kernels' NDC projection with the polynomials numU / den of the reference's camera-pose
    closed form (__init__.py:674-788). They cannot: numU / den is not p_hom.x / p_hom.w of the projection the reference's own
    matrices (quat_to_mat @ getProjectionMatrix, golden camera.npz) define -- documented here so nobody relies on it."""
    d = np.load(os.path.join(G, "camgrad.npz"), allow_pickle=False)
    m, intr, ev = d["means3D"].astype(np.float64), d["intrinsic"], d["extrinsic_vector"]
    X, Y, Z = m[:, 0], m[:, 1], m[:, 2]
    qx, qy, qz, qw, tx, ty, tz = [float(v) for v in ev]
    Xs, Ys = X / math.tan(float(intr[0, 0]) / 2), Y / math.tan(float(intr[1, 1]) / 2)
    numU = (Xs * (2 * qx ** 2 - 4 * qx * qy - 2 * qz ** 2 + 1) + Ys * (-2 * qw * qz + 2 * qx * qy)
            + Z * (2.000200020002 * (qw * qy + qx * qz) + tx) - 0.02000200020002 * (qw * qy + qx * qz))
    den = (Xs * (-2 * qw * qy + 2 * qx * qz) + Ys * (2 * qw * qx + 2 * qy * qz)
           + Z * (-4.000400040004 * qx * qy + tz + 1.000100010001) + 0.04000400040004 * qx * qy - 0.01000100010001)
    cam = orc.camera(intr, ev)
    hom = np.concatenate([m, np.ones((len(m), 1))], 1) @ cam["projmatrix"].astype(np.float64).reshape(4, 4)
    ndc_x = hom[:, 0] / (hom[:, 3] + 1e-7)
    assert np.abs(numU / den - ndc_x).max() > 0.05


def test_argument_checks_and_no_cpu_path():
    intr, ev = synth.camera(64, 48, 40.0)
    rs = c3dgs_amd.GaussianRasterizationSettings(intr, ev, torch.zeros(3), 1.0, 3, False, False, True)
    rast = c3dgs_amd.GaussianRasterizer(rs)
    sc = synth.scene(10, 64, 48, 40.0)
    m = sc["means3D"]
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(m, m, sc["opacities"], scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(m, m, sc["opacities"], shs=sc["shs"], colors_precomp=m, scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        rast(m, m, sc["opacities"], shs=sc["shs"], scales=sc["scales"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        rast(m, m, sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=torch.zeros(10, 6))
    # CPU tensors: the product refuses instead of falling back
    with pytest.raises(RuntimeError, match="GPU tensor"):
        rast(m, m, sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], extrinsic_vector=ev)
    with pytest.raises(RuntimeError, match="GPU"):
        c3dgs_amd.weightedDistance(torch.zeros(4, 6), torch.zeros(2, 6))
    with pytest.raises(RuntimeError, match="dimension 2"):
        c3dgs_amd.weightedDistance(torch.zeros(4), torch.zeros(2, 6))
    with pytest.raises(RuntimeError, match="same number of channels"):
        c3dgs_amd.weightedDistance(torch.zeros(4, 5), torch.zeros(2, 6))
    idx = c3dgs_amd.GaussianRasterizerIndexed(rs, optimize_camera=True)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        idx(m, m, sc["opacities"], torch.zeros(10, dtype=torch.long), torch.zeros(10, dtype=torch.long))
    # markVisible: the one-launch pose path is for GPU tensors only; CPU positions take the general path, which refuses them
    from c3dgs_amd import rasterizer as rz
    assert rz._mark_visible_from_pose(m, ev) is None
    assert rz._mark_visible_from_pose(m, torch.eye(4)) is None
    with pytest.raises(RuntimeError, match="GPU tensor"):
        rast.markVisible(m, extrinsic_vector=ev)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        idx.markVisible(m, extrinsic_vector=ev)


def test_join_features_and_settings():
    feats = torch.arange(12.).reshape(6, 2)
    keep = torch.tensor([True, False, False, True, False, False])
    cb = torch.tensor([[100., 100.], [200., 200.]])
    comp, idx = c3dgs_amd.join_features(feats, keep, cb, torch.tensor([0, 1, 1, 0]))
    assert comp.shape == (4, 2) and idx.tolist() == [2, 0, 1, 3, 1, 0]
    torch.testing.assert_close(comp[idx][keep], feats[keep])
    cs = c3dgs_amd.CompressionSettings(4096, 0.0, None, 0.9, 100, 0.8, 2 ** 18)
    assert cs.codebook_size == 4096 and cs.importance_include is None


def test_compress_plumbing_with_a_stub_quantiser(monkeypatch):
    """compress_gaussians end to end on CPU with vq_features replaced by a stub (nearest of K fixed rows): pruning, the
    relative keep thresholds, the [codebook ; kept rows] tables and the row numbers handed to the model."""
    from c3dgs_amd import vq as vqm
    g = torch.Generator().manual_seed(2)
    P = 400
    sh = torch.randn(P, 4, 3, generator=g)
    cov = torch.rand(P, 6, generator=g) + 0.1
    cimp, gimp = torch.rand(P, generator=g), torch.rand(P, generator=g)
    calls = []

    def stub(features, importance, codebook_size, vq_chunk, steps, scale_normalize=False, silent=False, group=None):
        calls.append((features.shape, codebook_size, vq_chunk, steps, scale_normalize))
        cb = features[:codebook_size].clone()
        return cb, torch.cdist(features, cb).argmin(1)
    monkeypatch.setattr(vqm, "vq_features", stub)

    class Model:
        def __init__(self):
            self.sh, self.cov = sh.clone(), cov.clone()
        get_features = property(lambda self: self.sh)

        def get_normalized_covariance(self, strip_sym=True):
            return self.cov

        def mask_splats(self, m):
            self.sh, self.cov = self.sh[m], self.cov[m]

        def set_color_indexed(self, table, rows):
            self.color = (table, rows)

        def set_gaussian_indexed(self, rot, scale, rows):
            self.gauss = (rot, scale, rows)
    m = Model()
    cc = vqm.CompressionSettings(16, 0.0, None, 0.9, 7, 0.8, 64)
    gc = vqm.CompressionSettings(8, 0.0, None, 0.75, 9, 0.8, 32)
    vqm.compress_gaussians(m, cimp.clone(), gimp.clone(), cc, gc, color_compress_non_dir=False, prune_threshold=0.05, silent=True,
                           extract_rot_scale=lambda full: (full[:, 0], full[:, 1]), to_full_cov=lambda c6: torch.stack([c6, 2 * c6], 1))
    alive = cimp > 0.05
    n = int(alive.sum())
    assert m.sh.shape[0] == n and calls[0][1:] == (16, 64, 7, False) and calls[1][1:] == (8, 32, 9, True)
    table, rows = m.color
    keep = cimp[alive] > torch.quantile(cimp[alive], 0.9)
    assert table.shape == (16 + int(keep.sum()), 3, 3) and rows.shape == (n,) and rows.dtype == torch.long
    torch.testing.assert_close(table[rows][keep], sh[alive][keep][:, 1:])             # kept rows verbatim (DC excluded)
    assert int(rows[~keep].max()) < 16 and (rows[keep] == 16 + torch.arange(int(keep.sum()))).all()
    rot, scale, grows = m.gauss
    gkeep = gimp[alive] > torch.quantile(gimp[alive], 0.75)
    torch.testing.assert_close(rot[grows][gkeep], cov[alive][gkeep])
    torch.testing.assert_close(scale[grows][gkeep], 2 * cov[alive][gkeep])
    # a job without settings is skipped, and with everything kept the table is just the rows themselves
    m2 = Model()
    vqm.compress_gaussians(m2, cimp.clone(), gimp.clone(), vqm.CompressionSettings(16, 0.0, -1.0, 0.9, 7, 0.8, 64), None,
                           color_compress_non_dir=True, prune_threshold=-1.0, silent=True)
    assert not hasattr(m2, "gauss") and m2.color[0].shape == (P, 4, 3) and (m2.color[1] == torch.arange(P)).all()


def test_install_as_reference_modules():
    c3dgs_amd.install_as_reference_modules()
    import diff_gaussian_rasterization_no_camera as dgr
    from weighted_distance._C import weightedDistance
    assert dgr.GaussianRasterizerIndexed is c3dgs_amd.GaussianRasterizerIndexed and callable(weightedDistance)
    for n in ("diff_gaussian_rasterization_no_camera", "diff_gaussian_rasterization", "diff_gaussian_rasterization_camera",
              "weighted_distance", "weighted_distance._C"):
        sys.modules.pop(n, None)


def test_batch_draws_equal_reference_stream():
    """vq.py:69 draws `torch.randint(0, N, [chunk])` once per step on the CPU default generator; so do we."""
    from c3dgs_amd.vq import _BatchDraws
    torch.manual_seed(3)
    ref = [torch.randint(low=0, high=1000, size=[37]) for _ in range(11)]
    torch.manual_seed(3)
    d = _BatchDraws(1000, 37, 11, "cpu")
    assert all(torch.equal(a, d.next_batch()) for a in ref)


@pytest.mark.parametrize("seed,N,sizes", [(3, 1000, [37, 1, 700, 623, 624, 625, 5000]), (0, 4_500_000, [2 ** 16, 3, 2 ** 12]),
                                          (12345, 2 ** 28 - 1, [1000, 1249])])
def test_mt19937_fill_continues_the_torch_cpu_generator(seed, N, sizes):
    """csrc/draws.hip (host part): raw words % N == torch.randint's values, from any position inside a 624-word block,
    and the written-back state makes torch continue exactly where it would have been."""
    import ctypes as C
    from c3dgs_amd import _lib, vq
    L = _lib.lib()
    torch.manual_seed(seed)
    torch.rand(5)                                        # start somewhere inside a block
    st0 = torch.get_rng_state()
    ref = [torch.randint(low=0, high=N, size=[n]) for n in sizes]
    after = torch.rand(7)
    torch.set_rng_state(st0)
    w = st0.clone().view(torch.int64)
    key = w[vq._MT_STATE:vq._MT_STATE + 624].to(torch.int32).contiguous()
    left, nxt = C.c_int64(int(w.view(torch.int32)[vq._MT_LEFT])), C.c_int64(int(w[vq._MT_NEXT]))
    for n, r in zip(sizes, ref):
        out = torch.empty(n, dtype=torch.int32)
        _lib.check(L.c3dgs_mt19937_fill(key.data_ptr(), C.byref(left), C.byref(nxt), out.data_ptr(), n))
        got = (out.to(torch.int64) & 0xffffffff) % N
        assert torch.equal(got, r)
    w.view(torch.int32)[vq._MT_LEFT] = left.value
    w[vq._MT_NEXT] = nxt.value
    w[vq._MT_STATE:vq._MT_STATE + 624] = key.to(torch.int64) & 0xffffffff
    torch.set_rng_state(w.view(torch.uint8))
    assert torch.equal(torch.rand(7), after)
    # a freshly seeded generator (left == 1, nothing drawn yet)
    torch.manual_seed(seed + 1)
    w = torch.get_rng_state().view(torch.int64)
    assert int(w.view(torch.int32)[vq._MT_LEFT]) == 1
    key = w[vq._MT_STATE:vq._MT_STATE + 624].to(torch.int32).contiguous()
    left, nxt = C.c_int64(1), C.c_int64(int(w[vq._MT_NEXT]))
    out = torch.empty(10, dtype=torch.int32)
    _lib.check(L.c3dgs_mt19937_fill(key.data_ptr(), C.byref(left), C.byref(nxt), out.data_ptr(), 10))
    assert torch.equal((out.to(torch.int64) & 0xffffffff) % N, torch.randint(low=0, high=N, size=[10]))
    bad = C.c_int64(700)
    assert L.c3dgs_mt19937_fill(key.data_ptr(), C.byref(bad), C.byref(nxt), out.data_ptr(), 10) != 0


def test_scratch_owner_dies_without_the_cycle_collector():
    """The resize callbacks close over their owner (a reference cycle); release() must break it, otherwise every
    call's scratch (hundreds of MB) stays allocated until gc runs and the caching allocator keeps growing."""
    import gc
    import weakref
    gc.disable()
    try:
        s = rz._Scratch(torch.device("cpu"))
        cb = s.callback("geom")
        assert cb(None, 64) != 0
        wt = weakref.ref(s.bufs["geom"])
        bufs = s.release()
        ws = weakref.ref(s)
        del s, cb
        assert ws() is None                      # freed by reference counting alone
        assert wt() is not None and bufs["geom"].numel() == 64
        del bufs
        assert wt() is None
    finally:
        gc.enable()


def test_covariance_written_out_equals_matmul_form_and_mask_gather_equals_boolean_index():
    """model._covariance replaces the reference's batched `L @ L.transpose(1, 2)` (gaussian_model.py:55-64) by the six
    written-out products; model._MaskGather replaces `t[mask]` (gaussian_model.py:851-862) -- same values, same gradients."""
    from c3dgs_amd import model
    g = torch.Generator().manual_seed(4)
    s = torch.rand(500, 3, generator=g) + 0.1
    q = torch.randn(500, 4, generator=g)
    r = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                     1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                     1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (1.3 * s)[:, None, :]
    full = L @ L.transpose(1, 2)
    assert torch.allclose(model._covariance(s, 1.3, q, strip_sym=False), full, rtol=1e-5, atol=1e-6)
    sym = model._covariance(s, 1.3, q)
    assert torch.allclose(sym, torch.stack([full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2],
                                            full[:, 2, 2]], 1), rtol=1e-5, atol=1e-6)
    t1 = torch.randn(500, 4, 3, generator=g, requires_grad=True)
    t2 = t1.detach().clone().requires_grad_(True)
    mask = torch.rand(500, generator=g) < 0.4
    a = model._MaskGather.apply(t1, mask.nonzero().squeeze(1))
    b = t2[mask]
    assert torch.equal(a, b)
    wgt = torch.randn(a.shape, generator=g)
    (a * wgt).sum().backward(); (b * wgt).sum().backward()
    assert torch.equal(t1.grad, t2.grad)


def test_pipeline_parameter_defaults_and_lr_schedule():
    """arguments/__init__.py:85-136 defaults; utils/general_utils.py:32-65 schedule (values computed from the formula)."""
    import math
    from c3dgs_amd import model, pipeline
    o, c = pipeline.OptimizationParams(), pipeline.CompressionParams(finetune_iterations=7)
    assert (o.position_lr_init, o.position_lr_final, o.position_lr_max_steps, o.feature_lr, o.lambda_dssim) == \
        (0.00016, 0.0000016, 30_000, 0.0025, 0.2)
    assert (c.color_codebook_size, c.gaussian_batch_size, c.gaussian_cluster_iterations, c.finetune_iterations) == \
        (4096, 2 ** 20, 800, 7)
    with pytest.raises(TypeError):
        pipeline.CompressionParams(no_such_parameter=1)
    f = model.get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30_000)
    assert f(0) == pytest.approx(1.6e-4) and f(30_000) == pytest.approx(1.6e-6) and f(10 ** 6) == pytest.approx(1.6e-6)
    assert f(15_000) == pytest.approx(math.sqrt(1.6e-4 * 1.6e-6)) and f(-1) == 0.0
    assert model.get_expon_lr_func(0.0, 0.0)(5) == 0.0
    g = model.get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    assert g(0) == pytest.approx(1e-2 * 0.1)
    assert g(50) == pytest.approx((0.1 + 0.9 * math.sin(0.25 * math.pi)) * math.exp(math.log(1e-2) * 0.95 + math.log(1e-4) * 0.05))


def test_vq_features_host_logic_replays_the_reference_from_a_seed_alone():
    """c3dgs_amd.vq.vq_features (host logic; compute injected from the oracle, the product has no CPU path) against
    tests/golden/vq_config0.npz = the reference's own vq_features run from torch.manual_seed(0) on BASELINE.json configs[0]
    exactly: same kaiming_uniform_ consumption in VectorQuantize.__init__, same randint batches, same result."""
    import numpy as np
    from c3dgs_amd import vq
    from tests import vq_fixture
    from tests.oracle_ops import OracleOps
    fx = vq_fixture.load("vq_config0.npz")
    vq_fixture.seed_like_reference(fx)
    cb, idx = vq.vq_features(fx["features"], fx["importance"], fx["K"], fx["chunk"], fx["steps"], silent=True,
                             init_rand=fx["init_rand"], ops=OracleOps)
    np.testing.assert_allclose(cb.numpy(), fx["codebook"], rtol=1e-4, atol=2e-7)
    assert (idx.numpy() == fx["indices"]).mean() >= 0.999


def test_matrix_pose_camera_setup_vs_reference_golden():
    """The 4x4-`extrinsic` path of camera_matrices (the sibling packages' API, diff_gaussian_rasterization/__init__.py:129-135)
    against tests/golden/camera.npz: `extrinsic @ getProjectionMatrix(intrinsic)` and `extrinsic.inverse()[3, :3]` are the
    reference's own torch operations: the product reproduces the golden bit for bit on CPU tensors, the inverse to a few ulps."""
    from c3dgs_amd import rasterizer_matrix as rm
    d = np.load(os.path.join(G, "camera.npz"), allow_pickle=False)
    for k in range(d["extrinsic_vector"].shape[0]):
        intr, extrinsic = torch.from_numpy(d["intrinsic"][k]), torch.from_numpy(d["view"][k])
        view, proj, campos, tfx, tfy, H, W = rz.camera_matrices(intr, extrinsic, "cpu")
        np.testing.assert_array_equal(view.numpy().view(np.uint32), d["view"][k].view(np.uint32))
        np.testing.assert_array_equal(proj.numpy().view(np.uint32), d["proj"][k].view(np.uint32))
        # (the LU inverse sees a contiguous copy here, a transposed view in the generator: a few ulps)
        np.testing.assert_allclose(campos.numpy(), d["campos"][k], rtol=2e-6, atol=1e-6 * max(np.abs(d["campos"][k]).max(), 1.0))
        assert (tfx, tfy, H, W) == (d["scalars"][k, 0], d["scalars"][k, 1], int(d["scalars"][k, 2]), int(d["scalars"][k, 3]))
    assert rm.GaussianRasterizationSettings._fields == ("intrinsic", "bg", "scale_modifier", "sh_degree", "prefiltered", "debug", "clamp_color")
    with pytest.raises(RuntimeError, match="4x4"):
        rz.camera_matrices(torch.from_numpy(d["intrinsic"][0]), torch.zeros(3, 4), "cpu")
