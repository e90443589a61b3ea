"""-m gpu: the drop-in Python surface (reference diff_gaussian_rasterization_no_camera/__init__.py) end to end
through torch.autograd, for both module classes."""
import numpy as np
import pytest
import torch

from tests import cases, gpu_util, synth

pytestmark = pytest.mark.gpu


def _settings(hip, intr, ev, clamp=True, bg=(0.2, 0.4, 0.1), deg=3):
    return hip.GaussianRasterizationSettings(intrinsic=intr.cuda(), extrinsic_vector=ev.cuda(),
                                             bg=torch.tensor(bg, device="cuda"), scale_modifier=1.0, sh_degree=deg,
                                             prefiltered=False, debug=False, clamp_color=clamp)


def test_rasterizer_module_autograd(hip, orc):
    W, H, focal = 200, 136, 125.0
    ev_t = (0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2)
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=ev_t)
    inp, cam, _ = cases.make_case("base")
    rs = _settings(hip, intr, ev)
    rast = hip.GaussianRasterizer(rs)
    leaves = {k: inp[k].cuda().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    visible = rast.markVisible(leaves["means3D"], extrinsic_vector=ev.cuda())
    assert visible.dtype == torch.bool and visible.all()
    color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                        scales=leaves["scales"], rotations=leaves["rotations"], extrinsic_vector=ev.cuda())
    assert color.shape == (3, H, W) and radii.dtype == torch.int32 and radii.shape == (4000,)
    dL = synth.grad_image(W, H)
    (color * dL.cuda()).sum().backward()
    # the module builds its own camera matrices on the host; they must equal the oracle's camera set-up
    st = cases.oracle_forward(inp, cam)
    np.testing.assert_array_equal(radii.cpu().numpy(), st.radii)
    ref = orc.rasterize_backward(st, dL.numpy())
    pairs = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", shs="dL_dsh", scales="dL_dscales", rotations="dL_drotations")
    for k, r in pairs.items():
        assert gpu_util.rel_inf(leaves[k].grad.cpu().numpy(), ref[r].reshape(leaves[k].shape)) <= 1e-4, k
    assert gpu_util.rel_inf(means2D.grad.cpu().numpy(), ref["dL_dmeans2D"]) <= 1e-4


def test_indexed_module_autograd_and_camera_grad(hip, orc):
    W, H, focal = 200, 136, 125.0
    ev_t = (0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2)
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=ev_t)
    inp, cam, _ = cases.make_case("indexed")
    rs = _settings(hip, intr, ev)
    for optimize_camera in (False, True):
        rast = hip.GaussianRasterizerIndexed(rs, optimize_camera=optimize_camera)
        leaves = {k: inp[k].cuda().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "scale_factors", "rotations")}
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
        evg = ev.cuda().requires_grad_(optimize_camera)
        color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                            sh_indices=inp["sh_indices"].cuda(), g_indices=inp["g_indices"].cuda(), shs=leaves["shs"],
                            scales=leaves["scales"], scale_factors=leaves["scale_factors"], rotations=leaves["rotations"],
                            extrinsic_vector=evg)
        dL = synth.grad_image(W, H)
        (color * dL.cuda()).sum().backward()
        st = cases.oracle_forward(inp, cam)
        ref = orc.rasterize_backward(st, dL.numpy())
        pairs = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", shs="dL_dsh", scales="dL_dscales",
                     scale_factors="dL_dscale_factors", rotations="dL_drotations")
        for k, r in pairs.items():
            assert gpu_util.rel_inf(leaves[k].grad.cpu().numpy(), ref[r].reshape(leaves[k].shape)) <= 1e-4, k
        if optimize_camera:
            assert evg.grad is not None and evg.grad.shape == (7,) and torch.isfinite(evg.grad).all()
        else:
            assert evg.grad is None


def test_no_grad_render_and_error_paths(hip):
    W, H, focal = 200, 136, 125.0
    intr, ev = synth.camera(W, H, focal)
    inp, cam, _ = cases.make_case("base")
    rast = hip.GaussianRasterizer(_settings(hip, intr, ev))
    m = inp["means3D"].cuda()
    with torch.no_grad():
        color, radii = rast(means3D=m, means2D=torch.zeros_like(m), opacities=inp["opacities"].cuda(), shs=inp["shs"].cuda(),
                            scales=inp["scales"].cuda(), rotations=inp["rotations"].cuda(), extrinsic_vector=ev.cuda())
    assert torch.isfinite(color).all()
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=inp["opacities"].cuda(), scales=inp["scales"].cuda(), rotations=inp["rotations"].cuda())
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        rast(means3D=m, means2D=m, opacities=inp["opacities"].cuda(), shs=inp["shs"].cuda())
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        rast(means3D=m[:, :2].contiguous(), means2D=m, opacities=inp["opacities"].cuda(), shs=inp["shs"].cuda(),
             scales=inp["scales"].cuda(), rotations=inp["rotations"].cuda(), extrinsic_vector=ev.cuda())
    with pytest.raises(RuntimeError, match="GPU tensor"):
        rast(means3D=inp["means3D"], means2D=m, opacities=inp["opacities"].cuda(), shs=inp["shs"].cuda(),
             scales=inp["scales"].cuda(), rotations=inp["rotations"].cuda(), extrinsic_vector=ev.cuda())


def test_sensitivity_pass_on_gpu(hip, orc):
    """Row N2: calc_importance with the real renderer (non-indexed rasterizer, cov3D_precomp, clamp_color=False) and the
    fused loss; checked against the oracle's gradients for one camera (abs of the same backward)."""
    from types import SimpleNamespace
    from c3dgs_amd import sensitivity
    W, H, focal = 200, 136, 125.0
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=(0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2))
    inp, cam, _ = cases.make_case("cov_precomp")
    inp["clamp_color"] = False
    inp["bg"] = torch.zeros(3)
    dc = inp["shs"][:, :1].cuda().clone().requires_grad_()
    rest = inp["shs"][:, 1:].cuda().clone().requires_grad_()
    sf = torch.full((4000, 1), 1.3)
    cov_unit = (inp["cov3D_precomp"] / sf.square()).cuda().requires_grad_()
    render = sensitivity.make_render_fn(inp["means3D"].cuda(), inp["opacities"].cuda(), dc, rest, cov_unit, sf.cuda())
    camera = SimpleNamespace(intrinsic=intr.cuda(), extrinsic_vector=ev.cuda(), original_image=None)
    imp, cg = sensitivity.calc_importance(render, dc, rest, cov_unit, [camera], use_gt=False)
    st = cases.oracle_forward(inp, cam)
    ref = orc.rasterize_backward(st, np.ones((3, H, W), np.float32))
    npx = H * W
    want_imp = np.abs(ref["dL_dsh"]).reshape(4000, -1) / npx
    want_cg = np.abs(ref["dL_dcov3D"]) * (1.3 ** 2) / npx
    assert gpu_util.rel_inf(imp.cpu().numpy(), want_imp) <= 2e-4
    assert gpu_util.rel_inf(cg.cpu().numpy(), want_cg) <= 2e-4


def test_steady_state_makes_no_device_allocations(hip):
    """Scratch goes back to the caching allocator when a call returns (no reference cycles holding it): after a
    warm-up, repeated forward+backward calls must not hipMalloc."""
    import gc
    from c3dgs_amd import rasterizer as rz
    inp, cam, indexed = cases.make_case("base")
    dL = synth.grad_image(cam["W"], cam["H"]).numpy()
    gc.collect()
    gc.disable()
    try:
        for _ in range(4):
            fw = gpu_util.hip_forward(inp, cam, indexed)
            gpu_util.hip_backward(fw, dL)
            del fw
        n0 = torch.cuda.memory_stats()["num_device_alloc"]
        r0 = torch.cuda.memory_reserved()
        for _ in range(12):
            fw = gpu_util.hip_forward(inp, cam, indexed)
            gpu_util.hip_backward(fw, dL)
            del fw
        assert torch.cuda.memory_stats()["num_device_alloc"] == n0
        assert torch.cuda.memory_reserved() == r0
    finally:
        gc.enable()


def test_module_path_makes_no_device_allocations_in_steady_state(hip):
    """The autograd module path bench.py drives (markVisible + GaussianRasterizerIndexed + backward): nothing a step
    allocates may survive it through a reference cycle."""
    import gc
    sc = synth.scene(20000, 640, 360, 400.0, seed=3, scale_median=0.02)
    ix = {k: v.cuda() for k, v in synth.index_scene(sc, shs_extra=64, gs_extra=64).items()}
    intr, ev = synth.camera(640, 360, 400.0)
    rast = hip.GaussianRasterizerIndexed(_settings(hip, intr, ev), optimize_camera=True)
    leaves = {k: ix[k].clone().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "scale_factors", "rotations")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    evd = ev.cuda().requires_grad_()
    dL = synth.grad_image(640, 360).cuda()

    def step():
        for v in leaves.values():
            v.grad = None
        means2D.grad = None
        evd.grad = None
        rast.markVisible(leaves["means3D"], extrinsic_vector=evd)
        color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                            sh_indices=ix["sh_indices"], g_indices=ix["g_indices"], shs=leaves["shs"], scales=leaves["scales"],
                            scale_factors=leaves["scale_factors"], rotations=leaves["rotations"], extrinsic_vector=evd)
        torch.autograd.backward(color, dL)

    gc.collect()
    gc.disable()
    try:
        for _ in range(5):
            step()
        n0, r0 = torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_reserved()
        for _ in range(12):
            step()
        assert torch.cuda.memory_stats()["num_device_alloc"] == n0
        assert torch.cuda.memory_reserved() == r0
    finally:
        gc.enable()


def test_device_camera_setup_vs_reference_golden(hip):
    """c3dgs_camera_from_pose (the matrices the rendering path really uses) against tests/golden/camera.npz, i.e. the
    reference's quat_to_mat / getProjectionMatrix / camera set-up executed by make_golden.py: view bit-exact, `view @ P`
    and `inverse(view)[3, :3]` to fp32 round-off; and bit-equal to the package's host restatement."""
    import os
    from c3dgs_amd import rasterizer as rz
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "camera.npz"), allow_pickle=False)
    for k in range(d["extrinsic_vector"].shape[0]):
        ev, intr = torch.from_numpy(d["extrinsic_vector"][k]), torch.from_numpy(d["intrinsic"][k])
        view, proj, campos, tfx, tfy, H, W = rz.camera_matrices(intr.cuda(), ev.cuda(), "cuda")
        assert view.is_cuda and view.shape == (4, 4) and proj.shape == (4, 4) and campos.shape == (3,)
        np.testing.assert_array_equal(view.cpu().numpy().view(np.uint32), d["view"][k].view(np.uint32))
        np.testing.assert_allclose(proj.cpu().numpy(), d["proj"][k], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(campos.cpu().numpy(), d["campos"][k], rtol=1e-5, atol=2e-6 * max(np.abs(d["campos"][k]).max(), 1.0))
        assert (tfx, tfy, H, W) == (d["scalars"][k, 0], d["scalars"][k, 1], int(d["scalars"][k, 2]), int(d["scalars"][k, 3]))
        hv, hp, hc = rz.camera_matrices(intr, ev, "cpu")[:3]
        np.testing.assert_array_equal(view.cpu().numpy().view(np.uint32), hv.numpy().view(np.uint32))
        np.testing.assert_array_equal(proj.cpu().numpy().view(np.uint32), hp.numpy().view(np.uint32))
        np.testing.assert_array_equal(campos.cpu().numpy().view(np.uint32), hc.numpy().view(np.uint32))


def _render_indexed(hip, rs, ix, evd):
    rast = hip.GaussianRasterizerIndexed(rs, optimize_camera=True)
    with torch.no_grad():
        vis = rast.markVisible(ix["means3D"], extrinsic_vector=evd)
        color, radii = rast(means3D=ix["means3D"], means2D=torch.zeros_like(ix["means3D"]), opacities=ix["opacities"],
                            sh_indices=ix["sh_indices"], g_indices=ix["g_indices"], shs=ix["shs"], scales=ix["scales"],
                            scale_factors=ix["scale_factors"], rotations=ix["rotations"], extrinsic_vector=evd)
    return color.clone(), radii.clone(), vis.clone()


def test_pose_written_behind_autograds_back_is_never_stale(hip):
    """The camera matrices follow the pose tensor's CURRENT values whatever wrote them: a `.data` write (the reference's
    own style, compression/vq.py:46), the package's fused Adam (raw-pointer writes, no version bump) and an in-place torch
    op all change the very next render, and that render equals one with a fresh tensor holding the same values."""
    from c3dgs_amd import optim
    W, H, focal = 320, 200, 200.0
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=(0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2))
    sc = synth.scene(6000, W, H, focal, seed=5, scale_median=0.03, behind_fraction=0.2)
    ix = {k: v.cuda() for k, v in synth.index_scene(sc, shs_extra=64, gs_extra=64).items()}
    rs = _settings(hip, intr, ev)
    pose = ev.cuda().clone().requires_grad_()
    c0, r0, v0 = _render_indexed(hip, rs, ix, pose)
    # 1. `.data` write
    pose.data[4:] += torch.tensor([0.4, -0.3, -3.5], device="cuda")
    c1, r1, v1 = _render_indexed(hip, rs, ix, pose)
    cf, rf, vf = _render_indexed(hip, rs, ix, pose.detach().clone())
    assert not torch.equal(c0, c1) and not torch.equal(v0, v1)
    assert torch.equal(c1, cf) and torch.equal(r1, rf) and torch.equal(v1, vf)
    # 2. the fused Adam writes through raw pointers
    opt = optim.Adam([{"params": [pose], "lr": 0.05}], lr=0.0, eps=1e-15)
    pose.grad = torch.tensor([0.1, -0.2, 0.3, 0.0, 1.0, -1.0, 2.0], device="cuda")
    before = pose.detach().clone()
    opt.step()
    assert not torch.equal(before, pose.detach())
    c2, r2, v2 = _render_indexed(hip, rs, ix, pose)
    cf, rf, vf = _render_indexed(hip, rs, ix, pose.detach().clone())
    assert not torch.equal(c1, c2) and torch.equal(c2, cf) and torch.equal(r2, rf) and torch.equal(v2, vf)
    # 3. an in-place torch op under no_grad
    with torch.no_grad():
        pose[6] += 0.7
    c3, _, _ = _render_indexed(hip, rs, ix, pose)
    assert not torch.equal(c2, c3) and torch.equal(c3, _render_indexed(hip, rs, ix, pose.detach().clone())[0])


def test_intrinsic_written_through_data_is_noticed(hip):
    """The host scalars taken from `intrinsic` (image size, tan FoV) are cached per tensor object; a `.data` write does not
    bump the version counter, so the forward verifies the cache by value at its own synchronisation point and renders again."""
    W, H, focal = 320, 200, 200.0
    intr, ev = synth.camera(W, H, focal)
    sc = synth.scene(3000, W, H, focal, seed=6, scale_median=0.03)
    ix = {k: v.cuda() for k, v in synth.index_scene(sc, shs_extra=64, gs_extra=64).items()}
    intr_d = intr.cuda()
    rs = _settings(hip, intr_d, ev)
    evd = ev.cuda()
    c0, _, _ = _render_indexed(hip, rs, ix, evd)
    c0b, _, _ = _render_indexed(hip, rs, ix, evd)                  # second call: served from the cache
    assert c0.shape == (3, H, W) and torch.equal(c0, c0b)
    intr2, _ = synth.camera(256, 160, 140.0)
    intr_d.data.copy_(intr2.cuda())                                # same object, same version, new values
    c1, _, _ = _render_indexed(hip, rs, ix, evd)
    assert c1.shape == (3, 160, 256)
    fresh = _render_indexed(hip, _settings(hip, intr2.cuda(), ev), ix, evd)[0]
    assert torch.equal(c1, fresh)


def test_abs_accumulate_entry_point(hip):
    """c3dgs_abs_accumulate (the sensitivity pass's acc += |g|): sizes that are not multiples of 4, unaligned views, n = 0."""
    import torch
    from c3dgs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    for n, off in [(1, 0), (7, 0), (1000003, 0), (4096, 1), (4099, 3), (0, 0)]:
        a = torch.randn(n + off, generator=g).cuda()
        acc = torch.rand(n + off, generator=g).cuda()
        want = acc.clone()
        want[off:] += a[off:].abs()
        _lib.check(L.c3dgs_abs_accumulate(n, a[off:].data_ptr() if n else None, acc[off:].data_ptr() if n else None, None))
        torch.cuda.synchronize()
        assert torch.equal(acc, want), (n, off)
    assert L.c3dgs_abs_accumulate(5, None, None, None) == 1


def test_matrix_extrinsic_api_of_the_sibling_packages(hip, orc):
    """diff_gaussian_rasterization / diff_gaussian_rasterization_camera (reference __init__.py:41-96, 503-659): settings
    without a pose, `extrinsic=` 4x4 on forward / markVisible. Same kernels: the render and every gradient must equal the
    quaternion API's for the same pose (the camera set-up differs only by torch's matmul / inverse round-off), and
    install_as_reference_modules() must hand out THIS module under both sibling names."""
    import sys
    import c3dgs_amd
    from c3dgs_amd import rasterizer_matrix as rm
    W, H, focal = 200, 136, 125.0
    intr, ev = synth.camera(W, H, focal, extrinsic_vector=(0.05, -0.03, 0.02, 0.99, 0.1, -0.05, 0.2))
    extrinsic = hip.quat_to_mat(ev).cuda()
    rs = rm.GaussianRasterizationSettings(intrinsic=intr.cuda(), bg=torch.tensor((0.2, 0.4, 0.1), device="cuda"), scale_modifier=1.0,
                                          sh_degree=3, prefiltered=False, debug=False, clamp_color=True)
    assert rs._fields == ("intrinsic", "bg", "scale_modifier", "sh_degree", "prefiltered", "debug", "clamp_color")
    dL = synth.grad_image(W, H)
    for name, indexed in (("base", False), ("indexed", True)):
        inp, cam, _ = cases.make_case(name)
        st = cases.oracle_forward(inp, cam)
        ref = orc.rasterize_backward(st, dL.numpy())
        keys = ("means3D", "opacities", "shs", "scales", "rotations") + (("scale_factors",) if indexed else ())
        leaves = {k: inp[k].cuda().requires_grad_() for k in keys}
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
        if indexed:
            rast = rm.GaussianRasterizerIndexed(rs)
            color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                sh_indices=inp["sh_indices"].cuda(), g_indices=inp["g_indices"].cuda(), shs=leaves["shs"],
                                scales=leaves["scales"], scale_factors=leaves["scale_factors"], rotations=leaves["rotations"],
                                extrinsic=extrinsic)
        else:
            rast = rm.GaussianRasterizer(rs)
            color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                                scales=leaves["scales"], rotations=leaves["rotations"], extrinsic=extrinsic)
        assert rast.markVisible(leaves["means3D"], extrinsic).all()
        (color * dL.cuda()).sum().backward()
        np.testing.assert_array_equal(radii.cpu().numpy(), st.radii)
        assert gpu_util.psnr(color.detach().cpu().numpy(), st.out_color) >= 80.0
        pairs = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", shs="dL_dsh", scales="dL_dscales", rotations="dL_drotations")
        if indexed:
            pairs["scale_factors"] = "dL_dscale_factors"
        for k, r in pairs.items():
            assert gpu_util.rel_inf(leaves[k].grad.cpu().numpy(), ref[r].reshape(leaves[k].shape)) <= 1e-4, (name, k)
    with pytest.raises(RuntimeError, match="4x4"):
        rm.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=inp["shs"].cuda(),
                                  scales=inp["scales"].cuda()[:1], rotations=inp["rotations"].cuda()[:1], extrinsic=ev.cuda())
    saved = {k: sys.modules.get(k) for k in ("diff_gaussian_rasterization", "diff_gaussian_rasterization_camera",
                                             "diff_gaussian_rasterization_no_camera", "weighted_distance", "weighted_distance._C")}
    try:
        c3dgs_amd.install_as_reference_modules()
        import diff_gaussian_rasterization as a
        import diff_gaussian_rasterization_camera as b
        import diff_gaussian_rasterization_no_camera as c
        assert a is rm and b is rm and c is c3dgs_amd.rasterizer
        assert "extrinsic_vector" in c.GaussianRasterizationSettings._fields and "extrinsic_vector" not in a.GaussianRasterizationSettings._fields
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
