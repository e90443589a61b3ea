"""-m gpu parity tests proper: the HIP rasterizer (through the C-ABI front-end) against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star / BASELINE.md section 2):
  * radii, tiles_touched, scan, unsorted keys+values, sorted keys, sorted point_list, ranges: BIT-EXACT
  * image: PSNR(hip, oracle) >= 80 dB and |PSNR(hip,target) - PSNR(oracle,target)| <= 0.05 dB vs a fixed random target
  * gradients: ||g - g_ref||_inf / ||g_ref||_inf <= 1e-4 per tensor (oracle accumulates in float64)
"""
import numpy as np
import pytest
import torch

from tests import cases, fullsize, gpu_util, synth

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4


def _check_forward(hipo, st, borderline_ok=False):
    assert hipo["num_rendered"] == st.num_rendered
    np.testing.assert_array_equal(hipo["radii"], st.radii)
    if st.P > 0:
        np.testing.assert_array_equal(hipo["tiles_touched"], st.tiles_touched)
        # binning stage 1: Gaussians in (depth bits, id) order, culled ones last
        vis_ids = np.nonzero(st.radii > 0)[0]
        want = vis_ids[np.lexsort((vis_ids, st.depths[vis_ids].view(np.uint32)))]
        np.testing.assert_array_equal(hipo["depth_order"][:len(want)], want.astype(np.uint32))
        vis = st.radii > 0
        # per-Gaussian floats feeding the keys: bit-exact (same fp32 op order, no contraction)
        np.testing.assert_array_equal(hipo["depths"][vis].view(np.uint32), st.depths[vis].view(np.uint32))
        np.testing.assert_array_equal(hipo["means2D"][vis].view(np.uint32), st.means2D[vis].view(np.uint32))
        np.testing.assert_array_equal(hipo["conic_opacity"][vis].view(np.uint32), st.conic_opacity[vis].view(np.uint32))
        np.testing.assert_array_equal(hipo["clamped"][vis], st.clamped[vis])
        if st.inputs["colors_precomp"] is None:
            np.testing.assert_array_equal(hipo["rgb"][vis].view(np.uint32), st.rgb[vis].view(np.uint32))
    if st.num_rendered > 0:
        # same (key, value) pairs as the reference emits (emission ORDER is depth-major here: binning.hip) ...
        a = np.lexsort((hipo["values_unsorted"], hipo["keys_unsorted"]))
        b = np.lexsort((st.values_unsorted, st.keys_unsorted))
        np.testing.assert_array_equal(hipo["keys_unsorted"][a], st.keys_unsorted[b])
        np.testing.assert_array_equal(hipo["values_unsorted"][a], st.values_unsorted[b])
        # ... and bit-identical sorted keys and sorted point list
        np.testing.assert_array_equal(hipo["keys_sorted"], st.keys_sorted)
        np.testing.assert_array_equal(hipo["point_list"], st.point_list)
    np.testing.assert_array_equal(hipo["ranges"], st.ranges)
    # image
    a, b = hipo["out_color"], st.out_color
    assert np.isfinite(a).all()
    assert gpu_util.psnr(a, b) >= 80.0, f"PSNR(hip, oracle) = {gpu_util.psnr(a, b):.2f} dB"
    rng = np.random.default_rng(5)
    target = rng.random(a.shape, dtype=np.float32)
    assert abs(gpu_util.psnr(a, target) - gpu_util.psnr(b, target)) <= 0.05
    if borderline_ok:
        # deep blend lists (thousands of splats per pixel): an exp() ulp may flip one alpha >= 1/255 / T < 1e-4 decision at a
        # pixel, which moves it by at most one borderline contribution (alpha T c <= 1/255); everything else stays tight
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(b)
        assert bad.mean() <= 1e-4 and np.abs(a - b).max() <= 4e-3, (bad.sum(), np.abs(a - b).max())
    else:
        # every pixel tight -- except that any two exp() implementations disagree on a blend decision (alpha >= 1/255,
        # T < 1e-4) now and then (tests/fullsize.py: 32 such pixels over 315 random scenes, the same count for two
        # different roundings of the exponent): at most two pixels, each off by at most one borderline contribution
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(b)
        assert bad.any(0).sum() <= max(2, int(2e-5 * a.shape[1] * a.shape[2])) and np.abs(a - b).max() <= 4e-3, \
            (bad.sum(), np.abs(a - b).max())
    # per-pixel bookkeeping: identical except where an exp() ulp flips a threshold (expected: almost never)
    same = (hipo["n_contrib"] == st.n_contrib).mean()
    assert same >= 0.999, f"n_contrib agreement {same}"
    if borderline_ok:
        dT = np.abs(hipo["final_T"] - st.final_T)
        assert (dT > 1e-5 + 1e-4 * np.abs(st.final_T)).mean() <= 1e-4 and dT.max() <= 4e-3
    else:
        dT = np.abs(hipo["final_T"] - st.final_T)
        assert (dT > 1e-5 + 1e-4 * np.abs(st.final_T)).sum() <= max(2, int(2e-5 * dT.size)) and dT.max() <= 4e-3


@pytest.mark.parametrize("name", cases.FORWARD_CASES)
def test_forward_parity(hip, orc, name):
    inp, cam, indexed = cases.make_case(name)
    st = cases.oracle_forward(inp, cam)
    fw = gpu_util.hip_forward(inp, cam, indexed)
    _check_forward(gpu_util.unpack(fw), st)


@pytest.mark.parametrize("name", [c for c in cases.FORWARD_CASES])
def test_backward_parity(hip, orc, name):
    inp, cam, indexed = cases.make_case(name)
    st = cases.oracle_forward(inp, cam)
    dL = synth.grad_image(cam["W"], cam["H"]).numpy()
    ref = orc.rasterize_backward(st, dL)
    fw = gpu_util.hip_forward(inp, cam, indexed)
    got = gpu_util.hip_backward(fw, dL)
    fullsize.check_grads(st, gpu_util.unpack(fw), got, ref, GRAD_TOL, name)


def test_backward_is_deterministic(hip, orc):
    """No global float atomics on the non-indexed path: two runs are bitwise identical."""
    inp, cam, indexed = cases.make_case("base")
    dL = synth.grad_image(cam["W"], cam["H"]).numpy()
    fw = gpu_util.hip_forward(inp, cam, indexed)
    g1 = gpu_util.hip_backward(fw, dL)
    g2 = gpu_util.hip_backward(fw, dL)
    for k in g1:
        np.testing.assert_array_equal(g1[k].view(np.uint32), g2[k].view(np.uint32))


def test_mark_visible(hip, orc):
    inp, cam, _ = cases.make_case("behind")
    from c3dgs_amd import rasterizer
    got = rasterizer._C.mark_visible(inp["means3D"].cuda(), torch.from_numpy(cam["viewmatrix"]).cuda(),
                                     torch.from_numpy(cam["projmatrix"]).cuda()).cpu().numpy()
    ref = orc.mark_visible(inp["means3D"].numpy(), cam["viewmatrix"], cam["projmatrix"])
    np.testing.assert_array_equal(got, ref)
    assert 0 < ref.sum() < ref.size


def test_mark_visible_from_the_pose_equals_the_matrix_path(hip, orc):
    """GaussianRasterizer*.markVisible with a 7-element pose takes c3dgs_mark_visible_pose (one launch: the matrix entries the
    test reads are formed per thread). Its flags must equal camera_from_pose + mark_visible bit for bit -- checked on points
    spread THROUGH the z = 0.01 plane of a camera with an un-normalised quaternion, where one ulp of a matrix entry flips flags."""
    import c3dgs_amd
    from c3dgs_amd import rasterizer as rz
    g = torch.Generator().manual_seed(5)
    intr, _ = synth.camera(320, 200, 200.0)
    for trial in range(4):
        q = torch.randn(4, generator=g) * (1.0 if trial else 0.3)            # not normalised (the reference does not either)
        t = torch.randn(3, generator=g)
        ev = torch.cat([q, t]).float()
        cam = orc.camera(intr.numpy(), ev.numpy())
        W2C = torch.from_numpy(cam["viewmatrix"]).T.double()                   # viewmatrix is stored transposed
        n = 200_003
        pts_cam = torch.randn(n, 3, generator=g).double()
        pts_cam[:, 2] = 0.01 + (torch.rand(n, generator=g).double() - 0.5) * torch.logspace(-7, 0, n).double()
        Rm, tv = W2C[:3, :3], W2C[:3, 3]
        pts = (torch.linalg.solve(Rm, (pts_cam - tv).T).T).float().cuda()
        rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=intr, extrinsic_vector=ev.cuda(), bg=torch.zeros(3).cuda(), scale_modifier=1.0,
                                                     sh_degree=0, prefiltered=False, debug=False, clamp_color=True)
        got = c3dgs_amd.GaussianRasterizerIndexed(rs).markVisible(pts, extrinsic_vector=ev.cuda())
        assert rz._mark_visible_from_pose(pts, ev.cuda()) is not None
        view, proj = rz.camera_matrices(intr, ev.cuda(), pts.device)[:2]
        ref = rz._C.mark_visible(pts, view, proj)
        assert torch.equal(got, ref)
        assert 0.2 < float(ref.float().mean()) < 0.8
        np.testing.assert_array_equal(got.cpu().numpy(), orc.mark_visible(pts.cpu().numpy(), view.cpu().numpy(), proj.cpu().numpy()))


def test_forward_with_the_copy_based_host_read(hip):
    """The forward's one device->host read normally arrives through MAPPED host memory the scan kernel stores into;
    C3DGS_HOST_READ_COPY=1 forces the hipMemcpyAsync + event path (what a host that cannot map the pad falls back to). The
    parity cases must pass unchanged with it. Child process: the switch is read once per thread."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, C3DGS_HOST_READ_COPY="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_raster_gpu.py", "-q", "-m", "gpu", "-x",
                        "-k", "forward_parity and (base or p8193 or empty or all_behind or indexed)"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_full_hd_properties(hip):
    """BASELINE.json full size (1920x1080, 1M Gaussians) through size-independent properties:
    sorted keys are non-decreasing, the sorted list is a permutation of the unsorted one, ranges partition
    the list by tile, sum(tiles_touched) == R, transmittance in [0,1], and backward is finite."""
    W, H, focal, P = 1920, 1080, 1200.0, 1_000_000
    intr, ev = synth.camera(W, H, focal)
    from oracle import oracle as o
    cam = o.camera(intr.numpy(), ev.numpy())
    sc = synth.scene(P, W, H, focal)
    inp = dict(bg=torch.zeros(3), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
               rotations=sc["rotations"], degree=3, clamp_color=True)
    fw = gpu_util.hip_forward(inp, cam, False)
    u = gpu_util.unpack(fw)
    R = u["num_rendered"]
    assert R == int(u["tiles_touched"].astype(np.int64).sum()) and R > P
    ks = u["keys_sorted"]
    assert np.all(ks[1:] >= ks[:-1])
    np.testing.assert_array_equal(np.sort(u["keys_unsorted"]), ks)
    order = np.lexsort((u["values_unsorted"], u["keys_unsorted"]))     # (tile, depth, id): ties in id order
    np.testing.assert_array_equal(u["values_unsorted"][order], u["point_list"])
    tiles = (ks >> np.uint64(32)).astype(np.int64)
    rg = u["ranges"].astype(np.int64)
    cnt = np.bincount(tiles, minlength=rg.shape[0])
    np.testing.assert_array_equal(rg[:, 1] - rg[:, 0], cnt)
    assert ((u["final_T"] >= 0) & (u["final_T"] <= 1)).all()
    assert np.isfinite(u["out_color"]).all()
    g = gpu_util.hip_backward(fw, synth.grad_image(W, H).numpy())
    for k, v in g.items():
        assert np.isfinite(v).all(), k
    assert np.abs(g["dL_dmeans3D"]).max() > 0


def test_4k_image_properties(hip):
    """3840x2160 (32,400 tiles: 15 tile-id bits, two uneven digit passes of the tile sort) with the indexed path: the same
    size-independent properties, and the per-Gaussian gradients that involve no float atomics are bitwise reproducible."""
    W, H, focal, P = 3840, 2160, 2400.0, 600_000
    intr, ev = synth.camera(W, H, focal)
    from oracle import oracle as o
    cam = o.camera(intr.numpy(), ev.numpy())
    sc = synth.scene(P, W, H, focal, seed=77, scale_median=0.012)
    ix = synth.index_scene(sc, seed=78)
    inp = dict(bg=torch.zeros(3), means3D=ix["means3D"], opacities=ix["opacities"], shs=ix["shs"], scales=ix["scales"],
               rotations=ix["rotations"], scale_factors=ix["scale_factors"], sh_indices=ix["sh_indices"],
               g_indices=ix["g_indices"], degree=3, clamp_color=True)
    fw = gpu_util.hip_forward(inp, cam, True)
    u = gpu_util.unpack(fw)
    R = u["num_rendered"]
    assert R == int(u["tiles_touched"].astype(np.int64).sum()) and R > P
    ks = u["keys_sorted"]
    assert np.all(ks[1:] >= ks[:-1]) and int(ks[-1] >> np.uint64(32)) > 2 ** 14          # the 15th tile bit is in use
    order = np.lexsort((u["values_unsorted"], u["keys_unsorted"]))
    np.testing.assert_array_equal(u["values_unsorted"][order], u["point_list"])
    rg = u["ranges"].astype(np.int64)
    cnt = np.bincount((ks >> np.uint64(32)).astype(np.int64), minlength=rg.shape[0])
    np.testing.assert_array_equal(rg[:, 1] - rg[:, 0], cnt)
    assert ((u["final_T"] >= 0) & (u["final_T"] <= 1)).all() and np.isfinite(u["out_color"]).all()
    g = gpu_util.hip_backward(fw, synth.grad_image(W, H).numpy())
    for k, v in g.items():
        assert np.isfinite(v).all(), k
    assert np.abs(g["dL_dmeans3D"]).max() > 0
    g2 = gpu_util.hip_backward(fw, synth.grad_image(W, H).numpy())
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscale_factors"):                         # no float atomics on these
        np.testing.assert_array_equal(g[k].view(np.uint32), g2[k].view(np.uint32))


def test_debug_and_prefiltered_flags(hip, orc):
    """debug=True synchronises after every stage (reference CHECK_CUDA) and changes no result; prefiltered=True skips
    the near-plane test (auxiliary.h:146-147), which is a no-op on a scene that is entirely in front of the camera."""
    from c3dgs_amd import rasterizer as rz
    inp, cam, _ = cases.make_case("base")
    fw0 = gpu_util.hip_forward(inp, cam, False)
    a = list(fw0["args"])
    a[-2] = True                                    # debug
    out_dbg = rz._C.rasterize_gaussians(*a)
    assert out_dbg[0] == fw0["num_rendered"] and torch.equal(out_dbg[1], fw0["color"]) and torch.equal(out_dbg[2], fw0["radii"])
    a[-2], a[-3] = False, True                      # prefiltered
    out_pre = rz._C.rasterize_gaussians(*a)
    assert torch.equal(out_pre[1], fw0["color"]) and torch.equal(out_pre[2], fw0["radii"])
    g0 = gpu_util.hip_backward(fw0, synth.grad_image(cam["W"], cam["H"]).numpy())
    assert all(np.isfinite(v).all() for v in g0.values())


@pytest.mark.parametrize("mod,deg", [(1.0, 3), (1.7, 2), (1.0, 1)])
def test_hip_backward_vs_reference_autograd_fixtures(hip, orc, mod, deg):
    """The HIP backward DIRECTLY against what torch.autograd gives through the reference's own importable Python
    (tests/golden/sh_bwd.npz, cov3d_bwd.npz; generated by make_golden.py), no oracle in between:
      * SH backward (backward.cu:20-139):   dL_dsh[p, k, c] = basis_k(dir_p) * dL_dcolors[p, c] * !clamped[p, c], with basis taken
        from autograd through utils/sh_utils.py:eval_sh on normalize(pos - campos);
      * cov3D backward (backward.cu:278-341): dL_dscales * mod and the tangential part of dL_drotations are the reference's
        Jacobians of build_covariance_from_scaling_rotation applied to the kernel's own dL_dcov3D (the kernel returns
        dL/d(mod * scale) and differentiates w.r.t. the un-normalised quaternion: SURVEY App. A.2 / A.8)."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sh, cv = np.load(os.path.join(G, "sh_bwd.npz")), np.load(os.path.join(G, "cov3d_bwd.npz"))
    cam = orc.camera(sh["intrinsic"], sh["extrinsic_vector"])
    np.testing.assert_allclose(cam["campos"], sh["campos"], rtol=1e-5, atol=2e-6)         # the reference's own camera centre
    P = sh["pos"].shape[0]
    g = torch.Generator().manual_seed(33)
    inp = dict(bg=torch.tensor([0.1, 0.2, 0.3]), means3D=torch.from_numpy(sh["pos"]), opacities=torch.rand(P, 1, generator=g) * 0.7 + 0.2,
               shs=torch.from_numpy(sh["sh"]) * 2.0, colors_precomp=None, scales=torch.from_numpy(cv["scales"]),
               rotations=torch.from_numpy(cv["rotations"]), cov3D_precomp=None, scale_factors=None, sh_indices=None, g_indices=None,
               degree=deg, scale_modifier=mod, prefiltered=False, clamp_color=True)
    fw = gpu_util.hip_forward(inp, cam, False)
    u = gpu_util.unpack(fw)
    vis = u["radii"] > 0
    assert vis.sum() >= 0.9 * P                                                            # the fixture's points face the camera
    got = gpu_util.hip_backward(fw, synth.grad_image(cam["W"], cam["H"], seed=5).numpy())
    keep = (1 - u["clamped"].astype(np.float64)) * vis[:, None]
    assert 0 < (u["clamped"][vis] != 0).mean() < 0.5                                       # both clamped and unclamped channels occur
    # basis of the degree under test: autograd of eval_sh(deg, ...) is zero above (deg + 1)^2; sh was scaled by 2 (colour only)
    want_sh = sh[f"basis_deg{deg}"].astype(np.float64)[:, :, None] * (got["dL_dcolors"].astype(np.float64) * keep)[:, None, :]
    assert gpu_util.rel_inf(got["dL_dsh"], want_sh) <= 1e-4
    assert np.abs(got["dL_dsh"]).max() > 0
    dcov = got["dL_dcov3D"].astype(np.float64)
    want_s = np.einsum("pk,pkc->pc", dcov, cv[f"jac_scale_mod{mod}"].astype(np.float64))
    assert gpu_util.rel_inf(got["dL_dscales"].astype(np.float64) * mod, want_s) <= 1e-4
    q = cv["rotations"].astype(np.float64)
    dq = got["dL_drotations"].astype(np.float64)
    tang = dq - (dq * q).sum(1, keepdims=True) * q
    want_q = np.einsum("pk,pkc->pc", dcov, cv[f"jac_rot_mod{mod}"].astype(np.float64))
    assert gpu_util.rel_inf(tang, want_q) <= 1e-4
    assert np.abs(want_q).max() > 0 and np.abs(want_s).max() > 0


def test_8k_image_with_32_bit_tile_keys_matches_the_oracle(hip, orc):
    """7680x4320 = 480 x 270 = 129,600 tiles: beyond the 65,536 that 16-bit tile keys can name, the library switches to 32-bit
    keys (three digit passes of the tile sort), as the reference's 64-bit keys allow (rasterizer_impl.cu:98-108). Against the
    oracle on the same inputs: radii, the (tile << 32 | depth) keys, the sorted point list and the ranges equal, image PSNR >= 80 dB,
    gradients within the flip-aware bars. A few thousand Gaussians keep the oracle's 33-Mpixel view at a few seconds; some of them
    are made screen-filling so that rectangles of tens of thousands of tiles (the emission's division) are exercised."""
    W, H, focal, P = 7680, 4320, 4800.0, 6000
    intr, ev = synth.camera(W, H, focal)
    cam = orc.camera(intr.numpy(), ev.numpy())
    sc = synth.scene(P, W, H, focal, seed=81, scale_median=0.02)
    sc["scales"][:12] *= 60.0                                    # a dozen splats thousands of pixels wide
    sc["opacities"][:12] = 0.05
    inp = dict(bg=torch.tensor([0.1, 0.0, 0.2]), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], colors_precomp=None,
               scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=None, scale_factors=None, sh_indices=None, g_indices=None,
               degree=3, scale_modifier=1.0, prefiltered=False, clamp_color=True)
    st = cases.oracle_forward(inp, cam)
    assert st.T == 129_600 and st.num_rendered > 100_000
    fw = gpu_util.hip_forward(inp, cam, False)
    u = gpu_util.unpack(fw)
    assert u["num_rendered"] == st.num_rendered
    np.testing.assert_array_equal(u["radii"], st.radii)
    np.testing.assert_array_equal(u["keys_sorted"], st.keys_sorted)
    assert int(st.keys_sorted[-1] >> np.uint64(32)) > 65535      # tile ids beyond 16 bits are in use
    np.testing.assert_array_equal(u["point_list"], st.point_list)
    np.testing.assert_array_equal(u["ranges"], st.ranges)
    assert gpu_util.psnr(u["out_color"], st.out_color) >= 80.0
    dL = synth.grad_image(W, H).numpy()
    ref = orc.rasterize_backward(st, dL)
    got = gpu_util.hip_backward(fw, dL)
    fullsize.check_grads(st, u, got, ref, GRAD_TOL, "8k")
