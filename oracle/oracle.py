"""ctypes/numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (c3dgs_amd/) never does; it fails loudly when its HIP library is missing.

Everything numeric lives in oracle/c3dgs_oracle.c (see its header for the pinning statement);
this file only marshals numpy arrays and restates the reference's host-side camera set-up
(submodules/diff-gaussian-rasterization-no-camera/diff_gaussian_rasterization_no_camera/__init__.py:19-40,152-176).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libc3dgs_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class _Params(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("bg", _f32p), ("means3D", _f32p), ("shs", _f32p), ("colors_precomp", _f32p),
        ("opacities", _f32p), ("scales", _f32p), ("scale_factors", _f32p), ("rotations", _f32p),
        ("cov3D_precomp", _f32p), ("sh_indices", _i64p), ("g_indices", _i64p),
        ("viewmatrix", _f32p), ("projmatrix", _f32p), ("campos", _f32p),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int), ("clamp_color", C.c_int), ("SHS", C.c_int), ("GS", C.c_int),
    ]


class _Geom(C.Structure):
    _fields_ = [
        ("depths", _f32p), ("clamped", _u8p), ("radii", _i32p), ("means2D", _f32p), ("cov3D", _f32p),
        ("conic_opacity", _f32p), ("rgb", _f32p), ("tiles_touched", _u32p), ("point_offsets", _u32p),
    ]


def build(force=False):
    """Compile oracle/libc3dgs_oracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "c3dgs_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_forward_stage1.restype = C.c_int
        _lib.orc_forward_stage1.argtypes = [C.POINTER(_Params), C.POINTER(_Geom)]
        _lib.orc_forward_stage2.restype = None
        _lib.orc_forward_stage2.argtypes = [C.POINTER(_Params), C.POINTER(_Geom), C.c_int, _u64p, _u32p, _u64p,
                                            _u32p, _u32p, _f32p, _f32p, _u32p]
        _lib.orc_backward.restype = None
        _lib.orc_backward.argtypes = [C.POINTER(_Params), C.POINTER(_Geom), C.c_int, _u32p, _u32p, _f32p, _u32p,
                                      _f32p] + [_f32p] * 10
        _lib.orc_mark_visible.restype = None
        _lib.orc_mark_visible.argtypes = [C.c_int, _f32p, _f32p, _f32p, _u8p]
        _lib.orc_weighted_distance.restype = None
        _lib.orc_weighted_distance.argtypes = [C.c_int64, C.c_int, C.c_int, _f32p, _f32p, _f32p, _i64p]
        _lib.orc_vq_update.restype = C.c_double
        _lib.orc_vq_update.argtypes = [C.c_int64, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_double,
                                       C.c_double, _f32p, _i64p]
        _lib.orc_vq_trace_normalize.restype = None
        _lib.orc_vq_trace_normalize.argtypes = [C.c_int, C.c_int, _f32p]
        _lib.orc_get_higher_msb.restype = C.c_uint32
        _lib.orc_get_higher_msb.argtypes = [C.c_uint32]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_test_color_from_sh.restype = None
        _lib.orc_test_color_from_sh.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, _u8p, _f32p]
        _lib.orc_l1_ssim.restype = None
        _lib.orc_l1_ssim.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_double, C.POINTER(C.c_double), _f32p]
        _lib.orc_morton_codes.restype = None
        _lib.orc_morton_codes.argtypes = [C.c_int, _f32p, _i64p, _i32p]
        _lib.orc_test_cov3d.restype = None
        _lib.orc_test_cov3d.argtypes = [C.c_int, _f32p, C.c_float, _f32p, _f32p]
        _lib.orc_test_sh_backward.restype = None
        _lib.orc_test_sh_backward.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u8p, _f32p, _f32p, _f32p]
        _lib.orc_test_cov3d_backward.restype = None
        _lib.orc_test_cov3d_backward.argtypes = [C.c_int, _f32p, C.c_float, _f32p, _f32p, _f32p, _f32p]
    return _lib


def num_threads():
    return int(lib().orc_num_threads())


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


# ----------------------------------------------------------------------------- camera set-up
def projection_matrix(intrinsic):
    """getProjectionMatrix, DGR-NC __init__.py:19-30 (returned TRANSPOSED, as the reference does)."""
    znear, zfar, z_sign = 0.01, 100.0, 1.0
    ty = math.tan(float(intrinsic[1][1]) / 2)
    tx = math.tan(float(intrinsic[0][0]) / 2)
    Pm = np.array([[1.0 / tx, 0, 0, 0], [0, 1.0 / ty, 0, 0],
                   [0, 0, z_sign * zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
                   [0, 0, z_sign, 0]], dtype=np.float32)
    return np.ascontiguousarray(Pm.T)


def quat_to_mat(ev):
    """quat_to_mat, DGR-NC __init__.py:32-40 (returned TRANSPOSED). The reference evaluates the entries with fp32 tensor
    arithmetic on the elements of the pose, left to right; same fp32 operations here (pinned by tests/golden/camera.npz)."""
    f = np.float32
    x, y, z, w, tx, ty, tz = [f(v) for v in np.asarray(ev, dtype=np.float32)]
    one, two = f(1.0), f(2.0)
    d2 = y * y + z * z + x * x
    m = np.array([[one + two * (x * x - d2), two * (x * y - w * z), two * (x * z + w * y), tx],
                  [two * (x * y + w * z), one + two * (y * y - d2), two * (y * z - w * x), ty],
                  [two * (x * z - w * y), two * (y * z + w * x), one + two * (z * z - d2), tz],
                  [0, 0, 0, 1.0]], dtype=np.float32)
    return np.ascontiguousarray(m.T)


def camera(intrinsic, extrinsic_vector):
    """-> dict(viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, W, H); DGR-NC __init__.py:152-172."""
    view = quat_to_mat(extrinsic_vector)
    proj = (view @ projection_matrix(intrinsic)).astype(np.float32)
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=view, projmatrix=np.ascontiguousarray(proj), campos=np.ascontiguousarray(campos),
                tan_fovx=float(math.tan(float(intrinsic[0][0]) * 0.5)),
                tan_fovy=float(math.tan(float(intrinsic[1][1]) * 0.5)),
                W=int(intrinsic[0][2]), H=int(intrinsic[1][2]))


# ----------------------------------------------------------------------------- raster
class RasterState:
    """Everything the oracle computed for one forward (inputs kept alive for the backward)."""


def _make_params(st):
    i = st.inputs
    prm = _Params()
    prm.P, prm.D, prm.M, prm.W, prm.H = st.P, int(i["degree"]), st.M, st.W, st.H
    prm.bg = _p(i["bg"], _f32p)
    prm.means3D = _p(i["means3D"], _f32p)
    prm.shs = _p(i["shs"], _f32p)
    prm.colors_precomp = _p(i["colors_precomp"], _f32p)
    prm.opacities = _p(i["opacities"], _f32p)
    prm.scales = _p(i["scales"], _f32p)
    prm.scale_factors = _p(i["scale_factors"], _f32p)
    prm.rotations = _p(i["rotations"], _f32p)
    prm.cov3D_precomp = _p(i["cov3D_precomp"], _f32p)
    prm.sh_indices = _p(i["sh_indices"], _i64p)
    prm.g_indices = _p(i["g_indices"], _i64p)
    prm.viewmatrix = _p(i["viewmatrix"], _f32p)
    prm.projmatrix = _p(i["projmatrix"], _f32p)
    prm.campos = _p(i["campos"], _f32p)
    prm.tan_fovx, prm.tan_fovy = float(i["tan_fovx"]), float(i["tan_fovy"])
    prm.scale_modifier = float(i["scale_modifier"])
    prm.prefiltered, prm.clamp_color = int(i["prefiltered"]), int(i["clamp_color"])
    prm.SHS = 0 if i["shs"] is None else int(i["shs"].shape[0])
    prm.GS = 0 if i["scales"] is None else int(i["scales"].shape[0])
    return prm


def _make_geom(st):
    g = _Geom()
    g.depths = _p(st.depths, _f32p)
    g.clamped = _p(st.clamped, _u8p)
    g.radii = _p(st.radii, _i32p)
    g.means2D = _p(st.means2D, _f32p)
    g.cov3D = _p(st.cov3D, _f32p)
    g.conic_opacity = _p(st.conic_opacity, _f32p)
    g.rgb = _p(st.rgb, _f32p)
    g.tiles_touched = _p(st.tiles_touched, _u32p)
    g.point_offsets = _p(st.point_offsets, _u32p)
    return g


def rasterize_forward(*, bg, means3D, opacities, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, W, H,
                      shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                      scale_factors=None, sh_indices=None, g_indices=None, degree=0, scale_modifier=1.0,
                      prefiltered=False, clamp_color=True):
    """Reference Rasterizer::forward / forward_indexed (rasterizer_impl.cu:194-334, 440-586)."""
    L = lib()
    st = RasterState()
    means3D = _f32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    shs = None if shs is None else _f32(shs)
    if shs is not None and shs.ndim == 2:
        shs = shs.reshape(shs.shape[0], -1, 3)
    st.inputs = dict(
        bg=_f32(bg), means3D=means3D, shs=shs, colors_precomp=_f32(colors_precomp),
        opacities=_f32(opacities).reshape(-1), scales=_f32(scales), scale_factors=None if scale_factors is None else _f32(scale_factors).reshape(-1),
        rotations=_f32(rotations), cov3D_precomp=_f32(cov3D_precomp),
        sh_indices=None if sh_indices is None else np.ascontiguousarray(sh_indices, dtype=np.int64),
        g_indices=None if g_indices is None else np.ascontiguousarray(g_indices, dtype=np.int64),
        viewmatrix=_f32(viewmatrix).reshape(16), projmatrix=_f32(projmatrix).reshape(16), campos=_f32(campos).reshape(3),
        tan_fovx=tan_fovx, tan_fovy=tan_fovy, degree=degree, scale_modifier=scale_modifier,
        prefiltered=prefiltered, clamp_color=clamp_color)
    st.P, st.W, st.H = P, int(W), int(H)
    st.M = 0 if shs is None else int(shs.shape[1])
    gx, gy = (st.W + 15) // 16, (st.H + 15) // 16
    st.T, st.N = gx * gy, st.W * st.H
    st.depths = np.zeros(P, np.float32)
    st.clamped = np.zeros((P, 3), np.uint8)
    st.radii = np.zeros(P, np.int32)
    st.means2D = np.zeros((P, 2), np.float32)
    st.cov3D = np.zeros((P, 6), np.float32)
    st.conic_opacity = np.zeros((P, 4), np.float32)
    st.rgb = np.zeros((P, 3), np.float32)
    st.tiles_touched = np.zeros(P, np.uint32)
    st.point_offsets = np.zeros(P, np.uint32)
    st.out_color = np.zeros((3, st.H, st.W), np.float32)
    st.final_T = np.zeros(st.N, np.float32)
    st.n_contrib = np.zeros(st.N, np.uint32)
    st.ranges = np.zeros((st.T, 2), np.uint32)
    prm, geom = _make_params(st), _make_geom(st)
    R = L.orc_forward_stage1(C.byref(prm), C.byref(geom)) if P > 0 else 0
    st.num_rendered = int(R)
    st.keys_unsorted = np.zeros(R, np.uint64)
    st.values_unsorted = np.zeros(R, np.uint32)
    st.keys_sorted = np.zeros(R, np.uint64)
    st.point_list = np.zeros(R, np.uint32)
    if P > 0:
        L.orc_forward_stage2(C.byref(prm), C.byref(geom), R, _p(st.keys_unsorted, _u64p), _p(st.values_unsorted, _u32p),
                             _p(st.keys_sorted, _u64p), _p(st.point_list, _u32p), _p(st.ranges, _u32p),
                             _p(st.out_color, _f32p), _p(st.final_T, _f32p), _p(st.n_contrib, _u32p))
    return st


def rasterize_backward(st, dL_dout_color):
    """Reference Rasterizer::backward / backward_indexed (rasterizer_impl.cu:338-435, 590-697)."""
    L = lib()
    i = st.inputs
    P, M = st.P, st.M
    indexed = i["sh_indices"] is not None or i["g_indices"] is not None
    SHS = 0 if i["shs"] is None else i["shs"].shape[0]
    GS = 0 if i["scales"] is None else i["scales"].shape[0]
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 4), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((SHS if indexed else (P if i["shs"] is not None else 0), M, 3), np.float32),
        dL_dscales=np.zeros((GS if indexed else (P if i["scales"] is not None else 0), 3), np.float32),
        dL_dscale_factors=np.zeros((P, 1), np.float32),
        dL_drotations=np.zeros((GS if indexed else (P if i["scales"] is not None else 0), 4), np.float32))
    dpix = _f32(dL_dout_color).reshape(3, st.H, st.W)
    if P > 0:
        prm, geom = _make_params(st), _make_geom(st)
        L.orc_backward(C.byref(prm), C.byref(geom), st.num_rendered, _p(st.point_list, _u32p), _p(st.ranges, _u32p),
                       _p(st.final_T, _f32p), _p(st.n_contrib, _u32p), _p(dpix, _f32p),
                       _p(g["dL_dmeans2D"], _f32p), _p(g["dL_dconic"], _f32p), _p(g["dL_dopacity"], _f32p),
                       _p(g["dL_dcolors"], _f32p), _p(g["dL_dmeans3D"], _f32p), _p(g["dL_dcov3D"], _f32p),
                       _p(g["dL_dsh"], _f32p), _p(g["dL_dscales"], _f32p), _p(g["dL_dscale_factors"], _f32p),
                       _p(g["dL_drotations"], _f32p))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D).reshape(-1, 3)
    out = np.zeros(means3D.shape[0], np.uint8)
    v, pr = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16)
    lib().orc_mark_visible(means3D.shape[0], _p(means3D, _f32p), _p(v, _f32p), _p(pr, _f32p), _p(out, _u8p))
    return out.astype(bool)


def get_higher_msb(n):
    return int(lib().orc_get_higher_msb(int(n)))


# ----------------------------------------------------------------------------- VQ
def weighted_distance(coefs, codebook):
    """WD/weighted_distance.cu:20-58 -> (min_sq_dist f32[N], argmin i64[N])."""
    x, cb = _f32(coefs), _f32(codebook)
    if x.ndim != 2 or cb.ndim != 2:
        raise RuntimeError("ceofs and codebook must have dimension 2")
    if x.shape[1] != cb.shape[1]:
        raise RuntimeError("coefs and codebook must have same number of channels")
    N, K = x.shape
    d = np.zeros(N, np.float32)
    ix = np.zeros(N, np.int64)
    lib().orc_weighted_distance(N, cb.shape[0], K, _p(x, _f32p), _p(cb, _f32p), _p(d, _f32p), _p(ix, _i64p))
    return d, ix


def vq_update(x, w, codebook, entry_importance, decay=0.8, eps=1e-5):
    """VectorQuantize.update (compression/vq.py:28-35), in place on codebook/entry_importance."""
    x, w = _f32(x), _f32(w)
    assert codebook.dtype == np.float32 and codebook.flags.c_contiguous
    assert entry_importance.dtype == np.float32 and entry_importance.flags.c_contiguous
    B, D = x.shape
    md = np.zeros(B, np.float32)
    ix = np.zeros(B, np.int64)
    mean = lib().orc_vq_update(B, codebook.shape[0], D, _p(x, _f32p), _p(w, _f32p), _p(codebook, _f32p),
                               _p(entry_importance, _f32p), float(decay), float(eps), _p(md, _f32p), _p(ix, _i64p))
    return md, ix, float(mean)


def vq_features(features, importance, codebook_size, init_rand, batches, decay=0.8, scale_normalize=False):
    """vq_features (compression/vq.py:49-87) with the two RNG draws passed in as DATA:
    init_rand = the rand_like draw of uniform_init (vq.py:26), batches = the randint draws (vq.py:69)."""
    f = _f32(features)
    imp = _f32(importance)
    imp_n = (imp / imp.max()).astype(np.float32)
    amin, amax = f.min(), f.max()
    cb = np.ascontiguousarray((_f32(init_rand) * (amax - amin) + amin).astype(np.float32))
    assert cb.shape == (codebook_size, f.shape[1])
    ent = np.zeros(codebook_size, np.float32)
    errors = []
    for b in batches:
        b = np.asarray(b, dtype=np.int64)
        _, _, mean = vq_update(f[b], imp_n[b], cb, ent, decay=decay)
        errors.append(mean)
        if scale_normalize:
            lib().orc_vq_trace_normalize(cb.shape[0], cb.shape[1], _p(cb, _f32p))
    _, idx = weighted_distance(f, cb)
    return cb, idx, np.asarray(errors), ent


# ----------------------------------------------------------------------------- test hooks
def color_from_sh(deg, pos, campos, sh, clamp_color=False):
    """forward.cu:20-79 for n Gaussians: sh [n,M,3], pos [n,3] -> (rgb [n,3], clamped [n,3])."""
    pos, sh, campos = _f32(pos), _f32(sh), _f32(campos)
    n, M = sh.shape[0], sh.shape[1]
    rgb = np.zeros((n, 3), np.float32)
    cl = np.zeros((n, 3), np.uint8)
    lib().orc_test_color_from_sh(n, int(deg), M, _p(pos, _f32p), _p(campos, _f32p), _p(sh, _f32p), int(clamp_color),
                                 _p(cl, _u8p), _p(rgb, _f32p))
    return rgb, cl


def cov3d(scales, mod, rotations):
    """forward.cu:126-160 -> [n,6]."""
    s, q = _f32(scales), _f32(rotations)
    out = np.zeros((s.shape[0], 6), np.float32)
    lib().orc_test_cov3d(s.shape[0], _p(s, _f32p), float(mod), _p(q, _f32p), _p(out, _f32p))
    return out


def sh_backward(deg, pos, campos, sh, dL_dcolor, clamped=None):
    """backward.cu:20-139 -> (dL_dsh [n,M,3], SH part of dL_dmean [n,3])."""
    pos, campos, sh, g = _f32(pos), _f32(campos), _f32(sh), _f32(dL_dcolor)
    n, M = sh.shape[0], sh.shape[1]
    cl = np.zeros((n, 3), np.uint8) if clamped is None else np.ascontiguousarray(clamped, dtype=np.uint8)
    dsh, dmean = np.zeros((n, M, 3), np.float32), np.zeros((n, 3), np.float32)
    lib().orc_test_sh_backward(n, int(deg), M, _p(pos, _f32p), _p(campos, _f32p), _p(sh, _f32p), _p(cl, _u8p), _p(g, _f32p),
                               _p(dsh, _f32p), _p(dmean, _f32p))
    return dsh, dmean


def cov3d_backward(scales, mod, rotations, dL_dcov3D):
    """backward.cu:278-341 -> (dL_dscale [n,3] (w.r.t. mod * scale, as the reference returns it), dL_drot [n,4] w.r.t. the
    UN-normalised quaternion)."""
    s, q, g = _f32(scales), _f32(rotations), _f32(dL_dcov3D)
    ds, dq = np.zeros((s.shape[0], 3), np.float32), np.zeros((s.shape[0], 4), np.float32)
    lib().orc_test_cov3d_backward(s.shape[0], _p(s, _f32p), float(mod), _p(q, _f32p), _p(g, _f32p), _p(ds, _f32p), _p(dq, _f32p))
    return ds, dq


# ----------------------------------------------------------------------------- N3: loss
def l1_ssim(img, gt, lambda_dssim=0.2, want_grad=True):
    """(loss, l1, ssim, dL_dimg) of loss = (1-l)*L1 + l*(1-SSIM) (finetune.py:48, utils/loss_utils.py:17-63)."""
    img, gt = _f32(img), _f32(gt)
    Cc, H, W = img.shape
    out = (C.c_double * 3)()
    g = np.zeros_like(img) if want_grad else None
    lib().orc_l1_ssim(Cc, H, W, _p(img, _f32p), _p(gt, _f32p), float(lambda_dssim), out, _p(g, _f32p))
    return float(out[0]), float(out[1]), float(out[2]), g


# ----------------------------------------------------------------------------- N4: Morton order
def morton_codes(xyz):
    """(codes int64[P], axis order int32[3]) as GaussianModel._sort_morton computes them before sorting."""
    xyz = _f32(xyz).reshape(-1, 3)
    codes = np.zeros(xyz.shape[0], np.int64)
    order = np.zeros(3, np.int32)
    lib().orc_morton_codes(xyz.shape[0], _p(xyz, _f32p), _p(codes, _i64p), _p(order, _i32p))
    return codes, order


def morton_order(xyz):
    codes, _ = morton_codes(xyz)
    return np.argsort(codes, kind="stable")
