"""Host-side mirror of the reference's rasterizer package
(submodules/diff-gaussian-rasterization-no-camera/diff_gaussian_rasterization_no_camera/__init__.py)
on top of the MI355X C-ABI library.  Same names, argument order and error behaviour:

    GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerIndexed,
    rasterize_gaussians, rasterize_gaussians_indexed, rasterize_gaussians_indexed_camera,
    getProjectionMatrix, quat_to_mat, mat_to_quat, and `_C` with the five pybind entry points
    (submodules/diff-gaussian-rasterization/ext.cpp:15-21).

PyTorch is plumbing only here (device memory, streams, autograd bookkeeping); every numeric stage
runs in libc3dgs_hip.so.  There is no CPU path: tensors must live on the GPU.
"""
import ctypes as C
import threading
import math
import weakref
from types import SimpleNamespace
from typing import NamedTuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import RESIZE_FN, RasterGrads, RasterParams


# ----------------------------------------------------------------------------- camera helpers
def getProjectionMatrix(intrinsic, device=None):
    """reference __init__.py:19-30 (znear 0.01, zfar 100; returned transposed)."""
    znear, zfar, z_sign = 0.01, 100.0, 1.0
    tanHalfFovY = math.tan((float(intrinsic[1, 1]) / 2))
    tanHalfFovX = math.tan((float(intrinsic[0, 0]) / 2))
    m = torch.tensor([
        [1.0 / tanHalfFovX, 0.0, 0.0, 0.0],
        [0.0, 1.0 / tanHalfFovY, 0.0, 0.0],
        [0.0, 0.0, z_sign * zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
        [0.0, 0.0, z_sign, 0.0]], dtype=torch.float32).transpose(0, 1).contiguous()
    return m if device is None else m.to(device)


def quat_to_mat(extrinsic_vector, device=None):
    """reference __init__.py:32-40: (qx,qy,qz,qw,tx,ty,tz) -> 4x4 world->camera, returned transposed.
    The reference evaluates the entries with fp32 tensor arithmetic on the pose's elements; the same fp32 operations in
    the same order here (numpy scalars; one device->host transfer instead of the reference's seven scalar reads).
    The rendering path does not call this: it builds the matrices on the device (camera_matrices)."""
    f = np.float32
    x, y, z, w, tx, ty, tz = [f(v) for v in extrinsic_vector.detach().cpu().float().tolist()]
    one, two = f(1.0), f(2.0)
    d2 = y * y + z * z + x * x
    m = torch.from_numpy(np.array([
        [one + two * (x * x - d2), two * (x * y - w * z), two * (x * z + w * y), tx],
        [two * (x * y + w * z), one + two * (y * y - d2), two * (y * z - w * x), ty],
        [two * (x * z - w * y), two * (y * z + w * x), one + two * (z * z - d2), tz],
        [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)).transpose(0, 1).contiguous()
    return m if device is None else m.to(device)


def mat_to_quat(m, normed=True):
    """reference __init__.py:42-52."""
    w = torch.sqrt(1.0 + m[0, 0] + m[1, 1] + m[2, 2]) / 2.0
    w4 = 4.0 * w
    x = (m[2, 1] - m[1, 2]) / w4
    y = (m[0, 2] - m[2, 0]) / w4
    z = (m[1, 0] - m[0, 1]) / w4
    if normed:
        norm2 = (x * x + y * y + z * z + w * w) ** 0.5
        x, y, z, w = x / norm2, y / norm2, z / norm2, w / norm2
    return x, y, z, w, m[0, 3], m[1, 3], m[2, 3]


# ---- intrinsic -> host scalars. W, H size the outputs and tan(FoV/2) are kernel arguments, so they must be known on the
# host. For a CUDA intrinsic the values are cached per tensor object + in-place version (no device->host read on a hit)
# and VERIFIED by value at the forward's own synchronisation point (see _IntrinsicGuard), so a write the version counter
# does not see (`.data`, raw pointers) cannot go unnoticed either.
_INTRINSIC_CACHE = {}
_INTRINSIC_CACHE_MAX = 512


def _intrinsic_scalars_from_values(vals):
    """vals: the 9 floats of the reference's intrinsic ([0,0]=FoVx, [1,1]=FoVy, [0,2]=W, [1,2]=H; scene/cameras.py:39-41)
    -> (tanfovx, tanfovy, H, W, 1/tan(FoVx/2) as fp32, 1/tan(FoVy/2) as fp32); __init__.py:19-30, 152-157."""
    fovx, fovy = float(vals[0]), float(vals[4])
    tanfovx, tanfovy = float(math.tan(fovx * 0.5)), float(math.tan(fovy * 0.5))
    # getProjectionMatrix divides the fp32 FoV by 2 (exact) and takes tan / reciprocal in double; torch.Tensor rounds to fp32
    inv_x = float(np.float32(1.0 / math.tan(fovx / 2))) if tanfovx != 0 else float("inf")
    inv_y = float(np.float32(1.0 / math.tan(fovy / 2))) if tanfovy != 0 else float("inf")
    return tanfovx, tanfovy, int(vals[5]), int(vals[2]), inv_x, inv_y


def _intrinsic_scalars(intrinsic):
    """-> (scalars, values tuple, cached: bool)."""
    if not intrinsic.is_cuda:
        vals = tuple(intrinsic.detach().reshape(-1).float().tolist())
        return _intrinsic_scalars_from_values(vals), vals, False
    key = id(intrinsic)
    ent = _INTRINSIC_CACHE.get(key)
    if ent is not None and ent[0]() is intrinsic and ent[1] == intrinsic._version and not intrinsic.requires_grad:
        return ent[2], ent[3], True
    vals = tuple(intrinsic.detach().reshape(-1).float().cpu().tolist())            # one device->host read per new tensor
    sc = _intrinsic_scalars_from_values(vals)
    if len(_INTRINSIC_CACHE) >= _INTRINSIC_CACHE_MAX:
        _INTRINSIC_CACHE.clear()
    _INTRINSIC_CACHE[key] = (weakref.ref(intrinsic), intrinsic._version, sc, vals)
    return sc, vals, False


class _IntrinsicGuard:
    """Checks a cache hit of _intrinsic_scalars by VALUE without an extra synchronisation: the 9 floats are copied to pinned
    memory on the caller's stream BEFORE the forward is queued; the forward itself waits for a later event on that stream
    (its read of num_rendered), so afterwards the copy is complete and comparing costs nothing. `ok()` False -> the cache
    entry was stale: it has been dropped and the caller renders again with fresh values."""
    _pinned = threading.local()

    def __init__(self, intrinsic, vals, cached):
        self.active = bool(cached)
        if not self.active:
            return
        ring = getattr(self._pinned, "ring", None)
        if ring is None:
            ring = self._pinned.ring = [[torch.empty(9, dtype=torch.float32).pin_memory(), torch.cuda.Event()] for _ in range(4)]
            self._pinned.k = 0
        self.host, self.ev = ring[self._pinned.k % 4]
        self._pinned.k += 1
        self.ev.synchronize()                                   # the slot's previous copy (four guards ago): long done
        self.intrinsic, self.vals = intrinsic, vals
        with torch.cuda.device(intrinsic.device):
            self.host.copy_(intrinsic.detach().reshape(-1).float(), non_blocking=True)
            self.ev.record(torch.cuda.current_stream(intrinsic.device))

    def ok(self):
        if not self.active:
            return True
        self.ev.synchronize()
        if tuple(self.host.tolist()) == self.vals:
            return True
        _INTRINSIC_CACHE.pop(id(self.intrinsic), None)
        return False


def _camera_on_host(intrinsic_scalars, extrinsic_vector):
    """CPU restatement of csrc/preprocess.hip:camera_from_pose_kernel (same fp32 operations; used for CPU tensors, i.e. by
    the host-logic tests, and as what the GPU tests compare the kernel with)."""
    _, _, _, _, inv_x, inv_y = intrinsic_scalars
    f = np.float32
    view = quat_to_mat(extrinsic_vector).numpy()                 # transposed: view[i][j] = M[j][i]
    pa, pb = f(1.0 * 100.0 / (100.0 - 0.01)), f(-(100.0 * 0.01) / (100.0 - 0.01))
    proj = np.empty((4, 4), np.float32)
    proj[:, 0] = view[:, 0] * f(inv_x)
    proj[:, 1] = view[:, 1] * f(inv_y)
    proj[:, 2] = view[:, 2] * pa + view[:, 3] * pb
    proj[:, 3] = view[:, 2]
    R, t = view[:3, :3].T.astype(np.float64), view[3, :3].astype(np.float64)
    a, b, c, d, e, ff, g, h, k = R.reshape(-1)
    A, B, Cc = e * k - ff * h, -(d * k - ff * g), d * h - e * g
    det = a * A + b * B + c * Cc
    inv = np.array([[A / det, -(b * k - c * h) / det, (b * ff - c * e) / det],
                    [B / det, (a * k - c * g) / det, -(a * ff - c * d) / det],
                    [Cc / det, -(a * h - b * g) / det, (a * e - b * d) / det]])
    campos = (-(inv @ t)).astype(np.float32)
    return torch.from_numpy(view.copy()), torch.from_numpy(proj), torch.from_numpy(campos)


def camera_matrices(intrinsic, extrinsic_vector, device, _guard=None):
    """(viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W) as the reference's autograd wrapper assembles them
    (__init__.py:152-172). On a GPU the three tensors are computed BY THE DEVICE from the pose's live values
    (c3dgs_camera_from_pose, stream-ordered): nothing about the pose is cached on the host, so in-place optimiser updates of
    the pose -- torch ops, `.data` writes or the package's fused Adam writing through raw pointers -- are always seen, and the
    call never synchronises. The host scalars come from `intrinsic` (cached per tensor, verified by value at the forward's
    synchronisation point when `_guard` is given a list to receive the _IntrinsicGuard)."""
    dev = torch.device(device)
    sc, vals, cached = _intrinsic_scalars(intrinsic)
    tanfovx, tanfovy, image_height, image_width, inv_x, inv_y = sc
    if extrinsic_vector.dim() == 2:
        # the sibling packages' pose: the 4x4 world->camera matrix itself (transposed, as quat_to_mat returns it), set up with
        # the reference's own three tensor operations (diff_gaussian_rasterization/__init__.py:129-135): stream-ordered torch
        # ops on the pose's device, nothing cached on the host
        if tuple(extrinsic_vector.shape) != (4, 4):
            raise RuntimeError("extrinsic must be a 4x4 matrix")
        if _guard is not None and dev.type == "cuda":
            _guard.append(_IntrinsicGuard(intrinsic, vals, cached))
        view = extrinsic_vector.detach().to(device=dev, dtype=torch.float32).contiguous()
        pa, pb = np.float32(1.0 * 100.0 / (100.0 - 0.01)), np.float32(-(100.0 * 0.01) / (100.0 - 0.01))
        P = torch.tensor([[inv_x, 0.0, 0.0, 0.0], [0.0, inv_y, 0.0, 0.0], [0.0, 0.0, float(pa), float(pb)], [0.0, 0.0, 1.0, 0.0]],
                         dtype=torch.float32).transpose(0, 1).contiguous().to(dev)      # == getProjectionMatrix(intrinsic)
        return view, (view @ P).contiguous(), view.inverse()[3, :3].contiguous(), tanfovx, tanfovy, image_height, image_width
    if dev.type != "cuda":
        view, proj, campos = _camera_on_host(sc, extrinsic_vector)
        return view, proj, campos, tanfovx, tanfovy, image_height, image_width
    if _guard is not None:
        _guard.append(_IntrinsicGuard(intrinsic, vals, cached))
    pose = extrinsic_vector.detach()
    if pose.device != dev or pose.dtype != torch.float32 or not pose.is_contiguous():
        pose = pose.to(device=dev, dtype=torch.float32).contiguous()
    if pose.numel() != 7:
        raise RuntimeError("extrinsic_vector must have 7 elements (qx, qy, qz, qw, tx, ty, tz)")
    with torch.cuda.device(dev):
        out = torch.empty(36, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().c3dgs_camera_from_pose(pose.data_ptr(), inv_x, inv_y, out.data_ptr(), out.data_ptr() + 64,
                                                     out.data_ptr() + 128, _stream(dev)))
    return out[:16].view(4, 4), out[16:32].view(4, 4), out[32:35], tanfovx, tanfovy, image_height, image_width


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


# ----------------------------------------------------------------------------- C-ABI plumbing
class _Scratch:
    """Owns torch uint8 buffers grown by the library through the resize callbacks
    (the reference's resizeFunctional lambdas, rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self._cbs = {}

    def callback(self, name):
        def _resize(_user, nbytes):
            t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
            self.bufs[name] = t
            return t.data_ptr()
        cb = RESIZE_FN(_resize)
        self._cbs[name] = cb
        return cb

    def release(self):
        """-> the buffers; drops the callbacks. Must be called once the library call has returned: each callback
        closes over `self`, so the object sits in a reference cycle and would otherwise keep its buffers (hundreds of
        MB per call) alive until the cyclic garbage collector runs -- the caching allocator then has to hipMalloc new
        blocks every few calls."""
        bufs, self.bufs, self._cbs = self.bufs, {}, {}
        return bufs

    def _or_empty(self, bufs, name):
        t = bufs.get(name)
        return t if t is not None else torch.empty(0, dtype=torch.uint8, device=self.device)


def _f32c(t, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"c3dgs_amd: {name} must be a GPU tensor (there is no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


def _i64c(t, name):
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"c3dgs_amd: {name} must be a GPU tensor (there is no CPU path)")
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _params(keep, *, background, means3D, colors, opacity, scales, scale_factors, rotations, scale_modifier, cov3D_precomp,
            viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, sh_indices, g_indices, prefiltered,
            debug, clamp_color):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")      # rasterize_points.cu:58-60
    if not means3D.is_cuda:
        raise RuntimeError("c3dgs_amd: means3D must be a GPU tensor (there is no CPU path)")
    t = dict(background=_f32c(background, "background"), means3D=_f32c(means3D, "means3D"), sh=_f32c(sh, "sh"),
             colors_precomp=_f32c(colors, "colors_precomp"), opacities=_f32c(opacity, "opacities"),
             scales=_f32c(scales, "scales"), scale_factors=_f32c(scale_factors, "scale_factors"),
             rotations=_f32c(rotations, "rotations"), cov3D_precomp=_f32c(cov3D_precomp, "cov3D_precomp"),
             sh_indices=_i64c(sh_indices, "sh_indices"), g_indices=_i64c(g_indices, "g_indices"),
             viewmatrix=_f32c(viewmatrix, "viewmatrix"), projmatrix=_f32c(projmatrix, "projmatrix"),
             campos=_f32c(campos, "campos"))
    keep.append(t)
    p = RasterParams()
    p.P = int(means3D.size(0))
    p.D = int(degree)
    shp = t["sh"]
    p.M = int(shp.size(1)) if shp is not None else 0                           # rasterize_points.cu:84-88
    p.W, p.H = int(W), int(H)
    p.SHS = int(shp.size(0)) if shp is not None else 0
    p.GS = int(t["scales"].size(0)) if t["scales"] is not None else 0
    for k in ("background", "means3D", "sh", "colors_precomp", "opacities", "scales", "scale_factors", "rotations",
              "cov3D_precomp", "sh_indices", "g_indices", "viewmatrix", "projmatrix", "campos"):
        setattr(p, k, _ptr(t[k]))
    p.tan_fovx, p.tan_fovy, p.scale_modifier = float(tan_fovx), float(tan_fovy), float(scale_modifier)
    p.prefiltered, p.clamp_color, p.debug = int(bool(prefiltered)), int(bool(clamp_color)), int(bool(debug))
    return p, t


def _forward(indexed, background, means3D, colors, opacity, scales, scale_factors, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, sh_indices,
             g_indices, prefiltered, debug, clamp_color):
    L = _lib.lib()
    keep = []
    p, _ = _params(keep, background=background, means3D=means3D, colors=colors, opacity=opacity, scales=scales,
                   scale_factors=scale_factors, rotations=rotations, scale_modifier=scale_modifier,
                   cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, tan_fovx=tan_fovx,
                   tan_fovy=tan_fovy, H=image_height, W=image_width, sh=sh, degree=degree, campos=campos,
                   sh_indices=sh_indices, g_indices=g_indices, prefiltered=prefiltered, debug=debug, clamp_color=clamp_color)
    dev = means3D.device
    P, H, W = p.P, p.H, p.W
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        scratch = _Scratch(dev)
        num_rendered = C.c_int32(0)
        fn = L.c3dgs_rasterize_gaussians_indexed if indexed else L.c3dgs_rasterize_gaussians
        rc = fn(C.byref(p), scratch.callback("geom"), None, scratch.callback("binning"), None, scratch.callback("img"), None,
                out_color.data_ptr(), radii.data_ptr() if P > 0 else None, C.byref(num_rendered), _stream(dev))
    bufs = scratch.release()
    _lib.check(rc)
    return (int(num_rendered.value), out_color, radii, scratch._or_empty(bufs, "geom"), scratch._or_empty(bufs, "binning"),
            scratch._or_empty(bufs, "img"))


# Outputs the autograd wrappers do not need (gradients of ABSENT optional inputs) are neither allocated nor written:
# dL_dcolors (12 B x P) and dL_dcov3D (24 B x P) are 108 MB of stores per 3M-Gaussian backward. The pybind-order
# entry points (_C.*) keep returning every tensor like the reference.
_SKIP = threading.local()


def _backward(indexed, background, means3D, radii, colors, scales, scale_factors, rotations, scale_modifier, cov3D_precomp,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
              imageBuffer, debug, sh_indices, g_indices):
    L = _lib.lib()
    keep = []
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))               # rasterize_points.cu:143-145
    # opacities are not an input of the reference's backward either (they live in the geometry buffer)
    p, t = _params(keep, background=background, means3D=means3D, colors=colors, opacity=None, scales=scales,
                   scale_factors=scale_factors, rotations=rotations, scale_modifier=scale_modifier,
                   cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, tan_fovx=tan_fovx,
                   tan_fovy=tan_fovy, H=H, W=W, sh=sh, degree=degree, campos=campos, sh_indices=sh_indices,
                   g_indices=g_indices, prefiltered=False, debug=debug, clamp_color=False)
    dev = means3D.device
    P, M = p.P, p.M
    SHS, GS = p.SHS, p.GS
    opt = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        dL_dmeans3D = torch.empty((P, 3), **opt)
        dL_dmeans2D = torch.empty((P, 3), **opt)
        skip = getattr(_SKIP, "names", ())
        dL_dcolors = None if "dL_dcolors" in skip else torch.empty((P, 3), **opt)
        dL_dopacity = torch.empty((P, 1), **opt)
        dL_dcov3D = None if "dL_dcov3D" in skip else torch.empty((P, 6), **opt)
        if indexed:
            # zeroed + scatter-added inside the library; carved from ONE allocation (sh | rotations | scales, every start
            # 16-byte aligned) so that the library clears them with a single fill
            n_sh, n_rot, n_sc = SHS * M * 3, GS * 4, GS * 3
            flat = torch.empty(n_sh + n_rot + n_sc, **opt)
            dL_dsh = flat[:n_sh].view(SHS, M, 3)
            dL_drotations = flat[n_sh:n_sh + n_rot].view(GS, 4)
            dL_dscales = flat[n_sh + n_rot:].view(GS, 3)
            dL_dscale_factors = torch.empty((P, 1), **opt)
        else:
            # reference shapes (rasterize_points.cu:153-162): [P,M,3], [P,3], [P,4]; rows stay zero when the
            # corresponding input is absent
            dL_dsh = torch.empty((P, M, 3), **opt) if t["sh"] is not None else torch.zeros((P, M, 3), **opt)
            dL_dscales = torch.empty((P, 3), **opt) if t["scales"] is not None else torch.zeros((P, 3), **opt)
            dL_drotations = torch.empty((P, 4), **opt) if t["scales"] is not None else torch.zeros((P, 4), **opt)
            dL_dscale_factors = None
        g = RasterGrads()
        g.dL_dmeans2D, g.dL_dcolors, g.dL_dopacity = dL_dmeans2D.data_ptr(), _ptr(dL_dcolors), dL_dopacity.data_ptr()
        g.dL_dmeans3D, g.dL_dcov3D = dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D)
        g.dL_dsh = dL_dsh.data_ptr() if dL_dsh.numel() else None
        g.dL_dscales = dL_dscales.data_ptr() if dL_dscales.numel() else None
        g.dL_drotations = dL_drotations.data_ptr() if dL_drotations.numel() else None
        g.dL_dscale_factors = dL_dscale_factors.data_ptr() if (dL_dscale_factors is not None and P > 0) else None
        radii_c = radii.contiguous()
        dpix = _f32c(dL_dout_color, "dL_dout_color")
        scratch = _Scratch(dev)
        fn = L.c3dgs_rasterize_gaussians_backward_indexed if indexed else L.c3dgs_rasterize_gaussians_backward
        rc = fn(C.byref(p), radii_c.data_ptr() if P > 0 else None,
                geomBuffer.data_ptr() if geomBuffer.numel() else None,
                binningBuffer.data_ptr() if binningBuffer.numel() else None,
                imageBuffer.data_ptr() if imageBuffer.numel() else None, int(R), _ptr(dpix),
                scratch.callback("ws"), None, C.byref(g), _stream(dev))
    scratch.release()       # the workspace goes back to the (stream-ordered) caching allocator right away
    _lib.check(rc)
    if indexed:
        if t["scales"] is None:
            dL_dscale_factors.zero_()
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_dscale_factors,
                dL_drotations)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


# ---- the five entry points with the pybind argument order (ext.cpp:15-21, rasterize_points.h:18-122)
def _c_rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                           projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                           clamp_color):
    return _forward(False, background, means3D, colors, opacity, scales, None, rotations, scale_modifier, cov3D_precomp,
                    viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, None, None,
                    prefiltered, debug, clamp_color)


def _c_rasterize_gaussians_indexed(background, means3D, colors, opacity, scales, scale_factors, rotations, scale_modifier,
                                   cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                                   degree, campos, sh_indices, g_indices, prefiltered, debug, clamp_color):
    return _forward(True, background, means3D, colors, opacity, scales, scale_factors, rotations, scale_modifier,
                    cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                    sh_indices, g_indices, prefiltered, debug, clamp_color)


def _c_rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                    viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                    geomBuffer, R, binningBuffer, imageBuffer, debug):
    return _backward(False, background, means3D, radii, colors, scales, None, rotations, scale_modifier, cov3D_precomp,
                     viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                     binningBuffer, imageBuffer, debug, None, None)


def _c_rasterize_gaussians_backward_indexed(background, means3D, radii, colors, scales, scale_factors, rotations,
                                            scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                            dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                            debug, sh_indices, g_indices):
    return _backward(True, background, means3D, radii, colors, scales, scale_factors, rotations, scale_modifier,
                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer,
                     R, binningBuffer, imageBuffer, debug, sh_indices, g_indices)


def _c_mark_visible(means3D, viewmatrix, projmatrix):
    """rasterize_points.cu:202-221 -> bool[P]."""
    L = _lib.lib()
    if not means3D.is_cuda:
        raise RuntimeError("c3dgs_amd: means3D must be a GPU tensor (there is no CPU path)")
    m = _f32c(means3D, "means3D")
    P = int(means3D.size(0))
    present = torch.empty((P,), dtype=torch.bool, device=means3D.device)     # the kernel writes every element
    if P != 0:
        v, pr = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
        with torch.cuda.device(means3D.device):
            rc = L.c3dgs_mark_visible(P, m.data_ptr(), v.data_ptr(), pr.data_ptr(), present.data_ptr(), _stream(means3D.device))
        _lib.check(rc)
    return present


def _mark_visible_from_pose(positions, extrinsic_vector):
    """markVisible for a 7-element pose on the GPU in ONE launch (c3dgs_mark_visible_pose: same flags, bit for bit, as
    camera_matrices + _C.mark_visible, without the single-thread matrix kernel in front). None: not that case."""
    if not positions.is_cuda or extrinsic_vector.dim() != 1 or extrinsic_vector.numel() != 7:
        return None
    dev = positions.device
    pose = extrinsic_vector.detach()
    if pose.device != dev or pose.dtype != torch.float32 or not pose.is_contiguous():
        pose = pose.to(device=dev, dtype=torch.float32).contiguous()
    m = _f32c(positions, "means3D")
    P = int(positions.size(0))
    present = torch.empty((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            rc = _lib.lib().c3dgs_mark_visible_pose(P, m.data_ptr(), pose.data_ptr(), present.data_ptr(), _stream(dev))
        _lib.check(rc)
    return present


_C = SimpleNamespace(
    rasterize_gaussians=_c_rasterize_gaussians,
    rasterize_gaussians_backward=_c_rasterize_gaussians_backward,
    rasterize_gaussians_indexed=_c_rasterize_gaussians_indexed,
    rasterize_gaussians_backward_indexed=_c_rasterize_gaussians_backward_indexed,
    mark_visible=_c_mark_visible,
)


# ----------------------------------------------------------------------------- autograd functions
def _call_debug(fn, args, debug, dump):
    """reference __init__.py:179-206: on failure under debug, dump a CPU copy of the arguments and re-raise."""
    if debug:
        cpu_args = cpu_deep_copy_tuple(args)
        try:
            return fn(*args)
        except Exception as ex:
            torch.save(cpu_args, dump)
            print(f"\nAn error occured. Writing {dump} for debugging.")
            raise ex
    return fn(*args)


class _RasterizeGaussians(torch.autograd.Function):
    """reference __init__.py:136-323."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                extrinsic_vector):
        dev = means3D.device
        for _attempt in range(2):      # a second pass only if the cached intrinsic scalars turn out stale (_IntrinsicGuard)
            guard = []
            view, proj, campos, tanfovx, tanfovy, H, W = camera_matrices(raster_settings.intrinsic, extrinsic_vector, dev, guard)
            args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations, raster_settings.scale_modifier,
                    cov3Ds_precomp, view, proj, tanfovx, tanfovy, H, W, sh, raster_settings.sh_degree, campos,
                    raster_settings.prefiltered, raster_settings.debug, raster_settings.clamp_color)
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _call_debug(
                _C.rasterize_gaussians, args, raster_settings.debug, "snapshot_fw.dump")
            if all(g.ok() for g in guard):
                break
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.camera = (view, proj, campos, tanfovx, tanfovy)
        ctx.save_for_backward(extrinsic_vector, colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.image_shape = tuple(color.shape)
        ctx.mark_non_differentiable(radii)
        # without this, autograd hands backward a zero-filled int32[P] "gradient" for radii on every call (a 12 MB fill at P = 3M)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, *params):
        rs = ctx.raster_settings
        (extrinsic_vector, colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        grad_out_color = _dense_grad(grad_out_color, ctx)
        view, proj, campos, tanfovx, tanfovy = ctx.camera
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, view, proj,
                tanfovx, tanfovy, grad_out_color, sh, rs.sh_degree, campos, geomBuffer, ctx.num_rendered, binningBuffer,
                imgBuffer, rs.debug)
        _SKIP.names = tuple(n for n, t in (("dL_dcolors", colors_precomp), ("dL_dcov3D", cov3Ds_precomp)) if t is None or t.numel() == 0)
        try:
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
             grad_rotations) = _call_debug(_C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump")
        finally:
            _SKIP.names = ()
        return (grad_means3D, grad_means2D, _fit(grad_sh, sh), _fit(grad_colors_precomp, colors_precomp), grad_opacities,
                _fit(grad_scales, scales), _fit(grad_rotations, rotations), _fit(grad_cov3Ds_precomp, cov3Ds_precomp),
                None, None)


def _fit(grad, inp):
    """The reference returns full-size gradients even for absent ("empty") inputs; autograd ignores them because
    those inputs never require grad. Returning None for them is equivalent and skips the shape check."""
    if inp is None or inp.numel() == 0:
        return None
    return grad


def _indexed_forward(ctx, means3D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors, rotations,
                     cov3Ds_precomp, raster_settings, extrinsic_vector):
    dev = means3D.device
    for _attempt in range(2):          # a second pass only if the cached intrinsic scalars turn out stale (_IntrinsicGuard)
        guard = []
        view, proj, campos, tanfovx, tanfovy, H, W = camera_matrices(raster_settings.intrinsic, extrinsic_vector, dev, guard)
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, scale_factors, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, view, proj, tanfovx, tanfovy, H, W, sh,
                raster_settings.sh_degree, campos, sh_indices, g_indices, raster_settings.prefiltered, raster_settings.debug,
                raster_settings.clamp_color)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _call_debug(
            _C.rasterize_gaussians_indexed, args, raster_settings.debug, "snapshot_fw.dump")
        if all(g.ok() for g in guard):
            break
    ctx.raster_settings = raster_settings
    ctx.num_rendered = num_rendered
    ctx.camera = (view, proj, campos, tanfovx, tanfovy)
    ctx.save_for_backward(extrinsic_vector, colors_precomp, means3D, scales, scale_factors, rotations, cov3Ds_precomp, radii,
                          sh, geomBuffer, binningBuffer, imgBuffer, sh_indices, g_indices)
    ctx.mark_non_differentiable(radii)
    ctx.set_materialize_grads(False)       # no zero-filled int32[P] "gradient" for radii (see _RasterizeGaussians.forward)
    ctx.image_shape = tuple(color.shape)
    return color, radii


def _dense_grad(grad_out_color, ctx):
    """set_materialize_grads(False): an image nobody differentiated through arrives as None (backward still runs when only
    `radii` was used downstream of a graph that needs grad); the library wants a dense dL/dC."""
    if grad_out_color is not None:
        return grad_out_color
    return torch.zeros(ctx.image_shape, dtype=torch.float32, device=ctx.saved_tensors[2].device)


def _indexed_backward(ctx, grad_out_color):
    rs = ctx.raster_settings
    grad_out_color = _dense_grad(grad_out_color, ctx)
    (extrinsic_vector, colors_precomp, means3D, scales, scale_factors, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
     binningBuffer, imgBuffer, sh_indices, g_indices) = ctx.saved_tensors
    view, proj, campos, tanfovx, tanfovy = ctx.camera
    args = (rs.bg, means3D, radii, colors_precomp, scales, scale_factors, rotations, rs.scale_modifier, cov3Ds_precomp, view,
            proj, tanfovx, tanfovy, grad_out_color, sh, rs.sh_degree, campos, geomBuffer, ctx.num_rendered, binningBuffer,
            imgBuffer, rs.debug, sh_indices, g_indices)
    _SKIP.names = tuple(n for n, t in (("dL_dcolors", colors_precomp), ("dL_dcov3D", cov3Ds_precomp)) if t is None or t.numel() == 0)
    try:
        out = _call_debug(_C.rasterize_gaussians_backward_indexed, args, rs.debug, "snapshot_bw.dump")
    finally:
        _SKIP.names = ()
    return out, (extrinsic_vector, colors_precomp, means3D, scales, scale_factors, rotations, cov3Ds_precomp, sh)


class _RasterizeGaussiansIndexed(torch.autograd.Function):
    """reference __init__.py:326-530."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors, rotations,
                cov3Ds_precomp, raster_settings, extrinsic_vector):
        return _indexed_forward(ctx, means3D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors,
                                rotations, cov3Ds_precomp, raster_settings, extrinsic_vector)

    @staticmethod
    def backward(ctx, grad_out_color, *params):
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_scale_factors, grad_rotations), (_, colors_precomp, _m, scales, scale_factors, rotations, cov3Ds_precomp,
                                               sh) = _indexed_backward(ctx, grad_out_color)
        return (grad_means3D, grad_means2D, _fit(grad_sh, sh), None, None, _fit(grad_colors_precomp, colors_precomp),
                grad_opacities, _fit(grad_scales, scales), _fit(grad_scale_factors, scale_factors),
                _fit(grad_rotations, rotations), _fit(grad_cov3Ds_precomp, cov3Ds_precomp), None, None)


def camera_pose_jacobian_sum(means3D, intrinsic, extrinsic_vector, du, dv):
    """grad_mat[7] of the reference's _RasterizeGaussiansIndexedCamera.backward (__init__.py:674-844):
    sum_i grad_params[i,p,0]*du_i + grad_params[i,p,1]*dv_i, with the reference's closed-form
    grad_params written in factored form.  With
        numU, numV, den   the reference's three repeated polynomials in (X,Y,Z,q,t),
    every entry is  A_p * num/den^2 + B_p/den ; the A_p are shared by the u and v columns."""
    X, Y, Z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    fx, fy = intrinsic[0, 0].to(means3D.device), intrinsic[1, 1].to(means3D.device)
    qx, qy, qz, qw, tx, ty, tz = extrinsic_vector.to(means3D.device)
    ix, iy = 1.0 / torch.tan(fx / 2), 1.0 / torch.tan(fy / 2)
    # constants of the reference's znear=0.01 / zfar=100 projection: zfar/(zfar-znear) etc.
    a1, a2, a4 = 1.000100010001, 2.000200020002, 4.000400040004
    b1, b2, b4 = 0.01000100010001, 0.02000200020002, 0.04000400040004
    Xs, Ys = X * ix, Y * iy
    numU = (Xs * (2.0 * qx ** 2 - 4.0 * qx * qy - 2.0 * qz ** 2 + 1.0) + Ys * (-2.0 * qw * qz + 2.0 * qx * qy)
            + Z * (a2 * qw * qy + a2 * qx * qz + tx) - b2 * qw * qy - b2 * qx * qz)
    numV = (Xs * (2.0 * qw * qz + 2.0 * qx * qy) + Ys * (-4.0 * qx * qy + 2.0 * qy ** 2 - 2.0 * qz ** 2 + 1.0)
            + Z * (-a2 * qw * qx + a2 * qy * qz + ty) + b2 * qw * qx - b2 * qy * qz)
    den = (Xs * (-2.0 * qw * qy + 2.0 * qx * qz) + Ys * (2.0 * qw * qx + 2.0 * qy * qz)
           + Z * (-a4 * qx * qy + tz + a1) + b4 * qx * qy - b1)
    inv = 1.0 / den
    ru, rv = numU * inv * inv, numV * inv * inv
    zero = torch.zeros_like(X)
    A = [2.0 * Xs * qy - 2.0 * Ys * qx,
         -2.0 * Xs * qz - 2.0 * Ys * qw + a4 * Z * qy - b4 * qy,
         2.0 * Xs * qw - 2.0 * Ys * qz + a4 * Z * qx - b4 * qx,
         -2.0 * Xs * qx - 2.0 * Ys * qy,
         zero, zero, -Z]
    Bu = [-2.0 * Ys * qz + a2 * Z * qy - b2 * qy,
          Xs * (4.0 * qx - 4.0 * qy) + 2.0 * Ys * qy + a2 * Z * qz - b2 * qz,
          -4.0 * Xs * qx + 2.0 * Ys * qx + a2 * Z * qw - b2 * qw,
          -4.0 * Xs * qz - 2.0 * Ys * qw + a2 * Z * qx - b2 * qx,
          Z, zero, zero]
    Bv = [2.0 * Xs * qz - a2 * Z * qx + b2 * qx,
          2.0 * Xs * qy - 4.0 * Ys * qy - a2 * Z * qw + b2 * qw,
          2.0 * Xs * qx + Ys * (-4.0 * qx + 4.0 * qy) + a2 * Z * qz - b2 * qz,
          2.0 * Xs * qw - 4.0 * Ys * qz + a2 * Z * qy - b2 * qy,
          zero, Z, zero]
    out = torch.zeros(7, dtype=torch.float32, device=means3D.device)
    for p in range(7):
        gu = A[p] * ru + Bu[p] * inv
        gv = A[p] * rv + Bv[p] * inv
        if p == 5:
            gu = zero                      # reference sets grad_params[:,5,0] = 0 and [:,4,1] = 0 explicitly
        if p == 4:
            gv = zero
        out[p] = (gu * du + gv * dv).sum()
    return out


class _RasterizeGaussiansIndexedCamera(torch.autograd.Function):
    """reference __init__.py:537-866: the indexed rasterizer plus the analytic camera-pose gradient.
    The pose gradient is only evaluated when extrinsic_vector requires grad (in finetune.py it does
    not, and the reference computes and discards it)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors, rotations,
                cov3Ds_precomp, raster_settings, extrinsic_vector):
        return _indexed_forward(ctx, means3D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors,
                                rotations, cov3Ds_precomp, raster_settings, extrinsic_vector)

    @staticmethod
    def backward(ctx, grad_out_color, *params):
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_scale_factors, grad_rotations), (extrinsic_vector, colors_precomp, means3D, scales, scale_factors, rotations,
                                               cov3Ds_precomp, sh) = _indexed_backward(ctx, grad_out_color)
        grad_mat = None
        if ctx.needs_input_grad[12]:
            grad_mat = camera_pose_jacobian_sum(means3D, ctx.raster_settings.intrinsic, extrinsic_vector,
                                                grad_means2D[:, 0], grad_means2D[:, 1]).to(extrinsic_vector.device)
        return (grad_means3D, grad_means2D, _fit(grad_sh, sh), None, None, _fit(grad_colors_precomp, colors_precomp),
                grad_opacities, _fit(grad_scales, scales), _fit(grad_scale_factors, scale_factors),
                _fit(grad_rotations, rotations), _fit(grad_cov3Ds_precomp, cov3Ds_precomp), None, grad_mat)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        extrinsic_vector):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, extrinsic_vector)


def rasterize_gaussians_indexed(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors,
                                rotations, cov3Ds_precomp, raster_settings, extrinsic_vector):
    return _RasterizeGaussiansIndexed.apply(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales,
                                            scale_factors, rotations, cov3Ds_precomp, raster_settings, extrinsic_vector)


def rasterize_gaussians_indexed_camera(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales,
                                       scale_factors, rotations, cov3Ds_precomp, raster_settings, extrinsic_vector):
    return _RasterizeGaussiansIndexedCamera.apply(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities,
                                                  scales, scale_factors, rotations, cov3Ds_precomp, raster_settings,
                                                  extrinsic_vector)


# ----------------------------------------------------------------------------- public modules
class GaussianRasterizationSettings(NamedTuple):
    """reference __init__.py:868-878."""
    intrinsic: torch.Tensor
    extrinsic_vector: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    sh_degree: int
    prefiltered: bool
    debug: bool
    clamp_color: bool


def _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")


def _empty():
    return torch.Tensor([])


class GaussianRasterizer(nn.Module):
    """reference __init__.py:881-948. `forward` accepts the pose as `extrinsic_vector=` (what the reference's own
    caller passes, scene/gaussian_model.py:874) and, for compatibility with the declared signature, `extrinsic=`."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions, extrinsic_vector):
        with torch.no_grad():
            present = _mark_visible_from_pose(positions, extrinsic_vector)
            if present is not None:
                return present
            view, proj = camera_matrices(self.raster_settings.intrinsic, extrinsic_vector, positions.device)[:2]
            return _C.mark_visible(positions, view, proj)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extrinsic_vector=None, extrinsic=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        if extrinsic_vector is None:
            extrinsic_vector = extrinsic if extrinsic is not None else self.raster_settings.extrinsic_vector
        shs = _empty() if shs is None else shs
        colors_precomp = _empty() if colors_precomp is None else colors_precomp
        scales = _empty() if scales is None else scales
        rotations = _empty() if rotations is None else rotations
        cov3D_precomp = _empty() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, extrinsic_vector)


class GaussianRasterizerIndexed(nn.Module):
    """reference __init__.py:951-1046."""

    def __init__(self, raster_settings, optimize_camera=False):
        super().__init__()
        self.raster_settings = raster_settings
        self.optimize_camera = optimize_camera

    def markVisible(self, positions, extrinsic_vector):
        with torch.no_grad():
            present = _mark_visible_from_pose(positions, extrinsic_vector)
            if present is not None:
                return present
            view, proj = camera_matrices(self.raster_settings.intrinsic, extrinsic_vector, positions.device)[:2]
            return _C.mark_visible(positions, view, proj)

    def forward(self, means3D, means2D, opacities, sh_indices, g_indices, shs=None, colors_precomp=None, scales=None,
                scale_factors=None, rotations=None, cov3D_precomp=None, extrinsic_vector=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        if extrinsic_vector is None:
            extrinsic_vector = self.raster_settings.extrinsic_vector
        shs = _empty() if shs is None else shs
        colors_precomp = _empty() if colors_precomp is None else colors_precomp
        scales = _empty() if scales is None else scales
        scale_factors = _empty() if scale_factors is None else scale_factors
        rotations = _empty() if rotations is None else rotations
        cov3D_precomp = _empty() if cov3D_precomp is None else cov3D_precomp
        fn = rasterize_gaussians_indexed_camera if self.optimize_camera else rasterize_gaussians_indexed
        return fn(means3D, means2D, shs, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors, rotations,
                  cov3D_precomp, self.raster_settings, extrinsic_vector)
