"""Helpers for the -m gpu parity tests: run the HIP path through its C-ABI front-end and unpack the opaque
scratch buffers (layouts from c3dgs_get_*_layout) into numpy arrays comparable with the oracle's RasterState."""
import ctypes as C

import numpy as np
import torch

from c3dgs_amd import _lib, rasterizer


def _view(buf, off, count, dtype):
    nbytes = count * torch.tensor([], dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype)


def to_dev(x, dev="cuda"):
    if x is None:
        return torch.Tensor([])
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(dev)


def hip_forward(inp, cam, indexed=False, dev="cuda"):
    """inp: dict of CPU tensors / None (bg, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
    scale_factors, sh_indices, g_indices, degree, scale_modifier, prefiltered, clamp_color); cam: oracle.camera() dict."""
    g = lambda k: to_dev(inp.get(k), dev)
    view, proj, campos = to_dev(cam["viewmatrix"], dev), to_dev(cam["projmatrix"], dev), to_dev(cam["campos"], dev)
    common_tail = (view, proj, cam["tan_fovx"], cam["tan_fovy"], cam["H"], cam["W"], g("shs"), int(inp.get("degree", 0)), campos)
    flags = (bool(inp.get("prefiltered", False)), False, bool(inp.get("clamp_color", True)))
    if indexed:
        args = (g("bg"), g("means3D"), g("colors_precomp"), g("opacities"), g("scales"), g("scale_factors"), g("rotations"),
                float(inp.get("scale_modifier", 1.0)), g("cov3D_precomp")) + common_tail + (g("sh_indices"), g("g_indices")) + flags
        out = rasterizer._C.rasterize_gaussians_indexed(*args)
    else:
        args = (g("bg"), g("means3D"), g("colors_precomp"), g("opacities"), g("scales"), g("rotations"),
                float(inp.get("scale_modifier", 1.0)), g("cov3D_precomp")) + common_tail + flags
        out = rasterizer._C.rasterize_gaussians(*args)
    return dict(args=args, num_rendered=out[0], color=out[1], radii=out[2], geom=out[3], binning=out[4], img=out[5],
                indexed=indexed, W=cam["W"], H=cam["H"])


def hip_backward(fw, dL_dout, dev="cuda"):
    a = fw["args"]
    if fw["indexed"]:
        (bg, means3D, colors, opac, scales, sf, rot, smod, cov3D, view, proj, tfx, tfy, H, W, sh, deg, campos, shi, gi,
         pre, dbg, clamp) = a
        out = rasterizer._C.rasterize_gaussians_backward_indexed(
            bg, means3D, fw["radii"], colors, scales, sf, rot, smod, cov3D, view, proj, tfx, tfy, to_dev(dL_dout, dev), sh, deg,
            campos, fw["geom"], fw["num_rendered"], fw["binning"], fw["img"], False, shi, gi)
        names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                 "dL_dscale_factors", "dL_drotations")
    else:
        (bg, means3D, colors, opac, scales, rot, smod, cov3D, view, proj, tfx, tfy, H, W, sh, deg, campos, pre, dbg, clamp) = a
        out = rasterizer._C.rasterize_gaussians_backward(
            bg, means3D, fw["radii"], colors, scales, rot, smod, cov3D, view, proj, tfx, tfy, to_dev(dL_dout, dev), sh, deg,
            campos, fw["geom"], fw["num_rendered"], fw["binning"], fw["img"], False)
        names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                 "dL_drotations")
    torch.cuda.synchronize()
    return {n: o.detach().cpu().numpy() for n, o in zip(names, out)}


def unpack(fw):
    """-> dict of numpy arrays named like oracle.RasterState attributes."""
    L = _lib.lib()
    P = int(fw["radii"].numel())
    R, W, H = fw["num_rendered"], fw["W"], fw["H"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(radii=fw["radii"].cpu().numpy(), out_color=fw["color"].cpu().numpy(), num_rendered=R)
    il = _lib.ImageLayout()
    L.c3dgs_get_image_layout(W, H, C.byref(il))
    img = fw["img"]
    out["final_T"] = _view(img, il.final_T, W * H, torch.float32).cpu().numpy()
    out["n_contrib"] = _view(img, il.n_contrib, W * H, torch.int32).cpu().numpy().view(np.uint32)
    out["ranges"] = _view(img, il.ranges, 2 * T, torch.int32).cpu().numpy().view(np.uint32).reshape(T, 2)
    if P > 0:
        gl = _lib.GeomLayout()
        L.c3dgs_get_geom_layout(P, C.byref(gl))
        geom = fw["geom"]
        splat = _view(geom, gl.splat, 12 * P, torch.float32).cpu().numpy().reshape(P, 12)
        out["means2D"] = splat[:, 0:2].copy()
        out["conic_opacity"] = np.stack([splat[:, 2], splat[:, 3], splat[:, 4], splat[:, 5]], 1)
        out["rgb"] = np.stack([splat[:, 6], splat[:, 7], splat[:, 8]], 1)
        # depth_keys = bits of the view-space depth for visible Gaussians (0xFFFFFFFF = culled); tiles_touched = rect area
        out["depths"] = _view(geom, gl.depth_keys, P, torch.float32).cpu().numpy()
        rects = _view(geom, gl.rects, 4 * P, torch.int16).cpu().numpy().view(np.uint16).reshape(P, 4).astype(np.int64)
        out["tiles_touched"] = ((rects[:, 2] - rects[:, 0]) * (rects[:, 3] - rects[:, 1])).astype(np.uint32)
        out["depth_order"] = _view(geom, gl.depth_order, P, torch.int32).cpu().numpy().view(np.uint32)
        out["inst_offset"] = _view(geom, gl.inst_offset, P, torch.int32).cpu().numpy().view(np.uint32)
        cl = _view(geom, gl.clamped, P, torch.uint8).cpu().numpy()
        out["clamped"] = np.stack([(cl >> c) & 1 for c in range(3)], 1).astype(np.uint8)
    if R > 0:
        bl = _lib.BinningLayout()
        L.c3dgs_get_binning_layout(R, W, H, C.byref(bl))
        b = fw["binning"]
        # the library keeps 16-bit tile keys; the reference's 64-bit key is (tile << 32) | depth bits of the Gaussian
        dbits = out["depths"].view(np.uint32).astype(np.uint64)
        if T > 65536:                                     # more than 65,536 tiles: the library switches to 32-bit tile keys
            tk_u = _view(b, bl.keys_unsorted, R, torch.int32).cpu().numpy().view(np.uint32).astype(np.uint64)
            tk_s = _view(b, bl.keys_sorted, R, torch.int32).cpu().numpy().view(np.uint32).astype(np.uint64)
        else:
            tk_u = _view(b, bl.keys_unsorted, R, torch.int16).cpu().numpy().view(np.uint16).astype(np.uint64)
            tk_s = _view(b, bl.keys_sorted, R, torch.int16).cpu().numpy().view(np.uint16).astype(np.uint64)
        out["values_unsorted"] = _view(b, bl.values_unsorted, R, torch.int32).cpu().numpy().view(np.uint32)
        out["point_list"] = _view(b, bl.point_list, R, torch.int32).cpu().numpy().view(np.uint32)
        out["keys_unsorted"] = (tk_u << np.uint64(32)) | dbits[out["values_unsorted"]]
        out["keys_sorted"] = (tk_s << np.uint64(32)) | dbits[out["point_list"]]
    return out


def np_inputs(inp):
    """numpy view of the same inputs for the oracle."""
    o = {}
    for k, v in inp.items():
        o[k] = v.numpy() if isinstance(v, torch.Tensor) else v
    return o


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 200.0 if mse == 0 else 20 * np.log10(1.0) - 10 * np.log10(mse)   # utils/image_utils.py:17-19 (peak 1.0)


def rel_inf(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0
