"""ctypes binding of libc3dgs_hip.so (include/c3dgs_hip.h).  No CPU fallback: if the HIP library is
missing or cannot be loaded, importing any op raises immediately."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# C3DGS_LIB_PATH: test hook only -- loads a variant build of the SAME library (c3dgs_amd/build.py VARIANTS)
LIB_PATH = os.environ.get("C3DGS_LIB_PATH") or os.path.join(_HERE, "libc3dgs_hip.so")

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class RasterParams(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("SHS", C.c_int32), ("GS", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("sh", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("scale_factors", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("sh_indices", C.c_void_p), ("g_indices", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("clamp_color", C.c_int32), ("debug", C.c_int32),
    ]


class RasterGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
        "dL_dscale_factors", "dL_drotations")]


class GeomLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "total_bytes", "splat", "depth_keys", "depth_keys_sorted", "depth_order",
        "sorted_offsets", "inst_offset", "rects", "clamped", "scan_temp", "scan_temp_bytes", "block_base", "depth_base")]


class BinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "total_bytes", "keys_unsorted", "values_unsorted", "keys_sorted", "point_list", "sort_temp", "sort_temp_bytes")]


class StageTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("count", C.c_int64)]


class ImageLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("total_bytes", "final_T", "n_contrib", "ranges", "tile_used", "tile_order")]


class QatParams(C.Structure):
    _fields_ = [("P", C.c_int32), ("GS", C.c_int32), ("SHS", C.c_int32), ("M", C.c_int32),
                ("xyz", C.c_void_p), ("opacity", C.c_void_p), ("scaling_factor", C.c_void_p), ("scaling", C.c_void_p),
                ("rotation", C.c_void_p), ("features_dc", C.c_void_p), ("features_rest", C.c_void_p), ("state", C.c_void_p),
                ("observer_enabled", C.c_int32 * 6), ("fake_quant_enabled", C.c_int32 * 6),
                ("half_xyz", C.c_int32), ("averaging_constant", C.c_float)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("step_size", C.c_float), ("bias_correction2_sqrt", C.c_float)]


_vp = C.c_void_p
# name -> (restype, argtypes); every symbol include/c3dgs_hip.h declares
PROTOTYPES = {
    "c3dgs_camera_from_pose": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_mark_visible_pose": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_rasterize_gaussians": (C.c_int, [C.POINTER(RasterParams), RESIZE_FN, C.c_void_p, RESIZE_FN, C.c_void_p,
                                            RESIZE_FN, C.c_void_p, C.c_void_p, C.c_void_p, _i32p, C.c_void_p]),
    "c3dgs_rasterize_gaussians_indexed": (C.c_int, [C.POINTER(RasterParams), RESIZE_FN, C.c_void_p, RESIZE_FN, C.c_void_p,
                                                    RESIZE_FN, C.c_void_p, C.c_void_p, C.c_void_p, _i32p, C.c_void_p]),
    "c3dgs_rasterize_gaussians_backward": (C.c_int, [C.POINTER(RasterParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int32, C.c_void_p, RESIZE_FN, C.c_void_p,
                                                     C.POINTER(RasterGrads), C.c_void_p]),
    "c3dgs_rasterize_gaussians_backward_indexed": (C.c_int, [C.POINTER(RasterParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                                             C.c_void_p, C.c_int32, C.c_void_p, RESIZE_FN, C.c_void_p,
                                                             C.POINTER(RasterGrads), C.c_void_p]),
    "c3dgs_weighted_distance": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "c3dgs_vq_accumulate": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_weighted_distance_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "c3dgs_weighted_distance_ws": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c3dgs_debug_wd_scores": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_vq_sums": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c3dgs_vq_step_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "c3dgs_vq_step_sums": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c3dgs_vq_step_apply": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                      C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "c3dgs_vq_apply": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_int32, C.c_void_p]),
    "c3dgs_mt19937_fill": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64]),
    "c3dgs_draws_to_indices": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_host_buffer_device_address": (C.c_void_p, [C.c_void_p]),
    "c3dgs_draws_upload": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_morton_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "c3dgs_morton_order": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c3dgs_adam_step": (C.c_int, [C.c_int32, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double, _vp]),
    "c3dgs_extract_rot_scale": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp]),
    "c3dgs_l1_ssim_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "c3dgs_l1_ssim_value": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "c3dgs_l1_ssim_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "c3dgs_qat_workspace_bytes": (C.c_size_t, []),
    "c3dgs_qat_scan_bytes": (C.c_size_t, [C.c_int32]),
    "c3dgs_qat_observe": (C.c_int, [C.POINTER(QatParams), _vp, _vp]),
    "c3dgs_qat_codebooks": (C.c_int, [C.POINTER(QatParams), _vp, _vp, _vp, _vp]),
    "c3dgs_qat_codebooks_backward": (C.c_int, [C.POINTER(QatParams)] + [_vp] * 8),
    "c3dgs_qat_visible": (C.c_int, [C.POINTER(QatParams), _vp, _vp, _vp, _vp, _vp, _vp]),
    "c3dgs_qat_points": (C.c_int, [C.POINTER(QatParams)] + [_vp] * 10),
    "c3dgs_qat_points_backward": (C.c_int, [C.POINTER(QatParams)] + [_vp] * 11),
    "c3dgs_qat_quantize": (C.c_int, [C.POINTER(QatParams), C.c_int32] + [_vp] * 7),
    "c3dgs_fake_quantize": (C.c_int, [C.c_int64, _vp, _vp, C.c_int32, C.c_int32, C.c_float, _vp, _vp, _vp]),
    "c3dgs_fake_quantize_backward": (C.c_int, [C.c_int64, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "c3dgs_debug_lane_counters": (C.c_int, [C.POINTER(C.c_uint64), _vp]),
    "c3dgs_debug_sort_times": (C.c_int, [C.POINTER(C.c_uint64)]),
    "c3dgs_debug_gather_probe": (C.c_int, [C.c_int32, C.c_int64, _vp, _vp, _vp, _vp]),
    "c3dgs_debug_sort_temp_bytes": (C.c_size_t, [C.c_int32, C.c_int64, C.c_int32]),
    "c3dgs_debug_sort_pairs": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "c3dgs_get_geom_layout": (C.c_int, [C.c_int32, C.POINTER(GeomLayout)]),
    "c3dgs_get_binning_layout": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(BinningLayout)]),
    "c3dgs_get_image_layout": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(ImageLayout)]),
    "c3dgs_backward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "c3dgs_profile_enable": (C.c_int, [C.c_int]),
    "c3dgs_profile_only": (C.c_int, [C.c_char_p]),
    "c3dgs_profile_read": (C.c_int, [C.POINTER(StageTime), C.c_int]),
    "c3dgs_last_error": (C.c_char_p, []),
    "c3dgs_abi_version": (C.c_int, []),
    "c3dgs_abs_accumulate": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    """Load the HIP library (once). Raises RuntimeError loudly when it is absent: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"c3dgs_amd: {LIB_PATH} is missing. Build it with `python -m c3dgs_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this package.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"c3dgs_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().c3dgs_last_error()
        raise RuntimeError((msg or b"unknown error").decode("utf-8", "replace"))


def profile_enable(on=True, only=None):
    """Bracket stage launches with HIP events; `only` = one stage name to keep the queue cost to that kernel."""
    check(lib().c3dgs_profile_only(only.encode() if only else None))
    lib().c3dgs_profile_enable(1 if on else 0)


def profile_read():
    """-> {stage: (total_ms, count)} since the last read (synchronises the recorded events)."""
    arr = (StageTime * 48)()
    n = lib().c3dgs_profile_read(arr, 48)
    return {arr[i].name.decode(): (float(arr[i].total_ms), int(arr[i].count)) for i in range(n)}
