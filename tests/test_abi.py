"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/c3dgs_hip.h declares, and its
argument validation (no device work) returns the documented codes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from c3dgs_amd import build, _lib
    build.build()
    return _lib.lib()


def _declared_symbols():
    h = open(os.path.join(ROOT, "include", "c3dgs_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(c3dgs_[a-z0-9_]+)\s*\(", h)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from c3dgs_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/c3dgs_hip.h but not exported"
        assert s in _lib.PROTOTYPES, f"{s} has no ctypes prototype"
    assert L.c3dgs_abi_version() == 4


def test_layouts_are_consistent(L):
    from c3dgs_amd import _lib
    g = _lib.GeomLayout()
    assert L.c3dgs_get_geom_layout(1000, C.byref(g)) == 0
    offs = [g.splat, g.depth_keys, g.depth_keys_sorted, g.depth_order, g.sorted_offsets,
            g.inst_offset, g.rects, g.clamped, g.scan_temp]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and g.total_bytes > g.scan_temp
    assert g.depth_keys - g.splat >= 1000 * 48
    b = _lib.BinningLayout()
    assert L.c3dgs_get_binning_layout(5000, 1920, 1080, C.byref(b)) == 0
    assert b.values_unsorted - b.keys_unsorted >= 5000 * 2 and b.total_bytes > b.sort_temp
    im = _lib.ImageLayout()
    assert L.c3dgs_get_image_layout(1920, 1080, C.byref(im)) == 0
    assert im.n_contrib - im.final_T >= 1920 * 1080 * 4 and im.tile_used - im.ranges >= 8160 * 8
    assert L.c3dgs_backward_workspace_bytes(10, 1000) >= 1000 * 36
    assert L.c3dgs_get_geom_layout(-1, C.byref(g)) == 1


def test_validation_without_gpu(L):
    from c3dgs_amd import _lib
    # N == 0 is legal and touches nothing
    assert L.c3dgs_weighted_distance(0, 4, 6, None, None, None, None, None, None) == 0
    assert L.c3dgs_weighted_distance(10, 4, 6, None, None, None, None, None, None) == 1
    assert b"dimension 2" in L.c3dgs_last_error()
    assert L.c3dgs_mark_visible(0, None, None, None, None, None) == 0
    assert L.c3dgs_mark_visible(5, None, None, None, None, None) == 1
    assert L.c3dgs_mark_visible_pose(0, None, None, None, None) == 0
    assert L.c3dgs_mark_visible_pose(5, None, None, None, None) == 1
    p = _lib.RasterParams()
    p.P, p.W, p.H = 4, 64, 64
    n = C.c_int32(0)
    cb = _lib.RESIZE_FN(lambda u, b: 0)
    rc = L.c3dgs_rasterize_gaussians(C.byref(p), cb, None, cb, None, cb, None, None, None, C.byref(n), None)
    assert rc == 1 and b"means3D must have dimensions" in L.c3dgs_last_error()
    assert L.c3dgs_vq_apply(0, 4, None, None, None, 0.8, 0.2, 1e-5, 0, None) == 1
    st = (_lib.StageTime * 4)()
    assert L.c3dgs_profile_read(st, 4) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from c3dgs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_image_size_limits(L):
    """More than 65,536 tiles is no longer a limit (32-bit tile keys take over, tests/test_raster_gpu.py::test_8k_image_...); what is
    rejected up front: tile coordinates beyond 16 bits and (tiles per row)^2 x (tile rows) >= 2^32 (the pair emission's exact
    multiply-high division). Checked without a GPU: validation runs before any launch, the callbacks refuse to allocate."""
    from c3dgs_amd import _lib
    one = (C.c_float * 16)()

    def call(W, H):
        p = _lib.RasterParams()
        p.P, p.W, p.H = 4, W, H
        for name in ("background", "means3D", "sh", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos"):
            setattr(p, name, C.addressof(one))
        p.M, p.D = 16, 3
        n = C.c_int32(0)
        cb = _lib.RESIZE_FN(lambda u, b: 0)
        rc = L.c3dgs_rasterize_gaussians(C.byref(p), cb, None, cb, None, cb, None, C.addressof(one), C.addressof(one), C.byref(n), None)
        return rc, L.c3dgs_last_error()

    rc, msg = call(7680, 4320)                           # 480 x 270 = 129,600 tiles: accepted (fails later, at the refused allocation)
    assert rc != 0 and b"allocation failed" in msg, msg
    rc, msg = call(40_000, 30_000)                       # 2500^2 x 1875 tiles^3 > 2^32
    assert rc == 1 and b"image too large" in msg, msg
    rc, msg = call(16 * 65536, 16)
    assert rc == 1 and b"16-bit tile coordinates" in msg, msg
    il = _lib.ImageLayout()
    assert L.c3dgs_get_image_layout(7680, 4320, C.byref(il)) == 0 and il.tile_order >= il.tile_used + 4 * 129_600
    b2, b4 = _lib.BinningLayout(), _lib.BinningLayout()
    L.c3dgs_get_binning_layout(1000, 4096, 4096, C.byref(b2))        # 65,536 tiles: 16-bit keys
    L.c3dgs_get_binning_layout(1000, 4112, 4096, C.byref(b4))        # 65,792 tiles: 32-bit keys
    assert b2.values_unsorted - b2.keys_unsorted < b4.values_unsorted - b4.keys_unsorted
