"""-m gpu: the hand-written stable onesweep radix sort of the binning stage (csrc/radix_sort.hip) against torch's stable
sort: ragged sizes around the 8192-item tile, every digit split the tile sort can take, heavy ties, both key widths."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(keys, values, end_bit):
    from c3dgs_amd import _lib
    L = _lib.lib()
    kb = keys.element_size()
    n = keys.numel()
    tb = int(L.c3dgs_debug_sort_temp_bytes(kb, n, end_bit))
    temp = torch.empty(max(tb, 256), dtype=torch.uint8, device="cuda")
    ko, vo = torch.empty_like(keys), torch.empty_like(values)
    _lib.check(L.c3dgs_debug_sort_pairs(kb, n, end_bit, keys.data_ptr(), ko.data_ptr(), values.data_ptr(), vo.data_ptr(),
                                        temp.data_ptr(), tb, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return ko, vo


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 8191, 8192, 8193, 16384, 100_003, 1_000_000, 3_000_001])
def test_tile_key_sort_is_stable_for_ragged_sizes(n):
    g = torch.Generator(device="cuda").manual_seed(n)
    for end_bit, hi in ((13, 8160), (7, 100), (16, 65536), (9, 300), (1, 2)):
        keys = torch.randint(0, min(hi, 1 << end_bit), (n,), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
        vals = torch.arange(n, device="cuda", dtype=torch.int32)
        ko, vo = _sort(keys, vals, end_bit)
        k32 = keys.to(torch.int32) & 0xffff
        want = torch.sort(k32, stable=True)
        assert torch.equal(vo.long(), want.indices), (n, end_bit)
        assert torch.equal(ko.to(torch.int32) & 0xffff, want.values), (n, end_bit)


@pytest.mark.parametrize("n", [1, 511, 8192, 8193, 300_000, 3_000_000])
def test_depth_key_sort_matches_stable_sort(n):
    g = torch.Generator(device="cuda").manual_seed(n + 7)
    depth = torch.rand(n, device="cuda", generator=g) * 10 + 2
    depth[torch.rand(n, device="cuda", generator=g) < 0.06] = float("nan")          # culled: key 0xFFFFFFFF
    keys = depth.view(torch.int32).clone()
    keys[torch.isnan(depth)] = -1
    keys[::7] = int(keys[0])                                                       # heavy ties
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ko, vo = _sort(keys, vals, 32)
    ku = keys.to(torch.int64) & 0xffffffff
    want = torch.sort(ku, stable=True)
    assert torch.equal(vo.long(), want.indices)
    assert torch.equal(ko.to(torch.int64) & 0xffffffff, want.values)


def test_sort_of_already_sorted_and_constant_keys():
    n = 50_000
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    const = torch.full((n,), 4321, device="cuda", dtype=torch.int16)
    ko, vo = _sort(const, vals, 13)
    assert torch.equal(vo, vals) and torch.equal(ko, const)
    asc = (torch.arange(n, device="cuda") * 8160 // n).to(torch.int16)
    ko, vo = _sort(asc, vals, 13)
    assert torch.equal(vo, vals) and torch.equal(ko, asc)


def test_rocprim_fallback_gives_the_same_sorted_lists():
    """C3DGS_SORT_ROCPRIM=1 (also taken automatically from 2^30 items on) routes both sorts through rocPRIM: the raster
    parity cases that compare the sorted keys / point list bit-exactly must pass unchanged. Run in a child process
    because the switch is read once per process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, C3DGS_SORT_ROCPRIM="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_raster_gpu.py", "tests/test_sort_gpu.py", "-q", "-m", "gpu", "-x",
                        "-k", "(forward_parity and (base or wide_depth or p8193 or equal_depth or indexed)) or ragged or stable_sort"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_depth_sort_gather_path_gives_the_same_sorted_lists():
    """By default the depth sort carries every Gaussian's tile rectangle through its passes as a packed second payload (grids up
    to 255 x 255 tiles); C3DGS_DEPTH_SORT_GATHER=1 makes it gather the rectangles behind the last pass instead (what larger
    grids get). The parity cases that compare keys / point lists / ranges bit-exactly must pass on that path too. Child process:
    the switch is read once per process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, C3DGS_DEPTH_SORT_GATHER="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_raster_gpu.py", "-q", "-m", "gpu", "-x",
                        "-k", "forward_parity and (base or wide_depth or p8193 or p12289 or equal_depth or indexed or huge_splats)"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_TIMEOUT_CHILD = r"""
import numpy as np, torch, sys
from tests import cases, gpu_util, synth
from oracle import oracle as orc
from c3dgs_amd import _lib
assert _lib.LIB_PATH.endswith("libc3dgs_hip_spin1.so"), _lib.LIB_PATH
intr, ev = synth.camera(640, 360, 400.0)
cam = orc.camera(intr.numpy(), ev.numpy())
sc = synth.scene(400_000, 640, 360, 400.0, seed=3, scale_median=0.02)
inp = dict(bg=torch.zeros(3), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
           rotations=sc["rotations"], degree=3, clamp_color=True)
fw = gpu_util.hip_forward(inp, cam, False)           # 49 depth-sort tiles, hundreds of tile-sort tiles: look-backs give up
torch.cuda.synchronize()
assert torch.isnan(fw["color"]).all(), "a forward whose sort timed out must return a NaN image"
try:
    gpu_util.hip_forward(inp, cam, False)
except RuntimeError as e:
    assert "look-back timed out in an earlier rasterizer call" in str(e), str(e)
    print("RAISED")
else:
    sys.exit("the call after a timed-out sort did not fail")
# debug mode reports it in the same call
a = list(fw["args"]); a[-2] = True
from c3dgs_amd import rasterizer as rz
try:
    rz._C.rasterize_gaussians(*a)
except RuntimeError as e:
    assert "look-back timed out" in str(e), str(e)
    print("DEBUG_RAISED")
"""


def test_sort_timeout_is_not_silent():
    """A look-back that gives up (pre-empted / dead predecessor) must not yield a plausible wrong image outside debug mode:
    with the `spin1` variant of the library (C3DGS_OS_SPIN_LIMIT=1: every look-back gives up after one poll) the forward
    returns a NaN image, the next forward fails with C3DGS_E_HIP, and debug mode fails in the same call."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "c3dgs_amd", "libc3dgs_hip_spin1.so")
    if not os.path.exists(lib):
        from c3dgs_amd import build
        build.build_variant("spin1")
    r = subprocess.run([sys.executable, "-c", _TIMEOUT_CHILD], cwd=root, env=dict(os.environ, C3DGS_LIB_PATH=lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RAISED" in r.stdout and "DEBUG_RAISED" in r.stdout, r.stdout
