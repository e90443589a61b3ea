"""Lane efficiency of the two blend kernels on the bench workload (configs[2]: 3M Gaussians, 1920x1080, indexed), measured
with the "lanes" build variant of the library (render.hip compiled with -DC3DGS_COUNT_LANES; c3dgs_amd/build.py).

    C3DGS_LIB_PATH=c3dgs_amd/libc3dgs_hip_lanes.so python tools/lane_efficiency.py [out.txt]

What it says: of the 64 pixel lanes a wave spends on every (wave, Gaussian) pair that survives the per-quadrant culling, how
many USE the pair (forward: blend it; backward: alpha >= 1/255 and in front of the pixel's last contributor) -- and how many
loop iterations a finer scheduling unit (an 8x4-pixel half wave with its own list; a 4x4-pixel 16-lane row) would run."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from c3dgs_amd import _lib
from c3dgs_amd import rasterizer as rz

assert "lanes" in _lib.LIB_PATH, "run with C3DGS_LIB_PATH=c3dgs_amd/libc3dgs_hip_lanes.so"
dev = torch.device("cuda", 0)
lines = []
for name, P, scale_sigma in (("configs[2] 3M synth-v1", 3_000_000, None), ("1M synth-v1", 1_000_000, None), ("6M synth-v1", 6_000_000, None)):
    W, H = 1920, 1080
    intr, ev, t, dL, ix = bench.build_workload(P, W, H, 1200.0, dev)
    view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
    E = torch.Tensor([])
    bg = torch.zeros(3, device=dev)
    L = _lib.lib()
    out = (C.c_uint64 * 16)()
    L.c3dgs_debug_lane_counters(out, None)          # clear
    o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E,
                                          view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
    g = rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E,
                                                   view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False,
                                                   t["sh_indices"], t["g_indices"])
    torch.cuda.synchronize()
    L.c3dgs_debug_lane_counters(out, None)
    v = [int(x) for x in out]
    lines.append(f"== {name}: P={P} R={o[0]} {W}x{H}")
    for tag, c in (("render_forward ", v[:8]), ("render_backward", v[8:])):
        pairs, slots, lanes, live, half, blk, lists, aux = c
        if pairs == 0:
            lines.append(f"{tag}: no counts (not the lanes build?)")
            continue
        lines.append(f"{tag}: pairs={pairs} slots_incl_padding={slots} ({slots / pairs:.3f}x) lists={lists} "
                     f"useful_lanes_per_pair={lanes / pairs:.2f}/64 ({100 * lanes / pairs / 64:.1f} %) pairs_with_a_useful_lane={100 * live / pairs:.1f} %"
                     + (f" hit_lanes_incl_finished_pixels={aux / pairs:.2f}" if aux else ""))
        lines.append(f"{tag}: iterations now={pairs}; with 8x4 half-wave units={half} ({half / pairs:.3f}x); with 4x4 units={blk} ({blk / pairs:.3f}x)"
                     "   [upper bounds of the gain: counted on pixels that USE the pair, a conservative culling test keeps more]")
    del t, dL, ix, o, g
    torch.cuda.empty_cache()
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
