"""Where render_backward's waves spend their time on the bench workload: shader-clock ticks per phase, summed over all waves
("bwdtime" build variant).   C3DGS_LIB_PATH=c3dgs_amd/libc3dgs_hip_bwdtime.so python tools/bwd_phases.py [out.txt]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from c3dgs_amd import _lib
from c3dgs_amd import rasterizer as rz
assert "bwdtime" in _lib.LIB_PATH
dev = torch.device("cuda", 0)
P, W, H = 3_000_000, 1920, 1080
intr, ev, t, dL, ix = bench.build_workload(P, W, H, 1200.0, dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
E, bg = torch.Tensor([]), torch.zeros(3, device=dev)
L = _lib.lib()
out = (C.c_uint64 * 16)()
for it in range(3):
    L.c3dgs_debug_lane_counters(out, None)
    o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy,
                                          H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
    g = rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy,
                                                   dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
    torch.cuda.synchronize()
L.c3dgs_debug_lane_counters(out, None)
fine = [int(x) for x in out][:5]
v = [int(x) for x in out][8:]
stage, lst, loop, flush, pro, waves, total = v[:7]
txt = [f"render_backward on configs[2] (3M, 1920x1080, R={o[0]}): shader-clock ticks summed over {waves} waves that had work",
       f"  wave lifetime total {total:.3e} ticks = 100 %"]
for name, x in (("prologue (pixel state, wave_last)", pro), ("staging of a batch incl. its two barriers", stage), ("candidate-list compaction", lst),
                ("group loop (alpha, body, reduction, LDS adds)", loop), ("flush incl. barrier (LDS planes -> partial-sum slots)", flush)):
    txt.append(f"  {name:52s} {100.0 * x / total:5.1f} %")
for name, x in (("  staging: barrier at the top of the round", fine[0]), ("  staging: point list -> splat record -> LDS (dependent loads)", fine[1]),
                ("  staging: clearing the LDS planes", fine[2]), ("  staging: closing barrier", fine[3]), ("  flush: its barrier (slowest wave's group loop)", fine[4])):
    txt.append(f"  {name:60s} {100.0 * x / total:5.1f} %")
txt.append(f"  unaccounted (loop control between the stamps)        {100.0 * (total - stage - lst - loop - flush - pro) / total:5.1f} %")
print("\n".join(txt))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(txt) + "\n")
