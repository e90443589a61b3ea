"""Per-call Lloyd-loop time of back-to-back vq_features calls, with collector events and device-allocator activity (how the
pinned-ring stalls were found: sporadic 50-80 ms inside a later call while page-locked buffers were allocated per call). python tools/dbg_vq_stalls.py"""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import vq as vqm
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
N, D, K, B = 5_400_000, 48, 4096, 2 ** 18
feats = torch.randn(N, D, device=dev, generator=g) * 0.1
imp = torch.rand(N, device=dev, generator=g).pow(4)
ev = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: ev.append((info["generation"], info["collected"], (time.perf_counter() - cb.t) * 1e3))
gc.callbacks.append(cb)
for rep, steps in enumerate([3, 1, 30, 30, 30, 0, 30]):
    ev.clear()
    st = {}
    m0 = torch.cuda.memory_stats(dev)
    a0 = (m0.get("num_device_alloc", 0), m0.get("num_device_free", 0), m0.get("num_alloc_retries", 0))
    vqm.vq_features(feats, imp, K, B, steps, silent=True, stats=st)
    m1 = torch.cuda.memory_stats(dev)
    print("   device allocs/frees/retries in this call:", m1.get("num_device_alloc", 0) - a0[0], m1.get("num_device_free", 0) - a0[1], m1.get("num_alloc_retries", 0) - a0[2],
          "reserved MB", m1["reserved_bytes.all.current"] >> 20, "gc counts", gc.get_count())
    print(f"call {rep}: {steps} steps, lloyd {st['lloyd_seconds']*1e3:.2f} ms ({st['lloyd_seconds']*1e3/max(steps,1):.3f} per step), final {st['final_assignment_seconds']*1e3:.1f} ms; gc events (gen, collected, ms): {[(a, b, round(c, 2)) for a, b, c in ev]}", flush=True)
