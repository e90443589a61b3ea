"""Wall clock vs summed stage times of the direct C-ABI fwd+bwd at several sizes in one process (debug aid)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import rasterizer as rz, _lib
from tests import synth
dev = torch.device("cuda", 0)
E = torch.Tensor([])
W, H, focal = 1920, 1080, 1200.0
bg = torch.zeros(3, device=dev)
dL = synth.grad_image(W, H).to(dev)
intr, ev = synth.camera(W, H, focal)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, ev.to(dev), dev)
for P in [int(x) for x in sys.argv[1:]]:
    sc = synth.scene(P, W, H, focal)
    t = {k: v.to(dev) for k, v in synth.index_scene(sc).items()}

    def one():
        o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
        rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
    for _ in range(6):
        one()
    torch.cuda.synchronize()
    na0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    per = []
    for _ in range(20):
        a = time.perf_counter()
        one()
        per.append(round((time.perf_counter() - a) * 1e3, 2))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    print("timed loop: host ms per iteration", per, "device allocs", torch.cuda.memory_stats()["num_device_alloc"] - na0, flush=True)
    ms0 = dict(torch.cuda.memory_stats())
    its = []
    for _ in range(12):
        a = time.perf_counter()
        o = rz._C.rasterize_gaussians_indexed(bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
        b = time.perf_counter()
        torch.cuda.synchronize()
        c = time.perf_counter()
        rz._C.rasterize_gaussians_backward_indexed(bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
        d = time.perf_counter()
        torch.cuda.synchronize()
        e = time.perf_counter()
        its.append((round((b - a) * 1e3, 2), round((c - b) * 1e3, 2), round((d - c) * 1e3, 2), round((e - d) * 1e3, 2)))
    ms1 = torch.cuda.memory_stats()
    print("fwd host / fwd drain / bwd host / bwd drain (ms):", its, flush=True)
    print("allocs in loop:", ms1["num_device_alloc"] - ms0["num_device_alloc"], "frees:", ms1["num_device_free"] - ms0["num_device_free"],
          "reserved GB %.2f" % (ms1["reserved_bytes.all.current"] / 2**30), "active GB %.2f" % (ms1["active_bytes.all.current"] / 2**30), flush=True)
    _lib.profile_enable(True); _lib.profile_read()
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    st = _lib.profile_read(); _lib.profile_enable(False)
    tot = sum(v[0] / v[1] for v in st.values())
    print(P, "wall ms %.3f" % (wall * 1e3), "stages ms %.3f" % tot, {k: round(v[0] / v[1], 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}, flush=True)
    print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], flush=True)
    del t
    torch.cuda.empty_cache()
