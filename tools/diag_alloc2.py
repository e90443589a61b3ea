import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench, c3dgs_amd
from c3dgs_amd import rasterizer as rz
dev = torch.device("cuda",0)
P,W,H=3_000_000,1920,1080
intr, ev, t, dL, ix = bench.build_workload(P,W,H,1200.0,dev)
rs = c3dgs_amd.GaussianRasterizationSettings(intrinsic=intr, extrinsic_vector=ev.to(dev), bg=torch.zeros(3, device=dev), scale_modifier=1.0, sh_degree=3, prefiltered=False, debug=False, clamp_color=True)
evd = ev.to(dev)
view, proj, campos, tfx, tfy, _, _ = rz.camera_matrices(intr, evd, dev)
E = torch.Tensor([])
log = []
orig = rz._Scratch.callback
def patched(self, name):
    def _resize(_user, nbytes):
        t0 = time.perf_counter()
        tt = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
        log.append((name, int(nbytes), (time.perf_counter() - t0) * 1e3))
        self.bufs[name] = tt
        return tt.data_ptr()
    cb = rz.RESIZE_FN(_resize)
    self._cbs[name] = cb
    return cb
rz._Scratch.callback = patched
def fwd():
    return rz._C.rasterize_gaussians_indexed(rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, H, W, t["shs"], 3, campos, t["sh_indices"], t["g_indices"], False, False, True)
def bwd(o):
    return rz._C.rasterize_gaussians_backward_indexed(rs.bg, t["means3D"], o[2], E, t["scales"], t["scale_factors"], t["rotations"], 1.0, E, view, proj, tfx, tfy, dL, t["shs"], 3, campos, o[3], o[0], o[4], o[5], False, t["sh_indices"], t["g_indices"])
for it in range(6):
    log.clear()
    t0 = time.perf_counter(); o = fwd(); t1 = time.perf_counter(); g = bwd(o); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print(f"it{it} fwd_host {1e3*(t1-t0):.2f} bwd_host {1e3*(t2-t1):.2f} tail_sync {1e3*(t3-t2):.2f} dev_allocs {ms['num_device_alloc']} dev_frees {ms['num_device_free']} retries {ms['num_alloc_retries']} reserved {ms['reserved_bytes.all.current']>>20}MiB", [(n, b>>20, round(m,2)) for n,b,m in log])
