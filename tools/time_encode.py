"""Row N4 timings on one MI355X: extract_rot_scale (HIP Jacobi) vs the reference's composition with torch.linalg.eigh on
the same device, the int8 payload launch of save_npz, and the Morton order. python tools/time_encode.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from c3dgs_amd import encode
from c3dgs_amd.model import GaussianModel
from tests import synth

dev = torch.device("cuda", 0)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


GS = 754_096
g = torch.Generator(device=dev).manual_seed(0)
s = torch.exp(torch.randn(GS, 3, device=dev, generator=g) * 0.8)
s = s / s.norm(dim=1, keepdim=True)
q = torch.nn.functional.normalize(torch.randn(GS, 4, device=dev, generator=g))
from c3dgs_amd.model import _covariance
cov6 = _covariance(s, 1.0, q).contiguous()
print(f"extract_rot_scale HIP, n={GS}: {timed(lambda: encode.extract_rot_scale(cov6)):.3f} ms")


def torch_ref():
    full = encode.to_full_cov(cov6)
    S, R = torch.linalg.eigh(full + torch.eye(3, device=dev) * 1e-8, UPLO="U")
    return S.sqrt().nan_to_num(nan=1e-6), R * R.det()[..., None, None]


try:
    print(f"torch.linalg.eigh + det on device (reference composition, without the quaternion step): {timed(torch_ref, 2):.1f} ms")
except Exception as e:
    print("torch.linalg.eigh on device failed:", repr(e))

sc = synth.scene(3_000_000)
raw = synth.raw_params(synth.index_scene(sc))
m = GaussianModel(3, device=dev).set_tensors(**raw)
_ = (m.get_opacity, m.get_scaling_normalized, m.get_scaling_factor, m._rotation_post_activation, m._get_features_raw)
print(f"save_npz int8 payload (6 tensors, one launch), 3M scene: {timed(m.quantized_payload):.3f} ms")
print(f"morton_order 3M: {timed(lambda: encode.morton_order(m._xyz.detach())):.3f} ms")
