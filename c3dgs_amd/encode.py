"""Morton ordering for the compressed on-disk layout.

    morton_codes(xyz) -> int64[P]     the value `mortonEncode(xyz_q, pp_diap.argsort())` of GaussianModel._sort_morton
    morton_order(xyz) -> int64[P]     `...sort().indices` (stable), usable to permute every per-Gaussian tensor

    to_full_cov(cov6) -> [n,3,3]      utils/splats.py:7-24
    extract_rot_scale(cov) -> (rot [n,4], scaling [n,3])   utils/splats.py:27-35 (accepts [n,3,3] or the stripped [n,6])

Reference: scene/gaussian_model.py:997-1003 and :1417-1432; utils/splats.py.  No CPU path."""
import ctypes as C

import torch

from . import _lib


def _run(xyz):
    if xyz.dim() != 2 or xyz.size(1) != 3:
        raise RuntimeError("morton_order: xyz must have dimensions (num_points, 3)")
    if not xyz.is_cuda:
        raise RuntimeError("c3dgs_amd: morton_order needs a GPU tensor (there is no CPU path)")
    L = _lib.lib()
    x = xyz.detach().contiguous().float()
    P = int(x.size(0))
    codes = torch.empty(P, dtype=torch.int64, device=x.device)
    order = torch.empty(P, dtype=torch.int64, device=x.device)
    if P:
        ws = torch.empty(int(L.c3dgs_morton_workspace_bytes(P)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.c3dgs_morton_order(P, x.data_ptr(), codes.data_ptr(), order.data_ptr(), ws.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(rc)
    return codes, order


def morton_codes(xyz):
    return _run(xyz)[0]


def morton_order(xyz):
    return _run(xyz)[1]


def to_full_cov(cov):
    """utils/splats.py:7-24: stripped upper triangle [n,6] -> symmetric [n,3,3]."""
    idx = torch.tensor([0, 1, 2, 1, 3, 4, 2, 4, 5], device=cov.device)
    return cov[:, idx].reshape(-1, 3, 3)


def extract_rot_scale(cov):
    """utils/splats.py:27-35 for a batch of symmetric 3x3 covariances (or their stripped [n,6] form):
    -> (rot [n,4] unit quaternions, scaling [n,3] = sqrt of the ascending eigenvalues)."""
    if not cov.is_cuda:
        raise RuntimeError("c3dgs_amd: extract_rot_scale needs a GPU tensor (there is no CPU path)")
    if cov.dim() == 3:                       # UPLO="U": only the upper triangle is read
        cov = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)
    if cov.dim() != 2 or cov.size(1) != 6:
        raise RuntimeError("extract_rot_scale: cov must have dimensions (n, 3, 3) or (n, 6)")
    c6 = cov.detach().contiguous().float()
    n = int(c6.size(0))
    rot = torch.empty(n, 4, dtype=torch.float32, device=c6.device)
    scale = torch.empty(n, 3, dtype=torch.float32, device=c6.device)
    if n:
        _lib.check(_lib.lib().c3dgs_extract_rot_scale(n, c6.data_ptr(), rot.data_ptr(), scale.data_ptr(),
                                                      C.c_void_p(torch.cuda.current_stream(c6.device).cuda_stream)))
    return rot, scale
