"""Morton ordering for the compressed on-disk layout.

    morton_codes(xyz) -> int64[P]     the value `mortonEncode(xyz_q, pp_diap.argsort())` of GaussianModel._sort_morton
    morton_order(xyz) -> int64[P]     `...sort().indices` (stable), usable to permute every per-Gaussian tensor

Reference: scene/gaussian_model.py:997-1003 and :1417-1432.  No CPU path."""
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
