"""Host-side mirror of the reference's loss helpers (utils/loss_utils.py:17-63) on the MI355X C-ABI library, plus the
fused QAT loss of finetune.py:48.  (the step right after the rasterizer in the QAT loop.)

    l1_loss(x, y), ssim(img1, img2)                       same names / meaning as the reference
    l1_ssim_loss(image, gt, lambda_dssim=0.2)            == (1-l)*l1_loss(image, gt) + l*(1 - ssim(image, gt))

All three are differentiable w.r.t. their FIRST argument (the rendered image); the ground truth is treated as a constant,
which is how every call site of the reference uses them.  One fused forward kernel + one fused backward kernel instead of
five grouped convolutions and their autograd graph.  No CPU path."""
import ctypes as C

import torch

from . import _lib


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check(img, gt):
    if img.dim() != 3 or img.shape != gt.shape:
        raise RuntimeError("l1/ssim loss: expected two [C,H,W] tensors of the same shape")
    if not img.is_cuda or not gt.is_cuda:
        raise RuntimeError("c3dgs_amd: loss inputs must be GPU tensors (there is no CPU path)")
    return img.detach().contiguous().float(), gt.detach().contiguous().float()


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, l1_coeff, ssim_coeff, const):
        L = _lib.lib()
        x, y = _check(image, gt)
        Cc, H, W = x.shape
        need_bwd = bool(ctx.needs_input_grad[0])
        dmaps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=x.device) if need_bwd else None
        sums = torch.empty(128, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = L.c3dgs_l1_ssim_forward(Cc, H, W, x.data_ptr(), y.data_ptr(), dmaps.data_ptr() if need_bwd else None,
                                         sums.data_ptr(), _stream(x.device))
            _lib.check(rc)
            n = float(Cc * H * W)
            value = torch.empty((), dtype=torch.float32, device=x.device)
            _lib.check(L.c3dgs_l1_ssim_value(sums.data_ptr(), l1_coeff / n, ssim_coeff / n, float(const), value.data_ptr(),
                                             _stream(x.device)))
        ctx.coeffs = (float(l1_coeff), float(ssim_coeff))
        ctx.save_for_backward(x, y, dmaps if need_bwd else torch.empty(0))
        return value

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        x, y, dmaps = ctx.saved_tensors
        Cc, H, W = x.shape
        g = grad_out.detach().reshape(1).to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = L.c3dgs_l1_ssim_backward(Cc, H, W, x.data_ptr(), y.data_ptr(), dmaps.data_ptr(), g.data_ptr(),
                                          ctx.coeffs[0], ctx.coeffs[1], out.data_ptr(), _stream(x.device))
        _lib.check(rc)
        return out, None, None, None, None


def l1_ssim_loss(image, gt, lambda_dssim: float = 0.2):
    """finetune.py:48: (1 - lambda) * l1_loss + lambda * (1 - ssim)."""
    return _L1SSIM.apply(image, gt, 1.0 - lambda_dssim, -lambda_dssim, lambda_dssim)


def l1_loss(network_output, gt):
    """utils/loss_utils.py:17-18."""
    return _L1SSIM.apply(network_output, gt, 1.0, 0.0, 0.0)


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:33-43 (window_size 11, size_average=True: the only configuration the reference calls)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("c3dgs_amd.loss.ssim implements window_size=11, size_average=True")
    return _L1SSIM.apply(img1, img2, 0.0, 1.0, 0.0)
