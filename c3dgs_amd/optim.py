"""Fused Adam for the QAT inner loop (finetune.py:65-66; the optimizer the reference builds at
scene/gaussian_model.py:296-308: torch.optim.Adam(param_groups, lr=0.0, eps=1e-15)).

`Adam` is a torch.optim.Optimizer with torch.optim.Adam's constructor arguments and state keys ("step", "exp_avg",
"exp_avg_sq": state dicts are interchangeable), whose step() updates every parameter of every group with ONE kernel
launch per 16 tensors (csrc/adam.hip) instead of torch's dozen multi-tensor passes. weight_decay, amsgrad and maximize
are not used by the reference and are rejected. No CPU path."""
import ctypes as C
import math

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise RuntimeError("c3dgs_amd.optim.Adam mirrors the reference's plain Adam (no weight_decay / amsgrad / maximize)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        batch, keep = [], []
        key = None                                   # (betas, eps, device) shared by the tensors of one launch

        def flush():
            if not batch:
                return
            (beta1, beta2), eps, dev = key
            arr = (_lib.AdamTensor * len(batch))(*batch)
            with torch.cuda.device(dev):
                _lib.check(L.c3dgs_adam_step(len(batch), arr, beta1, beta2, eps,
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            batch.clear()
            keep.clear()

        for group in self.param_groups:              # the learning rate travels per tensor: groups share a launch
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("c3dgs_amd.optim.Adam: parameters must be contiguous float32 GPU tensors (there is no CPU path)")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                k = ((float(beta1), float(beta2)), eps, p.device)
                if key is not None and k != key:
                    flush()
                key = k
                keep.append(g)
                a = _lib.AdamTensor()
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                a.n = p.numel()
                a.step_size = lr / (1.0 - beta1 ** t)                      # double arithmetic, as torch's Python side
                a.bias_correction2_sqrt = math.sqrt(1.0 - beta2 ** t)
                batch.append(a)
                if len(batch) == 16:
                    flush()
        flush()
        return loss
