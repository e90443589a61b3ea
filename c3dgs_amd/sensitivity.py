"""Sensitivity pass of the compression pipeline, camera-sharded over GPUs.

Reference: compress.py:81-119 `calc_importance_experimental` -- for every evaluation camera render the scene with the
NON-indexed rasterizer (precomputed 3D covariance = unit-scale covariance x scaling_factor^2, clamp_color=False, black
background), back-propagate either image.sum() or the L1+SSIM loss against the ground truth, and accumulate
|d loss / d SH| and |d loss / d cov3d|; finally divide by the number of pixels seen.

The sum over cameras is the only coupling between views, so ranks take cameras r, r+G, r+2G, ... and the three
accumulators (plus the pixel count) are all-reduced ONCE at the end (RCCL over xGMI with backend "nccl"): no collective
inside the per-view data path.  Everything numeric runs in the package's HIP kernels (rasterizer fwd/bwd, fused loss).
"""
from typing import Callable, Iterable, Optional, Tuple

import torch

from . import loss as _loss
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _abs_accumulate(acc: torch.Tensor, grad: torch.Tensor) -> None:
    """acc += |grad| (compress.py:110-113). On the GPU one launch of the library (one pass over grad, one read-modify-write of
    acc); tensors that are not on a GPU (the gloo tests of the sharding logic inject their own compute) take the torch form."""
    if acc.is_cuda and grad.is_cuda and acc.is_contiguous() and acc.dtype == torch.float32 and grad.dtype == torch.float32:
        from . import _lib
        g = grad.contiguous()
        with torch.cuda.device(acc.device):
            _lib.check(_lib.lib().c3dgs_abs_accumulate(acc.numel(), g.data_ptr(), acc.data_ptr(),
                                                      torch.cuda.current_stream(acc.device).cuda_stream))
    else:
        acc += torch.abs(grad)


def _dist(group):
    import torch.distributed as dist
    if group is None or not dist.is_available() or not dist.is_initialized():
        return None, 0, 1, None
    pg = None if group is True else group
    return dist, dist.get_rank(pg), dist.get_world_size(pg), pg


def make_render_fn(xyz: torch.Tensor, opacity: torch.Tensor, features_dc: torch.Tensor, features_rest: torch.Tensor,
                   cov3d_unit: torch.Tensor, scaling_factor: torch.Tensor, sh_degree: int = 3) -> Callable:
    """The render call of compress.py:101 on raw tensors: GaussianRasterizer with cov3D_precomp = cov3d * coeff."""
    coeff = scaling_factor.detach().square()
    bg = torch.zeros(3, dtype=torch.float32, device=xyz.device)

    def render(camera):
        rs = GaussianRasterizationSettings(intrinsic=camera.intrinsic, extrinsic_vector=camera.extrinsic_vector, bg=bg,
                                           scale_modifier=1.0, sh_degree=sh_degree, prefiltered=False, debug=False,
                                           clamp_color=False)
        rast = GaussianRasterizer(rs)
        shs = torch.cat([features_dc, features_rest], 1)
        means2D = torch.zeros_like(xyz, requires_grad=True)
        color, _ = rast(means3D=xyz, means2D=means2D, opacities=opacity, shs=shs, cov3D_precomp=cov3d_unit * coeff,
                        extrinsic_vector=camera.extrinsic_vector)
        return color
    return render


def calc_importance(render_fn: Callable, features_dc: torch.Tensor, features_rest: torch.Tensor, cov3d: torch.Tensor,
                    cameras: Iterable, use_gt: bool = False, lambda_dssim: float = 0.2, group=None,
                    loss_fn: Optional[Callable] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (importance [P, M*3], cov_grad [P, 6]), both divided by the total pixel count (compress.py:115-119).
    features_dc / features_rest / cov3d must be leaf tensors with requires_grad=True that render_fn depends on.
    `group`: None = single process; True / ProcessGroup = cameras sharded round-robin over the ranks."""
    dist, rank, world, pg = _dist(group)
    if loss_fn is None:
        loss_fn = lambda image, gt: _loss.l1_ssim_loss(image, gt, lambda_dssim)   # noqa: E731  (finetune.py:48 form)
    accum1 = torch.zeros_like(features_dc)
    accum2 = torch.zeros_like(features_rest)
    accum3 = torch.zeros_like(cov3d)
    num_pixels = 0
    for k, camera in enumerate(cameras):
        if k % world != rank:
            continue
        for t in (features_dc, features_rest, cov3d):
            t.grad = None
        image = render_fn(camera)
        if not use_gt:
            image.sum().backward()                                   # compress.py:103-104
        else:
            gt_image = camera.original_image.to(image.device)
            loss_fn(image, gt_image).backward()                      # compress.py:105-109
        _abs_accumulate(accum1, features_dc.grad)
        _abs_accumulate(accum2, features_rest.grad)
        _abs_accumulate(accum3, cov3d.grad)
        num_pixels += image.shape[1] * image.shape[2]
    if world > 1:
        npx = torch.tensor([float(num_pixels)], dtype=torch.float64, device=accum1.device)
        for t in (accum1, accum2, accum3, npx):
            dist.all_reduce(t, group=pg)
        num_pixels = int(npx.item())
    importance = torch.cat([accum1, accum2], 1).flatten(-2)
    return importance / num_pixels, accum3 / num_pixels


def calc_importance_experimental(gaussians, cameras: Iterable, pipeline_params, silent: bool = True, use_gt: bool = False,
                                 group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """compress.py:81-119 with the reference's signature (`scene` replaced by the iterable of cameras): the model's
    own render() is used, with cov3d = unit-scale covariance x scaling_factor^2 and clamp_color=False."""
    cov3d_scaled = gaussians.get_covariance().detach()
    coeff = gaussians.get_scaling_factor.detach().square()
    cov3d = (cov3d_scaled / coeff).requires_grad_(True)
    background = torch.zeros(3, dtype=torch.float32, device=cov3d.device)

    def render_fn(camera):
        # gather_visible=False: every row goes to the rasterizer, which culls what lies outside the frustum itself -- the
        # reference's `t[visible]` copies (gaussian_model.py:851-862) and their scatter in the backward are ~4 GB of traffic
        # per camera at 6M Gaussians for identical gradients (rows outside the frustum get zeros either way)
        return gaussians.render(camera, pipeline_params, background, clamp_color=False, cov3d=cov3d * coeff,
                                gather_visible=False)["render"]

    return calc_importance(render_fn, gaussians._features_dc, gaussians._features_rest, cov3d, cameras, use_gt=use_gt,
                           group=group)
