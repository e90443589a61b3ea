"""The drivers either side of the hot path: the QAT fine-tuning loop and the compression run.

    finetune(...)            <- finetune.py:10-66        (hot loop C of SURVEY.md section 3)
    run_vq(...)              <- compress.py:202-290      (sensitivity -> prune + VQ -> fine-tune -> npz)
    OptimizationParams / CompressionParams               <- arguments/__init__.py:85-136 (defaults only, no argparse)

The reference's `Scene` (COLMAP / Blender loaders, camera JSON) is outside the path: here `scene` is anything with
`getTrainCameras()` -> sequence of cameras carrying `intrinsic`, `extrinsic_vector` and `original_image` (a plain list
of such cameras is accepted too), and `loaded_iter` (default 0). Everything numeric runs in the package's HIP kernels:
GaussianModel.render (fused getters + rasterizer), the fused L1+SSIM loss, the fused Adam.
"""
import gc
import json
import os
import time
from random import randint
from typing import Optional

import torch

from . import loss as _loss
from . import sensitivity as _sensitivity
from .vq import CompressionSettings, compress_gaussians


class OptimizationParams:
    """arguments/__init__.py:116-136."""

    def __init__(self, **overrides):
        self.iterations = 30_000
        self.position_lr_init = 0.00016
        self.position_lr_final = 0.0000016
        self.position_lr_delay_mult = 0.01
        self.position_lr_max_steps = 30_000
        self.feature_lr = 0.0025
        self.opacity_lr = 0.05
        self.scaling_lr = 0.005
        self.rotation_lr = 0.001
        self.percent_dense = 0.01
        self.lambda_dssim = 0.2
        self.densification_interval = 100
        self.opacity_reset_interval = 3000
        self.densify_from_iter = 500
        self.densify_until_iter = 15_000
        self.densify_grad_threshold = 0.0002
        self.random_background = False
        self.not_quantization_aware = False
        _apply(self, overrides)


class CompressionParams:
    """arguments/__init__.py:85-113."""

    def __init__(self, **overrides):
        self.load_iteration = -1
        self.finetune_iterations = 5000
        self.color_codebook_size = 2 ** 12
        self.color_importance_include = 0.6 * 1e-6
        self.color_importance_prune = 0.0
        self.color_cluster_iterations = 100
        self.color_decay = 0.8
        self.color_batch_size = 2 ** 18
        self.color_weights_per_param = False
        self.color_compress_non_dir = True
        self.not_compress_color = False
        self.gaussian_codebook_size = 2 ** 12
        self.gaussian_importance_include = 0.3 * 1e-5
        self.gaussian_cluster_iterations = 800
        self.gaussian_decay = 0.8
        self.gaussian_batch_size = 2 ** 20
        self.not_compress_gaussians = False
        self.not_sort_morton = False
        self.prune_threshold = 0.
        self.output_vq = "./eval_vq"
        self.start_checkpoint = ""
        _apply(self, overrides)


def _apply(obj, overrides):
    for k, v in overrides.items():
        if not hasattr(obj, k):
            raise TypeError(f"{type(obj).__name__} has no parameter {k!r}")
        setattr(obj, k, v)


def _train_cameras(scene):
    return list(scene.getTrainCameras()) if hasattr(scene, "getTrainCameras") else list(scene)


def finetune(scene, dataset, opt, comp, pipe, debug_from=-1, log=None):
    """finetune.py:10-66. One camera per iteration, drawn without replacement from a stack that is refilled when empty
    (`pop(randint(0, len - 1))` on Python's `random`, like the reference); render -> (1 - l) L1 + l (1 - SSIM) ->
    backward -> learning-rate update -> Adam step (not after the last iteration).

    The reference reads `loss.item()` every iteration for its progress bar, which drains the GPU queue each time; here
    the losses stay on the device and the same exponential moving average (0.4 / 0.6) is evaluated every 10 iterations.
    Returns that average after the last iteration. `dataset` needs `white_background` only; `log(iteration, ema)` is
    called where the reference updates its progress bar."""
    gaussians = scene.gaussians if hasattr(scene, "gaussians") else dataset.gaussians
    first_iter = int(getattr(scene, "loaded_iter", 0) or 0)
    max_iter = first_iter + comp.finetune_iterations
    bg_color = [1, 1, 1] if getattr(dataset, "white_background", False) else [0, 0, 0]
    background = torch.tensor(bg_color, dtype=torch.float32, device=gaussians.device)

    gaussians.training_setup(opt)
    gaussians.update_learning_rate(first_iter)

    viewpoint_stack = None
    ema_loss_for_log = 0.0
    pending = []
    first_iter += 1
    for iteration in range(first_iter, max_iter + 1):
        if not viewpoint_stack:
            viewpoint_stack = _train_cameras(scene).copy()
        viewpoint_cam = viewpoint_stack.pop(randint(0, len(viewpoint_stack) - 1))
        if (iteration - 1) == debug_from:
            pipe.debug = True
        render_pkg = gaussians.render(viewpoint_cam, pipe, background)
        image = render_pkg["render"]
        gt_image = viewpoint_cam.original_image.to(image.device)
        loss = _loss.l1_ssim_loss(image, gt_image, opt.lambda_dssim)
        loss.backward()
        gaussians.update_learning_rate(iteration)
        pending.append(loss.detach())
        if iteration % 10 == 0 or iteration == max_iter:
            for v in torch.stack(pending).tolist():                 # one host read per 10 iterations
                ema_loss_for_log = 0.4 * v + 0.6 * ema_loss_for_log
            pending = []
            if log is not None:
                log(iteration, ema_loss_for_log)
        if iteration < max_iter:
            gaussians.optimizer.step()
            gaussians.optimizer.zero_grad(set_to_none=True)
    return ema_loss_for_log


def run_vq(gaussians, scene, optim_params, pipeline_params, comp_params, dataset=None, group=None, silent=True,
           out_file: Optional[str] = None):
    """compress.py:202-290 on an already constructed model and camera set: sensitivity (use_gt) -> prune + colour /
    covariance VQ -> QAT fine-tuning -> Morton-sorted npz. Returns (timings dict, npz path). The reference's
    evaluation pass (`render_and_eval`: PSNR / SSIM / LPIPS over the test set) is not part of the path.
    `group`: process group for the camera-sharded sensitivity pass and the sharded Lloyd steps (None = one GPU)."""
    timings = {}
    dev = gaussians.device

    def clock():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.time()

    cameras = _train_cameras(scene)
    t0 = clock()
    color_importance, gaussian_sensitivity = _sensitivity.calc_importance_experimental(
        gaussians, cameras, pipeline_params, use_gt=True, group=group)
    timings["sensitivity_calculation"] = clock() - t0

    with torch.no_grad():
        t0 = clock()
        color_importance_n = color_importance.amax(-1)
        gaussian_importance_n = gaussian_sensitivity.amax(-1)
        del color_importance, gaussian_sensitivity
        color_settings = CompressionSettings(
            codebook_size=comp_params.color_codebook_size, importance_prune=comp_params.color_importance_prune,
            importance_include=None, importance_include_relative=0.9, steps=int(comp_params.color_cluster_iterations),
            decay=comp_params.color_decay, batch_size=comp_params.color_batch_size)
        gaussian_settings = CompressionSettings(
            codebook_size=comp_params.gaussian_codebook_size, importance_prune=None, importance_include=None,
            importance_include_relative=0.75, steps=int(comp_params.gaussian_cluster_iterations),
            decay=comp_params.gaussian_decay, batch_size=comp_params.gaussian_batch_size)
        compress_gaussians(gaussians, color_importance_n, gaussian_importance_n,
                           color_settings if not comp_params.not_compress_color else None,
                           gaussian_settings if not comp_params.not_compress_gaussians else None,
                           comp_params.color_compress_non_dir, prune_threshold=comp_params.prune_threshold,
                           silent=silent, group=group)
        timings["clustering"] = clock() - t0
    gc.collect()

    os.makedirs(comp_params.output_vq, exist_ok=True)
    with open(os.path.join(comp_params.output_vq, "cfg_args_comp"), "w") as f:
        f.write(str(vars(comp_params)))

    iteration = int(getattr(scene, "loaded_iter", 0) or 0) + comp_params.finetune_iterations
    if comp_params.finetune_iterations > 0:
        t0 = clock()
        holder = _SceneView(gaussians, cameras, getattr(scene, "loaded_iter", 0))
        finetune(holder, dataset if dataset is not None else _Dataset(), optim_params, comp_params, pipeline_params,
                 debug_from=-1)
        timings["finetune"] = clock() - t0

    if out_file is None:
        out_file = os.path.join(comp_params.output_vq, f"point_cloud/iteration_{iteration}/point_cloud.npz")
    os.makedirs(os.path.dirname(out_file) or ".", exist_ok=True)
    t0 = clock()
    gaussians.save_npz(out_file, sort_morton=not comp_params.not_sort_morton)
    timings["encode"] = clock() - t0
    timings["total"] = sum(timings.values())
    with open(os.path.join(comp_params.output_vq, "times.json"), "w") as f:
        json.dump(timings, f)
    return timings, out_file


class _Dataset:
    white_background = False


class _SceneView:
    def __init__(self, gaussians, cameras, loaded_iter):
        self.gaussians, self._cameras, self.loaded_iter = gaussians, cameras, int(loaded_iter or 0)

    def getTrainCameras(self):
        return self._cameras
