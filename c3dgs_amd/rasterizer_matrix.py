"""The matrix-`extrinsic` API of the reference's two sibling rasterizer packages
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py and the byte-identical
diff-gaussian-rasterization-camera copy): settings WITHOUT a pose (:503-513), modules whose `forward` /
`markVisible` take the 4x4 world->camera matrix as `extrinsic=` (:516-659), function wrappers (:41-96).

Same kernels and autograd functions as c3dgs_amd.rasterizer; only the camera set-up differs (`extrinsic`,
`extrinsic @ getProjectionMatrix(intrinsic)`, `extrinsic.inverse()[3, :3]`, :129-135). The reference returns a
`grad_matrix` for `extrinsic` that is a device copy of the view matrix, not a gradient (rasterizer_impl.cu:434,694-696;
SURVEY App. C): not reproduced -- the pose receives no gradient here."""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import rasterizer as _r
from .rasterizer import (_C, _check_exclusive, _empty, _RasterizeGaussians, _RasterizeGaussiansIndexed,  # noqa: F401
                         cpu_deep_copy_tuple, getProjectionMatrix, mat_to_quat, quat_to_mat)


def _pose(extrinsic):
    if extrinsic is None or extrinsic.dim() != 2 or tuple(extrinsic.shape) != (4, 4):
        raise RuntimeError("extrinsic must be a 4x4 matrix (world->camera, transposed as quat_to_mat returns it)")
    return extrinsic


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        extrinsic):
    """reference :41-63."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, _pose(extrinsic))


def rasterize_gaussians_indexed(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales, scale_factors,
                                rotations, cov3Ds_precomp, raster_settings, extrinsic):
    """reference :66-96."""
    return _RasterizeGaussiansIndexed.apply(means3D, means2D, sh, sh_indices, g_indices, colors_precomp, opacities, scales,
                                            scale_factors, rotations, cov3Ds_precomp, raster_settings, _pose(extrinsic))


class GaussianRasterizationSettings(NamedTuple):
    """reference :503-513 (no pose field)."""
    intrinsic: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    sh_degree: int
    prefiltered: bool
    debug: bool
    clamp_color: bool


class GaussianRasterizer(nn.Module):
    """reference :516-584."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions, extrinsic):
        with torch.no_grad():
            view, proj = _r.camera_matrices(self.raster_settings.intrinsic, _pose(extrinsic), positions.device)[:2]
            return _C.mark_visible(positions, view, proj)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extrinsic=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs = _empty() if shs is None else shs
        colors_precomp = _empty() if colors_precomp is None else colors_precomp
        scales = _empty() if scales is None else scales
        rotations = _empty() if rotations is None else rotations
        cov3D_precomp = _empty() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, extrinsic)


class GaussianRasterizerIndexed(nn.Module):
    """reference :587-659."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions, extrinsic):
        with torch.no_grad():
            view, proj = _r.camera_matrices(self.raster_settings.intrinsic, _pose(extrinsic), positions.device)[:2]
            return _C.mark_visible(positions, view, proj)

    def forward(self, means3D, means2D, opacities, sh_indices, g_indices, shs=None, colors_precomp=None, scales=None,
                scale_factors=None, rotations=None, cov3D_precomp=None, extrinsic=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs = _empty() if shs is None else shs
        colors_precomp = _empty() if colors_precomp is None else colors_precomp
        scales = _empty() if scales is None else scales
        scale_factors = _empty() if scale_factors is None else scale_factors
        rotations = _empty() if rotations is None else rotations
        cov3D_precomp = _empty() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians_indexed(means3D, means2D, shs, sh_indices, g_indices, colors_precomp, opacities, scales,
                                           scale_factors, rotations, cov3D_precomp, self.raster_settings, extrinsic)
