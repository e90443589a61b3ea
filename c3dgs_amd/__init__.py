"""c3dgs_amd -- MI355X (gfx950) rasterizer + VQ hot path of Compressed 3D Gaussian Splatting.

Drop-in for the reference's two native extensions and the Python directly around them:

    c3dgs_amd.rasterizer  <->  diff_gaussian_rasterization_no_camera (quaternion `extrinsic_vector` API)
    c3dgs_amd.rasterizer_matrix <-> diff_gaussian_rasterization / diff_gaussian_rasterization_camera (4x4 `extrinsic` API)
    c3dgs_amd.vq          <->  weighted_distance._C.weightedDistance + compression/vq.py
    c3dgs_amd.loss        <->  utils/loss_utils.py (l1_loss, ssim) + the fused QAT loss of finetune.py:48
    c3dgs_amd.sensitivity <->  compress.py:calc_importance_experimental (camera-sharded)
    c3dgs_amd.encode      <->  GaussianModel._sort_morton / mortonEncode
    c3dgs_amd.model       <->  GaussianModel getters + FakeQuantize modules + render() glue (scene/gaussian_model.py)
    c3dgs_amd.optim       <->  the torch.optim.Adam step of the QAT loop (finetune.py:65-66), one fused launch

The numeric work runs in c3dgs_amd/libc3dgs_hip.so (include/c3dgs_hip.h); build it with
`python -m c3dgs_amd.build`.  There is no CPU fallback.
"""
import sys
import types

from . import encode, loss, model, optim, rasterizer, rasterizer_matrix, sensitivity, vq  # noqa: F401
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerIndexed,  # noqa: F401
                         getProjectionMatrix, mat_to_quat, quat_to_mat, rasterize_gaussians,
                         rasterize_gaussians_indexed, rasterize_gaussians_indexed_camera)
from .vq import (CompressionSettings, VectorQuantize, compress_color, compress_covariance, compress_gaussians,  # noqa: F401
                 join_features, vq_features, weightedDistance)

from .loss import l1_loss, l1_ssim_loss, ssim  # noqa: F401,E402
from .encode import morton_codes, morton_order  # noqa: F401,E402

__version__ = "0.1.0"


def install_as_reference_modules():
    """Register this package under the module names the reference imports
    (scene/gaussian_model.py:43-44, compression/vq.py:12), so the reference's Python runs unmodified."""
    sys.modules["diff_gaussian_rasterization_no_camera"] = rasterizer
    for name in ("diff_gaussian_rasterization", "diff_gaussian_rasterization_camera"):   # the matrix-`extrinsic` siblings
        sys.modules[name] = rasterizer_matrix
    wd = types.ModuleType("weighted_distance")
    wdc = types.ModuleType("weighted_distance._C")
    wdc.weightedDistance = lambda coefs, codebook: weightedDistance(coefs, codebook)
    wd._C = wdc
    sys.modules["weighted_distance"] = wd
    sys.modules["weighted_distance._C"] = wdc
