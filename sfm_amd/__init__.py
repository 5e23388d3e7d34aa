"""sfm_amd - MI355X (gfx950) implementation of the view-synthesis warp + photometric
loss hot path of alncat/SFM, behind the reference's own function signatures.

    from sfm_amd.utils import projective_inverse_warp, projective_warp, \
        bilinear_sampler, bilinear_project, tf_ssim
    from sfm_amd.loss import multi_scale_loss
    from sfm_amd.pyramid import normalize_disp, resize_bilinear, compute_smooth_loss
    from sfm_amd.aspp import dilated_conv3x3

The compute lives in ``libsfm_amd.so`` (hand-written HIP, C ABI in include/sfm_amd.h);
this package only validates arguments, allocates outputs with torch and binds the
calls.  There is no CPU or PyTorch fallback.
"""
from . import _lib  # noqa: F401
from .utils import (bilinear_project, bilinear_sampler, cam2pixel, corner_indices, euler2mat, meshgrid,  # noqa: F401
                    pixel2cam, pose_vec2mat, project_coords, projective_inverse_warp, projective_warp, ssim_value,
                    tf_ssim)
from .loss import GraphedTotalLoss, multi_scale_loss  # noqa: F401
from .pyramid import compute_smooth_loss, normalize_disp, resize_bilinear  # noqa: F401
from .aspp import (GradBuckets, allreduce_weight_grads, batch_norm_elu, conv1x1, conv2d_bn_elu, dilated_conv3x3,  # noqa: F401
                   disp_aspp_u_dense_decoder, disp_bottleneck_dense, disp_head)
from .data_loader import AugmentDraws, DataLoader, snippet_unpack  # noqa: F401
from . import synth  # noqa: F401
from ._lib import SfmAmdError  # noqa: F401

__all__ = ["projective_inverse_warp", "projective_warp", "bilinear_sampler", "bilinear_project", "tf_ssim",
           "ssim_value", "pose_vec2mat", "euler2mat", "meshgrid", "pixel2cam", "cam2pixel", "project_coords",
           "corner_indices", "multi_scale_loss", "GraphedTotalLoss", "compute_smooth_loss", "resize_bilinear", "normalize_disp",
           "dilated_conv3x3", "batch_norm_elu", "conv2d_bn_elu", "conv1x1", "disp_bottleneck_dense", "disp_head", "disp_aspp_u_dense_decoder", "allreduce_weight_grads", "GradBuckets",
           "DataLoader", "AugmentDraws", "snippet_unpack",
           "SfmAmdError"]
