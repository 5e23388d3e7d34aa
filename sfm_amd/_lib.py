"""ctypes binding of libsfm_amd.so (the C ABI declared in include/sfm_amd.h).

There is no fallback: if the shared library is missing or a call fails, the caller
gets an exception.  torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsfm_amd.so")
PARAM_STRIDE = 32

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float
_LL = ctypes.c_longlong

# name -> argtypes; must list every symbol of include/sfm_amd.h (checked by tests/test_abi.py)
SIGNATURES = {
    "sfm_pose_setup": [_P, _L, _L, _P, _I, _I, _I, _P, _P],
    "sfm_pose_grad": [_P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _F, _P, _L, _L, _P],
    "sfm_bwd_blocks": [_I, _I],
    "sfm_euler2mat": [_P, _P, _P, _I, _P, _P],
    "sfm_pose_vec2mat": [_P, _I, _P, _P],
    "sfm_meshgrid": [_I, _I, _I, _I, _P, _P],
    "sfm_pixel2cam": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "sfm_cam2pixel": [_P, _P, _I, _I, _I, _P, _P, _P],
    "sfm_euler2mat_bwd": [_P, _P, _P, _P, _I, _P, _P, _P, _P],
    "sfm_pose_vec2mat_bwd": [_P, _P, _I, _P, _P],
    "sfm_geom_bwd_workspace_floats": [_I, _I, _I],
    "sfm_pixel2cam_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "sfm_cam2pixel_bwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "sfm_project_coords": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "sfm_bilinear_sampler_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sfm_bilinear_sampler_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_inverse_warp_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "sfm_inverse_warp_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sfm_bilinear_project_fwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "sfm_bilinear_project_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "sfm_splat_warp_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "sfm_splat_warp_fwd_static_max_floats": [],
    "sfm_splat_warp_fwd_static": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "sfm_splat_warp_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sfm_ssim_fwd": [_P, _P, _I, _I, _I, _P, _P],
    "sfm_ssim_bwd": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "sfm_loss_workspace_floats": [_I, _I, _I, _I, _I],
    "sfm_loss_num_partials": [_I, _I, _I, _I, _I],
    "sfm_loss_workspace_init": [_P, _I, _I, _I, _I, _I, _P],
    "sfm_loss_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _LL, _I, _P, _P],
    "sfm_loss_forward_pyr": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _LL, _I, _P, _P],
    "sfm_loss_backward_pyr": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _LL, _F, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "sfm_loss_forward_warps": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "sfm_loss_forward_warps_pyr": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "sfm_loss_forward_gather": [_P, _P, _I, _I, _I, _I, _I, _LL, _I, _P],
    "sfm_loss_forward_splat": [_P, _P, _I, _I, _I, _I, _I, _LL, _I, _P, _P, _P],
    "sfm_loss_workspace_region": [_I, _I, _I, _I, _I, _I, _P, _P],
    "sfm_aspp_conv3x3_workspace_floats": [_I, _I, _I, _I, _I],
    "sfm_aspp_conv3x3_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv3x3_bwd_data": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv3x3_bwd_weight": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv3x3_fwd_ld": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv3x3_bwd_weight_ld": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_disp_normalize_workspace_floats": [_P, _I, _I],
    "sfm_disp_normalize_fwd": [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    "sfm_disp_normalize_bwd": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "sfm_pyramid_upsample_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "sfm_pyramid_upsample_bwd_workspace_floats": [_P, _P, _I, _I, _I, _I, _I],
    "sfm_pyramid_upsample_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_resize_bilinear_bwd_workspace_floats": [_I, _I, _I, _I, _I, _I],
    "sfm_resize_bilinear_fwd": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "sfm_resize_bilinear_bwd": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_smooth_workspace_floats": [_P, _P, _P, _I, _I],
    "sfm_smooth_loss": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    "sfm_scale_grads": [_P, _P, _I, _P, _P],
    "sfm_aspp_conv3x3_bn_act_infer": [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv1x1_fwd": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv1x1_bwd_data": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_aspp_conv1x1_bwd_weight": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "sfm_disp_head_workspace_floats": [_I, _I, _I, _I],
    "sfm_disp_head_fwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "sfm_disp_head_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sfm_bn_elu_workspace_floats": [_LL, _I],
    "sfm_bn_elu_fwd": [_P, _P, _P, _LL, _I, _F, ctypes.c_double, _I, _P, _P, _P, _P, _P, _P, _P],
    "sfm_bn_elu_bwd": [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _P, _P, _P, _P, _P],
    "sfm_bn_elu_fwd_ld": [_P, _P, _P, _LL, _I, _F, ctypes.c_double, _I, _P, _P, _P, _I, _P, _P, _P, _P],
    "sfm_bn_elu_infer": [_P, _P, _P, _P, _P, _LL, _I, _F, _I, _P, _I, _P],
    "sfm_aspp_conv_bn_act_infer_ld": [_P, _I, _P, _I, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "sfm_bn_elu_bwd_ld": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _LL, _I, _I, _P, _P, _P, _P, _P],
    "sfm_bn_local_stats": [_P, _LL, _I, _P, _P, _P],
    "sfm_bn_elu_apply_stats": [_P, _P, _P, _P, _I, _LL, _I, _F, ctypes.c_double, _I, _P, _P, _P, _P, _P, _P],
    "sfm_bn_elu_bwd_sums": [_P, _P, _P, _P, _P, _LL, _I, _I, _P, _P, _P, _P],
    "sfm_bn_elu_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _LL, _I, _I, _P, _P],
    "sfm_snippet_workspace_floats": [_I, _I, _I, _I],
    "sfm_snippet_unpack": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _I, _F, _P, _P, _P, _P, _P],
    "sfm_prof_enable": [_I],
    "sfm_prof_stages": [],
    "sfm_prof_read": [_P],
    "sfm_loss_backward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _LL, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P],
}

_lib: Optional[ctypes.CDLL] = None


class SfmAmdError(RuntimeError):
    """A C-ABI call returned non-zero (argument rejected or HIP launch failure)."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing - build it first (python -c 'import __graft_entry__ as g; g.build()' "
                f"or make -C sfm_amd/csrc).  sfm_amd has no CPU or PyTorch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the build is stale
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name.endswith(("_workspace_floats", "_max_floats")) else ctypes.c_int
        _lib = lib
    return _lib


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def call(name: str, *args) -> None:
    rc = getattr(load(), name)(*args)
    if rc != 0:
        kind = "argument rejected" if rc < 0 else "hipError_t"
        raise SfmAmdError(f"{name} failed: {kind} {rc}")
