"""The dilated 3x3 convolutions of the ASPP decoder (``nets.py:451-471`` ``disp_bottleneck_dense``,
the ``slim.conv2d(x, out_channels, [3, 3], rate=r)`` calls at :457, :458, :463) on fp32-input MFMA.

``dilated_conv3x3(x, w, rate)`` is the convolution itself - NHWC activations, HWIO weights, SAME
padding, stride 1, fp32 - forward and backward (data and weights) through the C ABI
(``sfm_aspp_conv3x3_*``, csrc/aspp.hip).  ``batch_norm_elu`` is the normaliser + activation the
arg-scope appends to it (``slim.batch_norm`` in training mode, then ``tf.nn.elu``; nets.py:653-656,
resnet/resnet_utils.py:232-251; csrc/bn_elu.hip), and ``conv2d_bn_elu`` chains the two: what one
``slim.conv2d(x, C, [3, 3], rate=r)`` of ``disp_bottleneck_dense`` computes.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call, ptr, stream_of
from .utils import _check, _same_device

Tensor = torch.Tensor
_workspaces = {}


def _workspace(B, H, W, Cin, Cout, device) -> Tensor:
    """One scratch buffer per layer shape and device (split-K slabs, flipped weights)."""
    key = (B, H, W, Cin, Cout, str(device))
    ws = _workspaces.get(key)
    if ws is None:
        n = _lib.load().sfm_aspp_conv3x3_workspace_floats(B, H, W, Cin, Cout)
        if n <= 0:
            raise ValueError(f"unsupported conv shape B={B} H={H} W={W} Cin={Cin} Cout={Cout} (channels must be multiples of 32)")
        ws = _workspaces[key] = torch.empty(n, dtype=torch.float32, device=device)
    return ws


class _DilatedConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, rate):
        B, H, W, Cin = x.shape
        Cout = w.shape[3]
        with torch.cuda.device(x.device):
            ws = _workspace(B, H, W, Cin, Cout, x.device)
            out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
            call("sfm_aspp_conv3x3_fwd", ptr(x), ptr(w), B, H, W, Cin, Cout, rate, ptr(out), ptr(ws), stream_of(x))
        ctx.save_for_backward(x, w)
        ctx.rate = rate
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w = ctx.saved_tensors
        B, H, W, Cin = x.shape
        Cout = w.shape[3]
        gout = gout.contiguous()
        gx = gw = None
        with torch.cuda.device(x.device):
            ws = _workspace(B, H, W, Cin, Cout, x.device)
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                call("sfm_aspp_conv3x3_bwd_data", ptr(gout), ptr(w), B, H, W, Cin, Cout, ctx.rate, ptr(gx), ptr(ws), stream_of(x))
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(w)
                call("sfm_aspp_conv3x3_bwd_weight", ptr(x), ptr(gout), B, H, W, Cin, Cout, ctx.rate, ptr(gw), ptr(ws),
                     stream_of(x))
        return gx, gw, None


def dilated_conv3x3(x: Tensor, w: Tensor, rate: int) -> Tensor:
    """``slim.conv2d(x, w.shape[3], [3, 3], rate=rate)`` without normaliser / activation / bias:
    x [B,H,W,Cin], w [3,3,Cin,Cout] (HWIO) -> [B,H,W,Cout].  Channels must be multiples of 32."""
    x = _check(x, "x", 4)
    w = _check(w, "w", 4)
    if w.shape[0] != 3 or w.shape[1] != 3 or w.shape[2] != x.shape[3]:
        raise ValueError(f"w must be [3, 3, {x.shape[3]}, Cout], got {tuple(w.shape)}")
    if int(rate) < 1:
        raise ValueError("rate must be >= 1")
    _same_device(x, w)
    return _DilatedConv3x3.apply(x, w, int(rate))


BN_DECAY = 0.9997    # args.batch_norm_decay, SfMLearner_uni.py:19
BN_EPSILON = 1e-5    # args.batch_norm_epsilon, SfMLearner_uni.py:20


class _BatchNormElu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, decay, epsilon):
        C = x.shape[-1]
        M = x.numel() // C
        lib = _lib.load()
        n_ws = lib.sfm_bn_elu_workspace_floats(M, C)
        if n_ws < 0:
            raise ValueError(f"unsupported batch-norm shape M={M} C={C} (C must be a multiple of 4 with C/4 dividing 256)")
        with torch.cuda.device(x.device):
            ws = torch.empty(n_ws // 2 + 1, dtype=torch.float64, device=x.device)  # fp64 partials: 8-byte aligned
            y = torch.empty_like(x)
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            call("sfm_bn_elu_fwd", ptr(x), ptr(gamma), ptr(beta), M, C, float(epsilon), float(decay), ptr(moving_mean),
                 ptr(moving_var), ptr(y), ptr(mean), ptr(invstd), ptr(ws), stream_of(x))
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.mark_non_differentiable(mean, invstd)
        return y, mean, invstd

    @staticmethod
    def backward(ctx, gy, _gm, _gi):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        gy = gy.contiguous()
        with torch.cuda.device(x.device):
            ws = torch.empty(_lib.load().sfm_bn_elu_workspace_floats(M, C) // 2 + 1, dtype=torch.float64, device=x.device)
            gx = torch.empty_like(x)
            ggamma = torch.empty_like(gamma)
            gbeta = torch.empty_like(gamma)
            call("sfm_bn_elu_bwd", ptr(x), ptr(y), ptr(gy), ptr(gamma), ptr(mean), ptr(invstd), M, C, ptr(gx), ptr(ggamma),
                 ptr(gbeta), ptr(ws), stream_of(x))
        return gx, ggamma, gbeta, None, None, None, None


def batch_norm_elu(x: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Tensor = None, moving_var: Tensor = None,
                   decay: float = BN_DECAY, epsilon: float = BN_EPSILON) -> Tensor:
    """``tf.nn.elu(slim.batch_norm(x, decay, epsilon, scale=True, fused=True, is_training=True,
    updates_collections=None))`` - the normaliser + activation of the ASPP arg-scope (nets.py:653-656).
    x [..., C] NHWC; gamma, beta [C]; moving_mean / moving_var [C] are updated in place when given."""
    x = _check(x, "x", x.dim() if isinstance(x, torch.Tensor) else 4)
    C = x.shape[-1]
    gamma = _check(gamma, "gamma", 1, C)
    beta = _check(beta, "beta", 1, C)
    for name, t in (("moving_mean", moving_mean), ("moving_var", moving_var)):
        if t is not None:
            _check(t, name, 1, C)
            if not t.is_contiguous():
                raise ValueError(f"{name} is updated in place: it must be contiguous")
    _same_device(x, gamma, beta, moving_mean, moving_var)
    return _BatchNormElu.apply(x, gamma, beta, moving_mean, moving_var, float(decay), float(epsilon))[0]


def conv2d_bn_elu(x: Tensor, w: Tensor, gamma: Tensor, beta: Tensor, rate: int, moving_mean: Tensor = None,
                  moving_var: Tensor = None, decay: float = BN_DECAY, epsilon: float = BN_EPSILON) -> Tensor:
    """One ``slim.conv2d(x, Cout, [3, 3], rate=rate)`` of ``disp_bottleneck_dense`` under its arg-scope
    (nets.py:457, 458, 463): dilated convolution (no bias: a normaliser is set), batch norm, ELU."""
    return batch_norm_elu(dilated_conv3x3(x, w, rate), gamma, beta, moving_mean, moving_var, decay, epsilon)
