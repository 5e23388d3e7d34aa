"""The dilated 3x3 convolutions of the ASPP decoder (``nets.py:451-471`` ``disp_bottleneck_dense``,
the ``slim.conv2d(x, out_channels, [3, 3], rate=r)`` calls at :457, :458, :463) on fp32-input MFMA.

``dilated_conv3x3(x, w, rate)`` is the convolution itself - NHWC activations, HWIO weights, SAME
padding, stride 1, fp32 - forward and backward (data and weights) through the C ABI
(``sfm_aspp_conv3x3_*``, csrc/aspp.hip).  The normaliser / activation slim applies after it
(batch norm + ELU from the arg-scope, nets.py:653-656) are outside this operator.
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
