"""The steps of ``build_train_graph`` either side of the warp loop that work on the prediction
pyramid, as hand-written kernels (csrc/pyramid.hip):

    normalize_disp        pred_disp = d / reduce_mean(d, [1,2,3]); pred_depth = 1/pred_disp   SfMLearner_uni.py:40,47
    resize_bilinear       tf.image.resize_bilinear(x, [H, W]), TF-1 defaults                  SfMLearner_uni.py:91,113
    compute_smooth_loss   SfMLearner.compute_smooth_loss                                      SfMLearner_uni.py:205-219
    smooth_regulariser    the whole smooth_loss accumulation of the scale loop                SfMLearner_uni.py:94-101

Each is differentiable (torch.autograd) and, like the rest of the package, has no CPU path.
The ``pyramid_*`` helpers are the batched forms ``sfm_amd.loss.multi_scale_loss`` uses: all
scales in one launch, outputs in the stacked layout the fused loss reads.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream_of
from .utils import _check

Tensor = torch.Tensor
MAX_MAPS = 16
SUPPORTED_C = (1, 2, 3, 4, 6, 9, 12)


def _ptrs(ts: Sequence[Optional[Tensor]]):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _ints(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def _floats(v: Sequence[float]):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


def _check_maps(maps: Sequence[Tensor], name: str, last: Optional[int] = None) -> List[Tensor]:
    if not 1 <= len(maps) <= MAX_MAPS:
        raise ValueError(f"{name}: between 1 and {MAX_MAPS} scales, got {len(maps)}")
    out = [_check(m, f"{name}[{s}]", 4, last) for s, m in enumerate(maps)]
    for m in out:
        if m.shape[0] != out[0].shape[0] or m.shape[3] != out[0].shape[3] or m.device != out[0].device:
            raise ValueError(f"{name}: every scale must share batch size, channel count and device")
    return out


# ----------------------------------------------------------------------------- upsample
def pyramid_upsample(maps: Sequence[Tensor], out_h: int, out_w: int) -> Tensor:
    """``[resize_bilinear(m, [H, W]) for m in maps]`` stacked: [S,B,H,W,C] (not differentiable here;
    ``pyramid_upsample_grad`` is its adjoint)."""
    maps = _check_maps(maps, "maps")
    bsz, _, _, ch = maps[0].shape
    if ch not in SUPPORTED_C:
        raise ValueError(f"channel count {ch} is not one of {SUPPORTED_C}")
    if any(m.shape[1] > out_h or m.shape[2] > out_w for m in maps):
        raise ValueError("the pyramid is only ever upsampled (SfMLearner_uni.py:91,113): every scale must be <= [H, W]")
    out = torch.empty(len(maps), bsz, out_h, out_w, ch, dtype=torch.float32, device=maps[0].device)
    with torch.cuda.device(out.device):
        call("sfm_pyramid_upsample_fwd", _ptrs(maps), _ints([m.shape[1] for m in maps]),
             _ints([m.shape[2] for m in maps]), len(maps), bsz, ch, out_h, out_w, ptr(out), stream_of(out))
    return out


def pyramid_upsample_grad(gout: Tensor, sizes: Sequence[Tuple[int, int]],
                          accumulate_into: Optional[Sequence[Tensor]] = None) -> List[Tensor]:
    """Adjoint of ``pyramid_upsample``: gout [S,B,H,W,C] -> [B,h_s,w_s,C] per scale (two separable gather
    passes, no atomics).  ``accumulate_into``: existing gradients the result is added to, in place."""
    gout = _check(gout, "gout", 5)
    n_s, bsz, out_h, out_w, ch = gout.shape
    if len(sizes) != n_s:
        raise ValueError("one (h, w) per scale")
    if accumulate_into is not None:
        gin = list(accumulate_into)
        if len(gin) != n_s or any(tuple(t.shape) != (bsz, h, w, ch) or not t.is_contiguous()
                                  for t, (h, w) in zip(gin, sizes)):
            raise ValueError("accumulate_into: one contiguous [B,h_s,w_s,C] tensor per scale")
    else:
        gin = [torch.empty(bsz, h, w, ch, dtype=torch.float32, device=gout.device) for h, w in sizes]
    hs, ws_ = _ints([h for h, _ in sizes]), _ints([w for _, w in sizes])
    with torch.cuda.device(gout.device):
        n_ws = _lib.load().sfm_pyramid_upsample_bwd_workspace_floats(hs, ws_, n_s, bsz, ch, out_h, out_w)
        if n_ws < 0:
            raise ValueError(f"unsupported pyramid shape (rc {n_ws})")
        ws = torch.empty(n_ws, dtype=torch.float32, device=gout.device)
        call("sfm_pyramid_upsample_bwd", ptr(gout), hs, ws_, n_s, bsz, ch, out_h, out_w,
             int(accumulate_into is not None), _ptrs(gin), ptr(ws), stream_of(gout))
    return gin


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_h, out_w):
        bsz, h, w, ch = x.shape
        ctx.dims = (bsz, h, w, ch, out_h, out_w)
        out = torch.empty(bsz, out_h, out_w, ch, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            call("sfm_resize_bilinear_fwd", ptr(x), bsz, h, w, ch, out_h, out_w, ptr(out), stream_of(x))
        return out

    @staticmethod
    def backward(ctx, g):
        bsz, h, w, ch, out_h, out_w = ctx.dims
        g = g.contiguous()
        gin = torch.empty(bsz, h, w, ch, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            ws = torch.empty(_lib.load().sfm_resize_bilinear_bwd_workspace_floats(bsz, h, w, ch, out_h, out_w),
                             dtype=torch.float32, device=g.device)
            call("sfm_resize_bilinear_bwd", ptr(g), bsz, h, w, ch, out_h, out_w, ptr(gin), ptr(ws), stream_of(g))
        return gin, None, None


def resize_bilinear(x_nhwc: Tensor, out_h: int, out_w: int) -> Tensor:
    """``tf.image.resize_bilinear(x, [out_h, out_w])`` with TF-1 defaults (align_corners=False, no
    half-pixel centres; SfMLearner_uni.py:91,113; nets.py:454,465).  Upsampling only, as in the reference;
    channels: 1, 2, 3, 4, 6, 9, 12 or any multiple of 4."""
    x = _check(x_nhwc, "images", 4)
    ch = x.shape[3]
    if ch not in SUPPORTED_C and ch % 4:
        raise ValueError(f"channel count {ch}: must be one of {SUPPORTED_C} or a multiple of 4")
    if x.shape[1] > out_h or x.shape[2] > out_w:
        raise ValueError("upsampling only (SfMLearner_uni.py:91,113; nets.py:454): the output must not be smaller")
    return _Resize.apply(x, int(out_h), int(out_w))


# ----------------------------------------------------------------------------- disp normalisation
def pyramid_normalize_disp(disp: Sequence[Tensor]):
    """-> (pred_disp list, pred_depth list, mean [S,B]); no autograd."""
    disp = _check_maps(disp, "pred_disp", 1)
    bsz = disp[0].shape[0]
    dn = [torch.empty_like(d) for d in disp]
    dp = [torch.empty_like(d) for d in disp]
    mean = torch.empty(len(disp), bsz, dtype=torch.float32, device=disp[0].device)
    n = _ints([d[0].numel() for d in disp])
    with torch.cuda.device(mean.device):
        ws = torch.empty(_lib.load().sfm_disp_normalize_workspace_floats(n, len(disp), bsz), dtype=torch.float32,
                         device=mean.device)
        call("sfm_disp_normalize_fwd", _ptrs(disp), n, len(disp), bsz, _ptrs(dn), _ptrs(dp), ptr(mean), ptr(ws),
             stream_of(mean))
    return dn, dp, mean


def pyramid_normalize_disp_grad(disp: Sequence[Tensor], mean: Tensor, g_disp_norm: Sequence[Optional[Tensor]],
                                g_depth: Sequence[Optional[Tensor]]) -> List[Tensor]:
    bsz = disp[0].shape[0]
    g = [torch.empty_like(d) for d in disp]
    gu = [None if t is None else t.contiguous() for t in g_disp_norm]
    gd = [None if t is None else t.contiguous() for t in g_depth]
    n = _ints([d[0].numel() for d in disp])
    with torch.cuda.device(mean.device):
        ws = torch.empty(_lib.load().sfm_disp_normalize_workspace_floats(n, len(disp), bsz), dtype=torch.float32,
                         device=mean.device)
        call("sfm_disp_normalize_bwd", _ptrs(disp), ptr(mean), _ptrs(gu), _ptrs(gd), n, len(disp), bsz, _ptrs(g),
             ptr(ws), stream_of(mean))
    return g


class _NormalizeDisp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *disp):
        dn, dp, mean = pyramid_normalize_disp(disp)
        ctx.save_for_backward(mean, *[d.contiguous() for d in disp])
        return tuple(dn) + tuple(dp)

    @staticmethod
    def backward(ctx, *g):
        mean, *disp = ctx.saved_tensors
        n = len(disp)
        return tuple(pyramid_normalize_disp_grad(disp, mean, g[:n], g[n:]))


def normalize_disp(disp: Sequence[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
    """``pred_disp = [d / tf.reduce_mean(d, axis=[1,2,3], keep_dims=True) for d in pred_disp]`` and
    ``pred_depth = [1. / d for d in pred_disp]`` (SfMLearner_uni.py:40, 47), every scale in one launch."""
    disp = _check_maps(disp, "pred_disp", 1)
    out = _NormalizeDisp.apply(*disp)
    return list(out[:len(disp)]), list(out[len(disp):])


# ----------------------------------------------------------------------------- smoothness
def smooth_terms(maps: Sequence[Tensor], w_smooth: Sequence[float], w_abs: Sequence[float], global_batch: int,
                 need_grad: bool, gscale: float = 1.0, out_f64: Optional[Tensor] = None):
    """sum_m w_smooth[m] compute_smooth_loss(maps[m]) + w_abs[m] mean|maps[m]| -> (loss [] , grads or None).
    Means divide by the global element count (``global_batch`` in place of the local batch size).
    ``out_f64``: a one-element fp64 tensor (a slot of the all-reduce buffer) that receives the value as well."""
    maps = [_check(m, f"maps[{k}]", 4) for k, m in enumerate(maps)]
    if not 1 <= len(maps) <= MAX_MAPS:
        raise ValueError(f"between 1 and {MAX_MAPS} maps")
    bsz = maps[0].shape[0]
    for m in maps:
        if m.shape[0] != bsz or m.device != maps[0].device:
            raise ValueError("every map must share batch size and device")
        if m.shape[1] < 3 or m.shape[2] < 3:
            raise ValueError("compute_smooth_loss takes second differences: height and width must be >= 3")
    dev = maps[0].device
    h, w, c = (_ints([m.shape[k] for m in maps]) for k in (1, 2, 3))
    lib = _lib.load()
    grads = [torch.empty_like(m) for m in maps] if need_grad else None
    loss = torch.empty((), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty(lib.sfm_smooth_workspace_floats(h, w, c, len(maps), bsz), dtype=torch.float32, device=dev)
        call("sfm_smooth_loss", _ptrs(maps), _ptrs(grads) if grads else None, h, w, c, _floats(w_smooth), _floats(w_abs),
             len(maps), bsz, int(global_batch), float(gscale), ptr(ws), ptr(loss), ptr(out_f64), stream_of(loss))
    return loss, grads


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred):
        loss, grads = smooth_terms([pred], [1.0], [0.0], pred.shape[0], ctx.needs_input_grad[0])
        if grads:
            ctx.save_for_backward(grads[0])
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g


def compute_smooth_loss(pred_disp: Tensor) -> Tensor:
    """``SfMLearner.compute_smooth_loss`` (SfMLearner_uni.py:205-219): value and gradient in one pass."""
    return _Smooth.apply(_check(pred_disp, "pred_disp", 4))


def smooth_weights(num_scales: int, smooth_weight: float, explain_reg_weight: float):
    """(w_smooth, w_abs) of the maps [pred_disp[0..S), pred_exp_logits[0..S)] in the scale loop
    (SfMLearner_uni.py:94-101)."""
    ws = [smooth_weight / 2 ** s for s in range(num_scales)] + [explain_reg_weight / 2 ** s for s in range(num_scales)]
    wa = [0.0] * num_scales + [explain_reg_weight / 2 ** s for s in range(num_scales)]
    return ws, wa
