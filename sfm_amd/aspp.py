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

import weakref

import torch

from . import _lib
from ._lib import call, ptr, stream_of
from .utils import _check, _same_device

Tensor = torch.Tensor
_workspaces = {}
_grad_sinks = {}  # id of a registered parameter -> (weakref to it, GradBuckets, bucket index, offset, shape)


def _sink_of(param):
    """the GradBuckets slot of this very tensor object (an id that outlived its tensor and was reused matches nothing)"""
    if param is None:
        return None
    slot = _grad_sinks.get(id(param))
    if slot is None:
        return None
    if slot[0]() is not param:
        del _grad_sinks[id(param)]
        return None
    return slot


def _grad_out(param: Tensor) -> Tensor:
    """Where a weight-gradient kernel writes: a view into the parameter's slot of a ``GradBuckets`` if it is registered
    in one (the gradient is then born inside the flat all-reduce buffer), else a new tensor.  The slot is handed out
    once per step: a second gradient of the same parameter before the next ``allreduce()`` (a weight used twice in one
    graph, or gradient accumulation over several backward passes) gets a new tensor, which autograd ADDS to the
    first - kernels overwrite their output, so writing the slot again would lose what it holds (and with ``.grad``
    aliasing the slot, ``grad += slot`` would double the new gradient instead of accumulating)."""
    slot = _sink_of(param)
    if slot is None:
        return torch.empty_like(param)
    _, buckets, k, off, shape = slot
    view = buckets.flat[k][off:off + param.numel()].view(shape)
    if (k, off) in buckets._written or (param.grad is not None and param.grad.data_ptr() == view.data_ptr()):
        buckets._written.add((k, off))  # (an aliased .grad from an earlier step: what autograd adds lands in the slot)
        return torch.empty_like(param)
    buckets._written.add((k, off))
    return view


def _workspace(B, H, W, Cin, Cout, device) -> Tensor:
    """One scratch buffer per layer shape and device (split-K slabs, flipped weights)."""
    # per stream as well: calls sharing a workspace must be stream-ordered (include/sfm_amd.h)
    key = (B, H, W, Cin, Cout, str(device), torch.cuda.current_stream(device).cuda_stream)
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
                gw = _grad_out(w)
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
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, decay, epsilon, act):
        C = x.shape[-1]
        M = x.numel() // C
        lib = _lib.load()
        n_ws = lib.sfm_bn_elu_workspace_floats(M, C)
        if n_ws < 0:
            raise ValueError(f"unsupported batch-norm shape M={M} C={C} (C must be a multiple of 4)")
        with torch.cuda.device(x.device):
            ws = torch.empty(n_ws // 2 + 1, dtype=torch.float64, device=x.device)  # fp64 partials: 8-byte aligned
            y = torch.empty_like(x)
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            call("sfm_bn_elu_fwd", ptr(x), ptr(gamma), ptr(beta), M, C, float(epsilon), float(decay), int(act), ptr(moving_mean),
                 ptr(moving_var), ptr(y), ptr(mean), ptr(invstd), ptr(ws), stream_of(x))
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.beta_ref = beta if _sink_of(beta) is not None else None  # only to find its slot in a GradBuckets
        ctx.act = int(act)
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
            ggamma = _grad_out(gamma)
            gbeta = _grad_out(ctx.beta_ref) if ctx.beta_ref is not None else torch.empty_like(gamma)
            call("sfm_bn_elu_bwd", ptr(x), ptr(y), ptr(gy), ptr(gamma), ptr(mean), ptr(invstd), M, C, ctx.act, ptr(gx), ptr(ggamma),
                 ptr(gbeta), ptr(ws), stream_of(x))
        return gx, ggamma, gbeta, None, None, None, None, None


class _SyncBatchNormElu(torch.autograd.Function):
    """batch norm + ELU with statistics over all ranks of ``group``: all-gather of (n, mean, M2) per channel in the
    forward, all-reduce of (sum dz xhat, sum dz) in the backward - two small collectives per layer and direction."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, decay, epsilon, act, group, stats_override, global_rows):
        import torch.distributed as dist
        C = x.shape[-1]
        M = x.numel() // C
        lib = _lib.load()
        n_ws = lib.sfm_bn_elu_workspace_floats(M, C)
        if n_ws < 0:
            raise ValueError(f"unsupported batch-norm shape M={M} C={C} (C must be a multiple of 4)")
        with torch.cuda.device(x.device):
            ws = torch.empty(n_ws // 2 + 1, dtype=torch.float64, device=x.device)
            if stats_override is None:
                world = dist.get_world_size(group)
                stats_all = torch.empty(world, 3, C, dtype=torch.float64, device=x.device)
                mine = stats_all[dist.get_rank(group)]
                call("sfm_bn_local_stats", ptr(x), M, C, ptr(mine), ptr(ws), stream_of(x))
                dist.all_gather_into_tensor(stats_all.view(-1), mine.reshape(-1).clone(), group=group)
            else:  # (tests: the gathered triples of a rehearsed world, computed by the caller)
                stats_all = stats_override.contiguous()
                world = stats_all.shape[0]
            y = torch.empty_like(x)
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            call("sfm_bn_elu_apply_stats", ptr(x), ptr(gamma), ptr(beta), ptr(stats_all), world, M, C, float(epsilon),
                 float(decay), int(act), ptr(moving_mean), ptr(moving_var), ptr(y), ptr(mean), ptr(invstd), stream_of(x))
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.beta_ref = beta if _sink_of(beta) is not None else None
        ctx.act, ctx.group = int(act), group
        # rows of all ranks: given by the caller, else read from the gathered triples (every rank holds the same ones;
        # no further collective, but one host synchronisation per call)
        ctx.m_global = int(global_rows) if global_rows is not None else int(round(float(stats_all[:, 0, 0].sum())))
        ctx.rehearsed = stats_override is not None
        ctx.mark_non_differentiable(mean, invstd)
        return y, mean, invstd

    @staticmethod
    def backward(ctx, gy, _gm, _gi):
        import torch.distributed as dist
        x, y, gamma, mean, invstd = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        gy = gy.contiguous()
        with torch.cuda.device(x.device):
            ws = torch.empty(_lib.load().sfm_bn_elu_workspace_floats(M, C) // 2 + 1, dtype=torch.float64, device=x.device)
            gx = torch.empty_like(x)
            ggamma = _grad_out(gamma)
            gbeta = _grad_out(ctx.beta_ref) if ctx.beta_ref is not None else torch.empty_like(gamma)
            call("sfm_bn_elu_bwd_sums", ptr(x), ptr(y), ptr(gy), ptr(mean), ptr(invstd), M, C, ctx.act, ptr(ggamma), ptr(gbeta),
                 ptr(ws), stream_of(x))
            both = torch.stack([ggamma, gbeta])  # a copy: the local sums stay this rank's parameter gradients
            if ctx.rehearsed:
                if _SyncBatchNormElu.rehearsal_sums is None:
                    raise RuntimeError("rehearsed backward without rehearsal_sums")
                both = _SyncBatchNormElu.rehearsal_sums(both)
            else:
                dist.all_reduce(both, group=ctx.group)
            call("sfm_bn_elu_bwd_apply", ptr(x), ptr(y), ptr(gy), ptr(gamma), ptr(mean), ptr(invstd), ptr(both[0]), ptr(both[1]),
                 M, ctx.m_global, C, ctx.act, ptr(gx), stream_of(x))
        return gx, ggamma, gbeta, None, None, None, None, None, None, None, None

    rehearsal_sums = None  # tests: callable mapping this "rank"'s [2, C] sums to the sums over the rehearsed world


def batch_norm_elu(x: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Tensor = None, moving_var: Tensor = None,
                   decay: float = BN_DECAY, epsilon: float = BN_EPSILON, activation: bool = True,
                   process_group=None, _stats_all: Tensor = None, global_rows: int = None) -> Tensor:
    """``tf.nn.elu(slim.batch_norm(x, decay, epsilon, scale=True, fused=True, is_training=True,
    updates_collections=None))`` - the normaliser + activation of the ASPP arg-scope (nets.py:653-656).
    x [..., C] NHWC; gamma, beta [C]; moving_mean / moving_var [C] are updated in place when given.
    ``activation=False``: the batch norm alone (``slim.conv2d(..., activation_fn=None)``, nets.py:455, 470).
    ``process_group``: statistics over the batches of ALL ranks of the group (synchronised batch norm: the
    single-device graph on the global batch); default ``None`` = this rank's batch only.  ``global_rows``: the number
    of rows (B*H*W) over all ranks when the caller knows it (fixed shards: world * rows); omitted, it is read from the
    gathered statistics at the price of one host synchronisation per call."""
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
    if process_group is not None or _stats_all is not None:
        return _SyncBatchNormElu.apply(x, gamma, beta, moving_mean, moving_var, float(decay), float(epsilon), bool(activation),
                                       process_group, _stats_all, global_rows)[0]
    return _BatchNormElu.apply(x, gamma, beta, moving_mean, moving_var, float(decay), float(epsilon), bool(activation))[0]


def conv2d_bn_elu(x: Tensor, w: Tensor, gamma: Tensor, beta: Tensor, rate: int, moving_mean: Tensor = None,
                  moving_var: Tensor = None, decay: float = BN_DECAY, epsilon: float = BN_EPSILON,
                  is_training: bool = True, activation: bool = True, process_group=None) -> Tensor:
    """One ``slim.conv2d(x, Cout, [3, 3], rate=rate)`` of ``disp_bottleneck_dense`` under its arg-scope
    (nets.py:457, 458, 463): dilated convolution (no bias: a normaliser is set), batch norm, ELU.
    ``is_training=False`` (the inference graphs, test_kitti_depth_con.py:19-65): batch norm on the moving statistics
    and the ELU are applied where the MFMA kernel stores its sums - one pass, forward only."""
    if is_training:
        return batch_norm_elu(dilated_conv3x3(x, w, rate), gamma, beta, moving_mean, moving_var, decay, epsilon, activation,
                              process_group=process_group)
    x = _check(x, "x", 4)
    w = _check(w, "w", 4)
    if w.shape[0] != 3 or w.shape[1] != 3 or w.shape[2] != x.shape[3]:
        raise ValueError(f"w must be [3, 3, {x.shape[3]}, Cout], got {tuple(w.shape)}")
    if moving_mean is None or moving_var is None:
        raise ValueError("is_training=False needs moving_mean and moving_var")
    B, H, W, Cin = x.shape
    Cout = w.shape[3]
    vecs = [t.detach().contiguous() for t in (gamma, beta, moving_mean, moving_var)]
    if any(tuple(t.shape) != (Cout,) or t.dtype != torch.float32 or t.device != x.device for t in vecs):
        raise ValueError(f"gamma, beta, moving_mean, moving_var must be fp32 [{Cout}] on the device of x")
    with torch.cuda.device(x.device), torch.no_grad():
        ws = _workspace(B, H, W, Cin, Cout, x.device)
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
        call("sfm_aspp_conv3x3_bn_act_infer", ptr(x), ptr(w), ptr(vecs[0]), ptr(vecs[1]), ptr(vecs[2]), ptr(vecs[3]),
             float(epsilon), int(activation), B, H, W, Cin, Cout, int(rate), ptr(out), ptr(ws), stream_of(x))
    return out


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        B, H, W, Cin = x.shape
        Cout = w.shape[3]
        with torch.cuda.device(x.device):
            ws = _workspace(B, H, W, Cin, Cout, x.device)
            out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
            call("sfm_aspp_conv1x1_fwd", ptr(x), ptr(w), B, H, W, Cin, Cout, ptr(out), ptr(ws), stream_of(x))
        ctx.save_for_backward(x, w)
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
                call("sfm_aspp_conv1x1_bwd_data", ptr(gout), ptr(w), B, H, W, Cin, Cout, ptr(gx), ptr(ws), stream_of(x))
            if ctx.needs_input_grad[1]:
                gw = _grad_out(w)
                call("sfm_aspp_conv1x1_bwd_weight", ptr(x), ptr(gout), B, H, W, Cin, Cout, ptr(gw), ptr(ws), stream_of(x))
        return gx, gw


def conv1x1(x: Tensor, w: Tensor) -> Tensor:
    """``slim.conv2d(x, w.shape[3], [1, 1])`` without normaliser / activation / bias: x [B,H,W,Cin], w [1,1,Cin,Cout]
    (nets.py:455, 456, 461, 466, 470).  Channels must be multiples of 32."""
    x = _check(x, "x", 4)
    w = _check(w, "w", 4)
    if w.shape[0] != 1 or w.shape[1] != 1 or w.shape[2] != x.shape[3]:
        raise ValueError(f"w must be [1, 1, {x.shape[3]}, Cout], got {tuple(w.shape)}")
    _same_device(x, w)
    return _Conv1x1.apply(x, w)


def _slim_conv(x, p, name, rate=None, activation=True, process_group=None):
    """one slim.conv2d of the block under its arg-scope: p[name] = dict(w, gamma, beta[, moving_mean, moving_var])"""
    q = p[name]
    y = conv1x1(x, q["w"]) if rate is None else dilated_conv3x3(x, q["w"], rate)
    return batch_norm_elu(y, q["gamma"], q["beta"], q.get("moving_mean"), q.get("moving_var"), activation=activation,
                          process_group=process_group)


# ---- one decoder level as ONE autograd node, without a single concat ---------------------------------------------------
# The activations tf.concat joins (nets.py:458, 463, 467, 469) are written by their batch-norm kernels straight into channel
# slices of one [B,H,W,4C or 5C] buffer; the convolutions that read a concat of some of them read a slice of that buffer
# (row stride = the buffer's width); in the backward the gradient of the buffer comes out of the last 1x1 convolution in
# one piece, every batch norm reads its slice of it and ADDS the input gradients of the convolutions that also read its
# activation while loading (up to three sources): no torch.cat, no slice copies, no gradient-accumulation passes.

def _raw_resize(x, H, W):
    B, h, w, C = x.shape
    out = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
    call("sfm_resize_bilinear_fwd", ptr(x), B, h, w, C, H, W, ptr(out), stream_of(x))
    return out


def _raw_resize_bwd(g, h, w):
    B, H, W, C = g.shape
    gin = torch.empty(B, h, w, C, dtype=torch.float32, device=g.device)
    ws = torch.empty(_lib.load().sfm_resize_bilinear_bwd_workspace_floats(B, h, w, C, H, W), dtype=torch.float32, device=g.device)
    call("sfm_resize_bilinear_bwd", ptr(g), B, h, w, C, H, W, ptr(gin), ptr(ws), stream_of(g))
    return gin


def _raw_conv(x, ld, w, dims, rate):
    """x: a tensor whose data pointer is the first channel of the operand at pixel 0; ld: its row stride in floats"""
    B, H, W = dims
    taps, Cin, Cout = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
    ws = _workspace(B, H, W, Cin, Cout, x.device)
    out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    if taps == 1:
        assert ld == Cin
        call("sfm_aspp_conv1x1_fwd", ptr(x), ptr(w), B, H, W, Cin, Cout, ptr(out), ptr(ws), stream_of(x))
    else:
        call("sfm_aspp_conv3x3_fwd_ld", ptr(x), ld, ptr(w), B, H, W, Cin, Cout, rate, ptr(out), ptr(ws), stream_of(x))
    return out


def _raw_conv_bwd(x, ld, w, gout, dims, rate):
    """-> (dL/dx dense [B,H,W,Cin], dL/dw in the parameter's gradient slot)"""
    B, H, W = dims
    taps, Cin, Cout = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
    ws = _workspace(B, H, W, Cin, Cout, gout.device)
    gx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=gout.device)
    gw = _grad_out(w)
    st = stream_of(gout)
    if taps == 1:
        call("sfm_aspp_conv1x1_bwd_data", ptr(gout), ptr(w), B, H, W, Cin, Cout, ptr(gx), ptr(ws), st)
        call("sfm_aspp_conv1x1_bwd_weight", ptr(x), ptr(gout), B, H, W, Cin, Cout, ptr(gw), ptr(ws), st)
    else:
        call("sfm_aspp_conv3x3_bwd_data", ptr(gout), ptr(w), B, H, W, Cin, Cout, rate, ptr(gx), ptr(ws), st)
        call("sfm_aspp_conv3x3_bwd_weight_ld", ptr(x), ld, ptr(gout), B, H, W, Cin, Cout, rate, ptr(gw), ptr(ws), st)
    return gx, gw


def _raw_bn(x, q, act, y, ld_y):
    """batch norm (+ ELU) of the dense x into y (a tensor of its own or a channel slice, row stride ld_y) -> (mean, invstd)"""
    C = x.shape[-1]
    M = x.numel() // C
    n_ws = _lib.load().sfm_bn_elu_workspace_floats(M, C)
    if n_ws < 0:
        raise ValueError(f"unsupported batch-norm shape M={M} C={C} (C must be a multiple of 4)")
    ws = torch.empty(n_ws // 2 + 1, dtype=torch.float64, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    call("sfm_bn_elu_fwd_ld", ptr(x), ptr(q["gamma"]), ptr(q["beta"]), M, C, float(BN_EPSILON), float(BN_DECAY), int(act),
         ptr(q.get("moving_mean")), ptr(q.get("moving_var")), ptr(y), ld_y, ptr(mean), ptr(invstd), ptr(ws), stream_of(x))
    return mean, invstd


def _raw_bn_bwd(x, y, ld_y, srcs, q, stats, act):
    """srcs: [(tensor whose data pointer is the slice's first channel at row 0, row stride)], summed while loading"""
    import ctypes
    C = x.shape[-1]
    M = x.numel() // C
    ws = torch.empty(_lib.load().sfm_bn_elu_workspace_floats(M, C) // 2 + 1, dtype=torch.float64, device=x.device)
    gx = torch.empty_like(x)
    gg, gb = _grad_out(q["gamma"]), _grad_out(q["beta"])
    n = len(srcs)
    gy = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in srcs])
    ld = (ctypes.c_int * n)(*[int(l) for _, l in srcs])
    call("sfm_bn_elu_bwd_ld", ptr(x), ptr(y), ld_y, gy, ld, n, ptr(q["gamma"]), ptr(stats[0]), ptr(stats[1]), M, C, int(act),
         ptr(gx), ptr(gg), ptr(gb), ptr(ws), stream_of(x))
    return gx, gg, gb


_LEVEL_CONVS = ("up", "sup", "aspp_up", "aspp_up2", "skip", "skip2", "out")


class _DenseLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, concat, skip_feat, skip2, *flat):
        names, rates, use_skip, use_skip2, extras = cfg
        ctx.set_materialize_grads(False)
        # name -> dict(w, gamma, beta) / dict(gamma, beta) for 'skip_bn', plus the moving statistics (not differentiated)
        P, k = {}, 0
        for nm in names:
            keys = ("gamma", "beta") if nm == "skip_bn" else ("w", "gamma", "beta")
            P[nm] = dict(zip(keys, flat[k:k + len(keys)]), **extras.get(nm, {}))
            k += len(keys)
        B, h, w_, _ = concat.shape
        H, W = 2 * h, 2 * w_
        dims = (B, H, W)
        C = P["sup"]["w"].shape[3]
        nC = (5 if use_skip2 else 4) * C
        with torch.cuda.device(concat.device):
            u0 = _raw_resize(concat, H, W)                                                      # :454
            a_up = _raw_conv(u0, u0.shape[3], P["up"]["w"], dims, 0)                            # :455
            upnet = torch.empty_like(a_up)
            st = {"up": _raw_bn(a_up, P["up"], 0, upnet, upnet.shape[3])}
            X = torch.empty(B, H, W, nC, dtype=torch.float32, device=concat.device)            # the level's ONE concat buffer
            sl = lambda a, b: X[..., a * C:b * C]                                               # noqa: E731  (views: pointers only)
            a_sup = _raw_conv(upnet, upnet.shape[3], P["sup"]["w"], dims, 0)                    # :456
            st["sup"] = _raw_bn(a_sup, P["sup"], 1, sl(0, 1), nC)
            a_au = _raw_conv(upnet, upnet.shape[3], P["aspp_up"]["w"], dims, rates[0])          # :457
            st["aspp_up"] = _raw_bn(a_au, P["aspp_up"], 1, sl(1, 2), nC)
            a_au2 = _raw_conv(sl(0, 2), nC, P["aspp_up2"]["w"], dims, rates[1])                 # :458 reads [supnet, aspp_up]
            st["aspp_up2"] = _raw_bn(a_au2, P["aspp_up2"], 1, sl(2, 3), nC)
            sk = None
            if use_skip:
                sk = torch.empty_like(skip_feat)
                st["skip_bn"] = _raw_bn(skip_feat, P["skip_bn"], 1, sk, sk.shape[3])            # :460
                a_sk = _raw_conv(sk, sk.shape[3], P["skip"]["w"], dims, 0)                      # :461
            else:
                a_sk = _raw_conv(sl(1, 3), nC, P["skip"]["w"], dims, rates[1] * 2)              # :463 reads [aspp_up, aspp_up2]
            st["skip"] = _raw_bn(a_sk, P["skip"], 1, sl(3, 4), nC)
            s2 = a_s2 = None
            if use_skip2:
                s2 = _raw_resize(skip2, H, W)                                                   # :465
                a_s2 = _raw_conv(s2, s2.shape[3], P["skip2"]["w"], dims, 0)                     # :466
                st["skip2"] = _raw_bn(a_s2, P["skip2"], 1, sl(4, 5), nC)
            a_out = _raw_conv(X, nC, P["out"]["w"], dims, 0)                                    # :470 reads the whole buffer
            xconcat = torch.empty_like(a_out)
            st["out"] = _raw_bn(a_out, P["out"], 0, xconcat, C)
        ctx.cfg, ctx.P, ctx.st = cfg, P, st
        ctx.acts = (u0, a_up, a_sup, a_au, a_au2, sk, a_sk, s2, a_s2, X, a_out)
        ctx.io = (concat.shape, skip2.shape if use_skip2 else None)
        ctx.save_for_backward(xconcat, upnet, *([skip_feat] if use_skip else []))
        return xconcat, upnet

    @staticmethod
    def backward(ctx, g_x, g_up):
        names, rates, use_skip, use_skip2, _ = ctx.cfg
        P, st = ctx.P, ctx.st
        xconcat, upnet = ctx.saved_tensors[:2]
        skip_feat = ctx.saved_tensors[2] if use_skip else None
        u0, a_up, a_sup, a_au, a_au2, sk, a_sk, s2, a_s2, X, a_out = ctx.acts
        concat_shape, skip2_shape = ctx.io
        B, H, W, C = xconcat.shape
        dims = (B, H, W)
        nC = X.shape[3]
        G = {}
        with torch.cuda.device(xconcat.device):
            if g_x is None:
                g_x = torch.zeros_like(xconcat)
            g_x = g_x.contiguous()
            g_a, G["out.gamma"], G["out.beta"] = _raw_bn_bwd(a_out, xconcat, C, [(g_x, C)], P["out"], st["out"], 0)
            gX, G["out.w"] = _raw_conv_bwd(X, nC, P["out"]["w"], g_a, dims, 0)                  # dL/d(concat buffer), one piece
            gsl = lambda a: (gX[..., a * C:(a + 1) * C], nC)                                    # noqa: E731
            ysl = lambda a: X[..., a * C:(a + 1) * C]                                           # noqa: E731
            g_skip2 = None
            if use_skip2:
                g_a, G["skip2.gamma"], G["skip2.beta"] = _raw_bn_bwd(a_s2, ysl(4), nC, [gsl(4)], P["skip2"], st["skip2"], 1)
                g_s2, G["skip2.w"] = _raw_conv_bwd(s2, s2.shape[3], P["skip2"]["w"], g_a, dims, 0)
                g_skip2 = _raw_resize_bwd(g_s2, skip2_shape[1], skip2_shape[2])
            g_a, G["skip.gamma"], G["skip.beta"] = _raw_bn_bwd(a_sk, ysl(3), nC, [gsl(3)], P["skip"], st["skip"], 1)
            g_skip_feat, t4 = None, None
            if use_skip:
                g_sk, G["skip.w"] = _raw_conv_bwd(sk, sk.shape[3], P["skip"]["w"], g_a, dims, 0)
                g_skip_feat, G["skip_bn.gamma"], G["skip_bn.beta"] = _raw_bn_bwd(skip_feat, sk, sk.shape[3], [(g_sk, sk.shape[3])],
                                                                                P["skip_bn"], st["skip_bn"], 1)
            else:
                t4, G["skip.w"] = _raw_conv_bwd(ysl(1), nC, P["skip"]["w"], g_a, dims, rates[1] * 2)   # -> [aspp_up, aspp_up2]
            src = [gsl(2)] + ([(t4[..., C:], 2 * C)] if t4 is not None else [])
            g_a, G["aspp_up2.gamma"], G["aspp_up2.beta"] = _raw_bn_bwd(a_au2, ysl(2), nC, src, P["aspp_up2"], st["aspp_up2"], 1)
            t5, G["aspp_up2.w"] = _raw_conv_bwd(ysl(0), nC, P["aspp_up2"]["w"], g_a, dims, rates[1])   # -> [supnet, aspp_up]
            src = [gsl(1), (t5[..., C:], 2 * C)] + ([(t4, 2 * C)] if t4 is not None else [])
            g_a, G["aspp_up.gamma"], G["aspp_up.beta"] = _raw_bn_bwd(a_au, ysl(1), nC, src, P["aspp_up"], st["aspp_up"], 1)
            g_u1, G["aspp_up.w"] = _raw_conv_bwd(upnet, upnet.shape[3], P["aspp_up"]["w"], g_a, dims, rates[0])
            g_a, G["sup.gamma"], G["sup.beta"] = _raw_bn_bwd(a_sup, ysl(0), nC, [gsl(0), (t5, 2 * C)], P["sup"], st["sup"], 1)
            g_u2, G["sup.w"] = _raw_conv_bwd(upnet, upnet.shape[3], P["sup"]["w"], g_a, dims, 0)
            Cu = upnet.shape[3]
            src = [(g_u1, Cu), (g_u2, Cu)] + ([(g_up.contiguous(), Cu)] if g_up is not None else [])
            g_a, G["up.gamma"], G["up.beta"] = _raw_bn_bwd(a_up, upnet, Cu, src, P["up"], st["up"], 0)
            g_u0, G["up.w"] = _raw_conv_bwd(u0, u0.shape[3], P["up"]["w"], g_a, dims, 0)
            g_concat = _raw_resize_bwd(g_u0, concat_shape[1], concat_shape[2])
        flat = []
        for nm in names:
            for key in (("gamma", "beta") if nm == "skip_bn" else ("w", "gamma", "beta")):
                flat.append(G[f"{nm}.{key}"])
        return (None, g_concat, g_skip_feat, g_skip2, *flat)


def _check_level(params, concat, skip_feat, use_skip, use_skip2, skip2, need_moving=False):
    """argument checks of the concat-free level (the kernels index with these sizes and check nothing themselves) ->
    (names, flat parameter list, moving statistics per name, concat, skip_feat, skip2)"""
    names = [n for n in _LEVEL_CONVS if n != "skip2" or use_skip2] + (["skip_bn"] if use_skip else [])
    concat = _check(concat, "concat", 4)
    B, h, w_ = concat.shape[:3]
    C, Cin = params["sup"]["w"].shape[3], params["up"]["w"].shape[3]
    nsl = 5 if use_skip2 else 4
    if use_skip:
        skip_feat = _check(skip_feat, "skip_feat", 4)
        if tuple(skip_feat.shape[:3]) != (B, 2 * h, 2 * w_):
            raise ValueError(f"skip_feat must be [B, {2 * h}, {2 * w_}, C], got {tuple(skip_feat.shape)}")
    if use_skip2:
        skip2 = _check(skip2, "skip2", 4)
        if skip2.shape[0] != B or skip2.shape[1] > 2 * h or skip2.shape[2] > 2 * w_:
            raise ValueError(f"skip2 {tuple(skip2.shape)} does not upsample to [{B}, {2 * h}, {2 * w_}]")
    # input channels of every convolution of the block, as the concat-free wiring assumes them
    want = {"up": (1, concat.shape[3], Cin), "sup": (1, Cin, C), "aspp_up": (3, Cin, C), "aspp_up2": (3, 2 * C, C),
            "skip": (1, skip_feat.shape[3], C) if use_skip else (3, 2 * C, C), "out": (1, nsl * C, C)}
    if use_skip2:
        want["skip2"] = (1, skip2.shape[3], C)
    flat, extras = [], {}
    for nm in names:
        q = params[nm]
        vecs = [q["gamma"], q["beta"]]
        if nm != "skip_bn":
            k, ci, cv = want[nm]
            wt = _check(q["w"], f"{nm}.w", 4)
            if tuple(wt.shape) != (k, k, ci, cv):
                raise ValueError(f"{nm}.w must be [{k}, {k}, {ci}, {cv}], got {tuple(wt.shape)}")
        else:
            cv = skip_feat.shape[3]
        tensors = {"gamma": vecs[0], "beta": vecs[1], "moving_mean": q.get("moving_mean"), "moving_var": q.get("moving_var")}
        for key, v in tensors.items():
            if v is not None:
                _check(v, f"{nm}.{key}", 1, cv)
        for key, v in {**tensors, **({"w": q["w"]} if nm != "skip_bn" else {})}.items():
            if v is not None and not v.is_contiguous():
                raise ValueError(f"{nm}.{key} must be contiguous (its gradient / update is written in place)")
        _same_device(concat, *([q["w"]] if nm != "skip_bn" else []), *tensors.values())
        flat += vecs if nm == "skip_bn" else [q["w"], *vecs]
        extras[nm] = {k: q[k] for k in ("moving_mean", "moving_var") if q.get(k) is not None}
        if need_moving and len(extras[nm]) != 2:
            raise ValueError(f"is_training=False needs {nm}.moving_mean and {nm}.moving_var")
    return names, flat, extras, concat, skip_feat, skip2


def _dense_level(params, concat, skip_feat, rates, use_skip, use_skip2, skip2):
    names, flat, extras, concat, skip_feat, skip2 = _check_level(params, concat, skip_feat, use_skip, use_skip2, skip2)
    cfg = (tuple(names), tuple(int(r) for r in rates), bool(use_skip), bool(use_skip2), extras)
    return _DenseLevel.apply(cfg, concat, skip_feat if use_skip else None, skip2 if use_skip2 else None, *flat)


def _infer_conv(x, ld, q, dims, rate, act, y, ld_y):
    """one slim.conv2d of the inference graph in ONE kernel pass: convolution, batch norm on the moving statistics, ELU if
    act, written to y (a tensor of its own or a channel slice of the level's concat buffer, row stride ld_y)"""
    B, H, W = dims
    w = q["w"]
    taps, Cin, Cout = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
    ws = _workspace(B, H, W, Cin, Cout, x.device)
    call("sfm_aspp_conv_bn_act_infer_ld", ptr(x), ld, ptr(w), taps, ptr(q["gamma"]), ptr(q["beta"]), ptr(q["moving_mean"]),
         ptr(q["moving_var"]), float(BN_EPSILON), int(act), B, H, W, Cin, Cout, int(rate) if taps == 9 else 1, ptr(y), ld_y,
         ptr(ws), stream_of(x))


def _infer_bn(x, q, act, y, ld_y):
    C = x.shape[-1]
    call("sfm_bn_elu_infer", ptr(x), ptr(q["gamma"]), ptr(q["beta"]), ptr(q["moving_mean"]), ptr(q["moving_var"]),
         x.numel() // C, C, float(BN_EPSILON), int(act), ptr(y), ld_y, stream_of(x))


def _dense_level_infer(params, concat, skip_feat, rates, use_skip, use_skip2, skip2):
    """the level under is_training=False (test_kitti_depth_con.py:19-65): forward only, every slim.conv2d one kernel pass
    (batch norm on the moving statistics + ELU as the convolution's epilogue), outputs written as channel slices of the
    level's one concat buffer - nine launches + two resizes for the whole of nets.py:451-471"""
    names, _, _, concat, skip_feat, skip2 = _check_level(params, concat, skip_feat, use_skip, use_skip2, skip2, need_moving=True)
    P = params
    B, h, w_, _ = concat.shape
    H, W = 2 * h, 2 * w_
    dims = (B, H, W)
    C, Cin = P["sup"]["w"].shape[3], P["up"]["w"].shape[3]
    nC = (5 if use_skip2 else 4) * C
    dev = concat.device
    with torch.no_grad(), torch.cuda.device(dev):
        u0 = _raw_resize(concat, H, W)                                                          # :454
        upnet = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dev)
        _infer_conv(u0, u0.shape[3], P["up"], dims, 0, 0, upnet, Cin)                           # :455
        X = torch.empty(B, H, W, nC, dtype=torch.float32, device=dev)
        sl = lambda a, b: X[..., a * C:b * C]                                                   # noqa: E731
        _infer_conv(upnet, Cin, P["sup"], dims, 0, 1, sl(0, 1), nC)                             # :456
        _infer_conv(upnet, Cin, P["aspp_up"], dims, rates[0], 1, sl(1, 2), nC)                  # :457
        _infer_conv(sl(0, 2), nC, P["aspp_up2"], dims, rates[1], 1, sl(2, 3), nC)               # :458
        if use_skip:
            sk = torch.empty_like(skip_feat)
            _infer_bn(skip_feat, P["skip_bn"], 1, sk, sk.shape[3])                              # :460
            _infer_conv(sk, sk.shape[3], P["skip"], dims, 0, 1, sl(3, 4), nC)                   # :461
        else:
            _infer_conv(sl(1, 3), nC, P["skip"], dims, rates[1] * 2, 1, sl(3, 4), nC)           # :463
        if use_skip2:
            s2 = _raw_resize(skip2, H, W)                                                       # :465
            _infer_conv(s2, s2.shape[3], P["skip2"], dims, 0, 1, sl(4, 5), nC)                  # :466
        xconcat = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        _infer_conv(X, nC, P["out"], dims, 0, 0, xconcat, C)                                    # :470
    return xconcat, upnet


def disp_bottleneck_dense(params: dict, concat: Tensor, skip_feat: Tensor = None, rates=(2, 4), use_skip: bool = True,
                          use_skip2: bool = True, skip2: Tensor = None, process_group=None, fused_level: bool = True,
                          is_training: bool = True):
    """``nets.disp_bottleneck_dense`` (nets.py:451-471), one decoder level of ``disp_aspp_u_dense``, every op a
    hand-written kernel: x2 bilinear upsample, 1x1 and dilated 3x3 convolutions on fp32 MFMA, batch norm (training) +
    ELU.  ``fused_level`` (default; single-rank statistics only): the level is ONE autograd node that never materialises
    a concat - the batch norms write channel slices of one buffer, the convolutions read slices of it, the backward sums
    the gradients of an activation's several readers while loading them (``_DenseLevel``); False, or a ``process_group``:
    the operators composed through autograd with ``torch.cat``.  ``is_training=False`` (the inference graphs,
    test_kitti_depth_con.py:19-65; every parameter dict then needs ``moving_mean`` / ``moving_var``): forward only, every
    ``slim.conv2d`` ONE kernel pass with the batch norm on the moving statistics and the ELU as the convolution's epilogue,
    written as channel slices of the level's concat buffer.  ``params``: name -> dict(w, gamma, beta[, moving_mean, moving_var]) for
    'up' (:455), 'sup' (:456), 'aspp_up' (:457), 'aspp_up2' (:458), 'skip' (:461 or :463), ['skip2' (:466)], 'out'
    (:470) and, with use_skip, 'skip_bn' = dict(gamma, beta, ...) for the batch norm of the skip feature (:460).
    ``skip_feat`` = end_points[skip_block].  Returns (xconcat, upnet) like the reference.  ``process_group``: every
    batch norm of the block takes its statistics over all ranks of the group (see ``batch_norm_elu``)."""
    from .pyramid import resize_bilinear
    import functools
    import os
    if not is_training:
        return _dense_level_infer(params, concat, skip_feat, rates, use_skip, use_skip2, skip2)
    if process_group is None and fused_level and os.environ.get("SFM_LEVEL_FUSED", "1") != "0":
        return _dense_level(params, concat, skip_feat, rates, use_skip, use_skip2, skip2)
    conv = functools.partial(_slim_conv, process_group=process_group)
    h, w = concat.shape[1], concat.shape[2]
    upnet = resize_bilinear(concat, 2 * h, 2 * w)                                        # :454
    upnet = conv(upnet, params, "up", activation=False)                            # :455
    supnet = conv(upnet, params, "sup")                                            # :456
    aspp_up = conv(upnet, params, "aspp_up", rate=rates[0])                        # :457
    aspp_up2 = conv(torch.cat([supnet, aspp_up], dim=3), params, "aspp_up2", rate=rates[1])   # :458
    if use_skip:
        q = params["skip_bn"]
        skip = batch_norm_elu(skip_feat, q["gamma"], q["beta"], q.get("moving_mean"), q.get("moving_var"),
                              process_group=process_group)                               # :460
        skip = conv(skip, params, "skip")                                          # :461
    else:
        skip = conv(torch.cat([aspp_up, aspp_up2], dim=3), params, "skip", rate=rates[1] * 2)   # :463
    parts = [supnet, aspp_up, aspp_up2, skip]
    if use_skip2:
        s2 = resize_bilinear(skip2, 2 * h, 2 * w)                                        # :465
        parts.append(conv(s2, params, "skip2"))                                    # :466
    xconcat = conv(torch.cat(parts, dim=3), params, "out", activation=False)       # :470
    return xconcat, upnet


DISP_SCALING = 10    # nets.py:11
MIN_DISP = 0.01      # nets.py:12


class _DispHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        B, H, W, Cin = x.shape
        with torch.cuda.device(x.device):
            disp = torch.empty(B, H, W, 1, dtype=torch.float32, device=x.device)
            sig = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
            call("sfm_disp_head_fwd", ptr(x), ptr(w), ptr(b), B, H, W, Cin, ptr(disp), ptr(sig), stream_of(x))
        ctx.save_for_backward(x, w, sig)
        ctx.b_ref = b if _sink_of(b) is not None else None
        return disp

    @staticmethod
    def backward(ctx, gdisp):
        x, w, sig = ctx.saved_tensors
        B, H, W, Cin = x.shape
        gdisp = gdisp.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gx = gw = gb = ws = None
        with torch.cuda.device(x.device):
            if need_x:
                gx = torch.empty_like(x)
            if need_w:
                gw = _grad_out(w)
                gb = _grad_out(ctx.b_ref) if ctx.b_ref is not None else torch.empty(1, dtype=torch.float32, device=x.device)
                ws = torch.empty(_lib.load().sfm_disp_head_workspace_floats(B, H, W, Cin), dtype=torch.float32, device=x.device)
            call("sfm_disp_head_bwd", ptr(x), ptr(w), ptr(sig), ptr(gdisp), B, H, W, Cin, ptr(gx), ptr(gw), ptr(gb), ptr(ws),
                 stream_of(x))
        return gx, gw, gb


def disp_head(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """``DISP_SCALING * slim.conv2d(x, 1, [3, 3], activation_fn=tf.sigmoid, normalizer_fn=None) + MIN_DISP`` - the
    disparity heads of ``disp_aspp_u_dense`` (nets.py:678, 683, 689): x [B,H,W,Cin], w [3,3,Cin,1] (HWIO), b [1] ->
    [B,H,W,1].  Cin / 4 must be a power of two <= 64."""
    x = _check(x, "x", 4)
    Cin = x.shape[3]
    w = _check(w, "w", 4)
    b = _check(b, "b", 1, 1)
    if tuple(w.shape) != (3, 3, Cin, 1):
        raise ValueError(f"w must be [3, 3, {Cin}, 1], got {tuple(w.shape)}")
    if Cin % 4 or (Cin // 4) & (Cin // 4 - 1) or Cin > 256:
        raise ValueError(f"Cin / 4 must be a power of two <= 64, got Cin = {Cin}")
    _same_device(x, w, b)
    return _DispHead.apply(x, w, b)


def disp_aspp_u_dense_decoder(end_points: dict, params: dict, process_group=None, is_training: bool = True):
    """The decoder of ``nets.disp_aspp_u_dense`` (nets.py:666-692) from the ResNet end-points to the three disparity
    maps - everything of the network that is not the stock ResNet-v2-50 encoder: batch norm + ELU + 1x1 on block4
    (:672-673), the five ``disp_bottleneck_dense`` levels (:675-686) and the three heads (:678, 683, 689).

    ``end_points``: 'block4' [B,h,w,2048], 'block3' [B,2h,2w,1024], 'block2' [B,4h,4w,512], 'block1' [B,8h,8w,256],
    'root_block' [B,16h,16w,64] (``resnet_v2_50_mod`` at output_stride 32, resnet/resnet_v2.py:287-306; h x w = 4 x 13
    for a 128 x 416 image).  ``params``: 'stem_bn' = dict(gamma, beta[, moving_mean, moving_var]) (:672), 'stem' =
    dict(w [1,1,2048,1024], gamma, beta, ...) (:673), 'concat', 'concat1', 'concat2', 'concat3', 'concat4' = the
    parameter dicts of ``disp_bottleneck_dense``, 'disp2', 'disp3', 'disp4' = dict(w [3,3,C,1], b [1]).
    Returns ``[disp4, disp3, disp2]`` like the reference ([B,32h,32w,1], [B,16h,16w,1], [B,8h,8w,1]).
    ``is_training=False`` (test_kitti_depth_con.py:19-65; ``moving_mean`` / ``moving_var`` in every batch-norm dict): the
    serving path - forward only, no autograd graph, one kernel pass per ``slim.conv2d``, no concat."""
    q = params["stem_bn"]
    if not is_training:
        x4 = _check(end_points["block4"], "block4", 4)
        for nm in ("stem_bn", "stem"):
            if params[nm].get("moving_mean") is None or params[nm].get("moving_var") is None:
                raise ValueError(f"is_training=False needs {nm}.moving_mean and {nm}.moving_var")
        with torch.no_grad(), torch.cuda.device(x4.device):
            B, h, w_, c4 = x4.shape
            t = torch.empty_like(x4)
            _infer_bn(x4, q, 1, t, c4)                                                                       # :672
            co = params["stem"]["w"].shape[3]
            upnet = torch.empty(B, h, w_, co, dtype=torch.float32, device=x4.device)
            _infer_conv(t, c4, params["stem"], (B, h, w_), 0, 0, upnet, co)                                  # :673
    else:
        upnet = batch_norm_elu(end_points["block4"], q["gamma"], q["beta"], q.get("moving_mean"), q.get("moving_var"),
                               process_group=process_group)                                                  # :672
        upnet = _slim_conv(upnet, params, "stem", activation=False, process_group=process_group)             # :673
    lvl = lambda name, x, skip, **kw: disp_bottleneck_dense(params[name], x, skip, process_group=process_group,  # noqa: E731
                                                            is_training=is_training, **kw)
    import contextlib
    with (contextlib.nullcontext() if is_training else torch.no_grad()):
        concat, _ = lvl("concat", upnet, end_points["block3"], use_skip2=False)                                  # :675
        concat1, skip1 = lvl("concat1", concat, end_points["block2"], use_skip2=False)                           # :676
        xconcat2, skip2 = lvl("concat2", concat1, end_points["block1"], skip2=skip1)                             # :677
        disp2 = disp_head(xconcat2, params["disp2"]["w"], params["disp2"]["b"])                                  # :678
        xconcat3, skip3 = lvl("concat3", xconcat2, end_points["root_block"], rates=(3, 6), skip2=skip2)          # :681
        disp3 = disp_head(xconcat3, params["disp3"]["w"], params["disp3"]["b"])                                  # :683
        xconcat4, _ = lvl("concat4", xconcat3, None, rates=(3, 6), use_skip=False, skip2=skip3)                  # :686
        disp4 = disp_head(xconcat4, params["disp4"]["w"], params["disp4"]["b"])                                  # :689
    return [disp4, disp3, disp2]


DECODER_LEVELS = (  # name, Cin, Cout, skip end-point (channels) or None, channels of skip2 or None
    ("concat", 1024, 512, ("block3", 1024), None), ("concat1", 512, 256, ("block2", 512), None),
    ("concat2", 256, 128, ("block1", 256), 512), ("concat3", 128, 64, ("root_block", 64), 256),
    ("concat4", 64, 32, None, 128))


def make_decoder_params(device, seed: int = 0, width_div: int = 1, moving_stats: bool = False) -> dict:
    """Random-init parameters of ``disp_aspp_u_dense_decoder`` in the layout it reads (He-style weights, gamma ~ 1,
    beta ~ 0).  ``width_div``: every channel count divided by it (small test decoders).  ``moving_stats``: every batch norm
    also gets ``moving_mean`` ~ 0.1 N(0,1) and ``moving_var`` ~ U(0.5, 1.5) (drawn from a generator of their own: the other
    parameters of a seed do not change) - what ``is_training=False`` reads."""
    g = torch.Generator().manual_seed(seed)
    dv = lambda c: max(32, c // width_div)  # noqa: E731

    def vec(c, scale, shift):
        return (torch.randn(c, generator=g) * scale + shift).to(device)

    def conv(k, ci, co):
        return dict(w=(torch.randn(k, k, ci, co, generator=g) / (k * ci ** 0.5)).to(device), gamma=vec(co, 0.1, 1.0),
                    beta=vec(co, 0.1, 0.0))

    p = {"stem_bn": dict(gamma=vec(dv(2048), 0.1, 1.0), beta=vec(dv(2048), 0.1, 0.0)), "stem": conv(1, dv(2048), dv(1024))}
    for name, ci, co, skip, skip2 in DECODER_LEVELS:
        ci, co = dv(ci), dv(co)
        q = dict(up=conv(1, ci, ci), sup=conv(1, ci, co), aspp_up=conv(3, ci, co), aspp_up2=conv(3, 2 * co, co))
        n_parts = 4
        if skip is not None:
            q["skip_bn"] = dict(gamma=vec(dv(skip[1]), 0.1, 1.0), beta=vec(dv(skip[1]), 0.1, 0.0))
            q["skip"] = conv(1, dv(skip[1]), co)
        else:
            q["skip"] = conv(3, 2 * co, co)
        if skip2 is not None:
            q["skip2"] = conv(1, dv(skip2), co)
            n_parts = 5
        q["out"] = conv(1, n_parts * co, co)
        p[name] = q
    for name, c in (("disp2", 128), ("disp3", 64), ("disp4", 32)):
        c = dv(c)
        p[name] = dict(w=(torch.randn(3, 3, c, 1, generator=g) / (3 * c ** 0.5)).to(device), b=vec(1, 0.1, 0.0))
    if moving_stats:
        g2 = torch.Generator().manual_seed(seed + 7919)

        def add(q):
            c = q["gamma"].shape[0]
            q["moving_mean"] = (torch.randn(c, generator=g2) * 0.1).to(device)
            q["moving_var"] = (torch.rand(c, generator=g2) + 0.5).to(device)
        for q in p.values():
            if "gamma" in q:
                add(q)
            else:
                for qq in q.values():
                    if isinstance(qq, dict) and "gamma" in qq:
                        add(qq)
    return p


def make_end_points(B: int, h: int, w: int, device, seed: int = 0, width_div: int = 1) -> dict:
    """Synthetic ResNet end-points of the shapes ``resnet_v2_50_mod`` produces for a 32h x 32w image."""
    g = torch.Generator().manual_seed(seed + 1)
    dv = lambda c: max(32, c // width_div)  # noqa: E731
    shapes = {"block4": (h, w, 2048), "block3": (2 * h, 2 * w, 1024), "block2": (4 * h, 4 * w, 512),
              "block1": (8 * h, 8 * w, 256), "root_block": (16 * h, 16 * w, 64)}
    return {k: torch.randn(B, hh, ww, dv(c), generator=g).to(device) for k, (hh, ww, c) in shapes.items()}


# ----------------------------------------------------------------------------- data parallel
# multi_scale_loss returns the GLOBAL loss on every rank (all its reduce_means divide by the global element count), so
# each rank's parameter gradient is a partial sum of the global gradient: the reduction over ranks is a SUM, never an
# average (an average would shrink every ASPP weight / gamma / beta gradient by 1/world against the single-device graph).
class GradBuckets:
    """Flat all-reduce buffers the ASPP parameter gradients are written into directly.

    ``params``: the weights / gammas / betas in the order the backward produces their gradients (last layer first).
    Registered parameters get their gradient from ``dilated_conv3x3`` / ``conv1x1`` / ``batch_norm_elu`` as a view
    into one of a few >= ``bucket_bytes`` flat buffers (no ``torch.cat`` before and no copy after the collective:
    that was ~100 MB of extra traffic per step at config 5).  ``allreduce()`` sums the buckets over the ranks in place,
    asynchronously, and returns a ``wait()``; ``grad(p)`` is the (reduced) gradient of ``p``.  A slot nobody wrote since
    the last ``allreduce()`` is zeroed first (its parameter took no part in the step), so no stale, already-reduced
    values go out again.  Parameters are registered as objects (weak references), not by address.  Call
    ``zero_grad(set_to_none=True)`` (torch's default) between steps: a ``.grad`` left in place aliases the slot and is
    then accumulated into, as torch would.
    RCCL over xGMI is a ring of point-to-point links (~150 GB/s each): a handful of >= 32 MB all-reduces runs at link
    rate where one call per filter (0.07 ... 19 MB, 11 filters) is latency-bound."""

    def __init__(self, params, bucket_bytes: int = 32 << 20):
        params = list(params)
        if not params:
            raise ValueError("no parameters")
        dev, dt = params[0].device, params[0].dtype
        self.flat, self._slots, self._written = [], {}, set()
        sizes, cur = [], 0
        layout = []
        for p in params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("one device and one dtype per GradBuckets")
            n = (p.numel() + 3) // 4 * 4  # 16-byte aligned slots (float4 stores of the weight-gradient kernels)
            if cur and (cur + n) * p.element_size() > bucket_bytes:
                sizes.append(cur)
                cur = 0
            layout.append((len(sizes), cur, tuple(p.shape), p))
            cur += n
        sizes.append(cur)
        self.flat = [torch.zeros(n, dtype=dt, device=dev) for n in sizes]
        for k, off, shape, p in layout:
            if id(p) in self._slots:
                raise ValueError("a parameter is listed twice")
            self._slots[id(p)] = (weakref.ref(p), k, off, shape, p.numel())
            _grad_sinks[id(p)] = (weakref.ref(p), self, k, off, shape)

    def grad(self, p: Tensor) -> Tensor:
        ref, k, off, shape, n = self._slots[id(p)]
        if ref() is not p:
            raise KeyError("not a parameter of this GradBuckets")
        return self.flat[k][off:off + n].view(shape)

    def release(self) -> None:
        """Unregister the parameters (their gradients are ordinary tensors again)."""
        for key in self._slots:
            slot = _grad_sinks.get(key)
            if slot is not None and slot[1] is self:
                del _grad_sinks[key]

    def allreduce(self, process_group=None):
        """Sum every bucket over the data-parallel ranks, in place, asynchronously; returns ``wait()``, which makes
        the current stream wait for the results.  One rank / no process group: a no-op."""
        import torch.distributed as dist
        for ref, k, off, shape, n in self._slots.values():
            if (k, off) not in self._written:
                self.flat[k][off:off + n].zero_()
        self._written = set()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
            return lambda: None
        handles = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=process_group, async_op=True) for f in self.flat]

        def wait():
            for h in handles:
                h.wait()
        return wait


def allreduce_weight_grads(grads, process_group=None, bucket_bytes: int = 32 << 20, average: bool = False):
    """SUM the ASPP weight / BN-parameter gradients over the data-parallel ranks in a few large buckets,
    asynchronously (for gradients that already exist as separate tensors; ``GradBuckets`` avoids the two copies made
    here).  The sum is the gradient of the global loss ``multi_scale_loss`` returns (see the note above);
    ``average=True`` divides by the world size and is only for callers whose loss is a per-rank mean.

    ``grads``: list of tensors (one device, one dtype), reduced IN PLACE.  Returns a ``wait()`` callable that blocks the
    current stream until every bucket has landed and the results have been copied back.  With no initialised process
    group (or one rank) it is a no-op.  Batch norm itself stays per-replica (each rank normalises with its own batch
    statistics): a declared deviation from the single-device graph, as SURVEY.md 8(e) allows."""
    import torch.distributed as dist
    grads = [g for g in grads if g is not None]
    if not grads or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return lambda: None
    world = dist.get_world_size(process_group)
    buckets, cur, size = [], [], 0
    for g in grads:  # in the order given: the caller lists them in the order its backward produces them
        nbytes = g.numel() * g.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group, async_op=True), flat, b))

    def wait():
        for handle, flat, b in pending:
            handle.wait()
            if average:
                flat.div_(world)
            off = 0
            for g in b:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
    return wait
