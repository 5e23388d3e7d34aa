"""The photometric loss loop of ``build_train_graph`` (SfMLearner_uni.py:75-176).

``multi_scale_loss`` mirrors the reference's loop given the same tensors:

    fused=True   one autograd.Function from the network outputs to the scalar: pyramid upsample and
                 smoothness regulariser (csrc/pyramid.hip) around one pass of the fused HIP pipeline
                 (sfm_loss_forward / sfm_loss_backward, csrc/loss.hip): geometry evaluated once per
                 (pixel, source), both warps, L1 + SSIM, masks, min-reprojection and every gradient.
    fused=False  the reference's op decomposition, each op a stand-alone HIP operator of
                 ``sfm_amd.utils`` / ``sfm_amd.pyramid`` chained by torch autograd.

Data parallelism: the batch of snippets is sharded over ranks.
Two small exchanges per step: the S*N fp64 mask sums (``proj_src_mask``, utils.py:478-479;
their global mean scales the SSIM term, SfMLearner_uni.py:144) are all-reduced on the
communication stream while the gather-side stencil runs, and the S*(3N+1) fp64 partial sums
between the two halves of the fused pipeline, because tf_ssim's global mean is taken before
its clip (utils.py:573-575).  Gradients are per-sample: no collective.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import call, ptr, stream_of
from . import pyramid as P
from . import utils as U
from .pyramid import compute_smooth_loss, resize_bilinear  # noqa: F401  (the reference keeps them beside the loss)

Tensor = torch.Tensor
ALPHA = 0.15  # SfMLearner_uni.py:84
MAX_PYR_SCALES = 8  # kMaxScales of csrc/loss_common.h: pointer table of a pyramid call


class LossEngine:
    """The two halves of the fused pipeline around the (optional) all-reduce."""

    def __init__(self, B: int, H: int, W: int, S: int, N: int, device):
        self.dims = (B, H, W, S, N)
        self.device = torch.device(device)
        n = _lib.load().sfm_loss_workspace_floats(B, H, W, S, N)
        if n <= 0:
            raise ValueError(f"unsupported fused-loss shape B={B} H={H} W={W} S={S} N={N} (needs 2 <= N <= 4)")
        self.n_partials = S * (3 * N + 1)
        with torch.cuda.device(self.device):
            self.ws = torch.empty(n, dtype=torch.float32, device=self.device)
            self.params = torch.empty(N * B, _lib.PARAM_STRIDE, dtype=torch.float32, device=self.device)
            call("sfm_loss_workspace_init", ptr(self.ws), B, H, W, S, N, stream_of(self.ws))

    def reinit(self) -> None:
        """Re-establish the workspace invariant (Ovf and the overflow flags all zero) after a failed step."""
        B, H, W, S, N = self.dims
        with torch.cuda.device(self.device):
            call("sfm_loss_workspace_init", ptr(self.ws), B, H, W, S, N, stream_of(self.ws))

    def overflow_state(self):
        """(number of raised overflow flags - one per (image, 64x32 splat tile) -, max |Ovf|): both 0 between steps
        (test / debug aid)."""
        ovf = self.region("Ovf")
        flags = self.region("flags").view(torch.int32)
        return int((flags != 0).sum()), float(ovf.abs().max())

    _REGIONS = {"Acc": 0, "Ovf": 1, "Hb": 2, "flags": 3, "msum": 4}

    def region(self, name: str) -> Tensor:
        """A view of one region of the workspace (test / debug aid; the map lives in csrc/loss_common.h)."""
        B, H, W, S, N = self.dims
        off, cnt = ctypes.c_longlong(0), ctypes.c_longlong(0)
        call("sfm_loss_workspace_region", B, H, W, S, N, self._REGIONS[name], ctypes.byref(off), ctypes.byref(cnt))
        return self.ws[off.value:off.value + cnt.value]

    @staticmethod
    def _pyr_args(depth, trans):
        """host arrays of a pyramid call: depth[s] [B,h,w] or [B,h,w,1], trans[s] [B,h,w,3N] at their native sizes"""
        return (P._ptrs(depth), P._ptrs(trans), P._ints([t.shape[1] for t in trans]), P._ints([t.shape[2] for t in trans]))

    def forward_partials(self, tgt, src, K, depth, pose, trans, n_elem_global: int, need_grad: bool = True,
                         out: Optional[Tensor] = None, process_group=None, dp: Optional[bool] = None,
                         msum_allreduce=None) -> Tensor:
        """First half: both warps, the L1/SSIM stencil and the local partial sums.  ``n_elem_global`` =
        B_global*H*W*3 (the reduce_mean denominators are global under data parallelism).
        ``depth`` / ``trans``: the maps resized to HxW and stacked ([S,B,H,W], [S,B,H,W,3N]), or LISTS of the
        pyramid's maps at their native sizes - the kernels then resize while loading (``sfm_loss_forward_pyr``).
        With more than one rank in ``process_group`` (or in the default group when torch.distributed is initialised and
        ``n_elem_global`` says the batch is larger than this rank's) the forward runs as three calls and the mask sums
        are all-reduced beside the gather-side stencil (``forward_warps`` / ``forward_gather`` / ``forward_splat``).
        ``dp``: force that sequence on (rehearsals with one rank) or off; ``msum_allreduce``: a callable that sums the
        mask-sum tensor over the ranks in place and returns a work handle or None, in place of ``dist.all_reduce(...,
        async_op=True)`` (rehearsal backends that cannot reduce device tensors)."""
        B, H, W, S, N = self.dims
        partials = out if out is not None else torch.empty(self.n_partials, dtype=torch.float64, device=self.device)
        if dp if dp is not None else _world_size(process_group) > 1:
            import torch.distributed as dist
            msum = self.forward_warps(tgt, src, K, depth, pose, trans)
            if msum_allreduce is not None:
                work = msum_allreduce(msum)
            else:
                work = _all_reduce_sum(msum, process_group, async_op=True)
            self.forward_gather(tgt, n_elem_global, need_grad)
            if work is not None:
                work.wait()  # device tensors: the current stream waits for the collective, not the host
            self.forward_splat(src, n_elem_global, need_grad, msum, partials)
            self._guarded = False  # the sums are not global yet: the backward applies the seed guard after the all-reduce
            return partials
        self._guarded = True       # sfm_loss_forward's own reduction applies it (single device: its sums are the global ones)
        st = stream_of(tgt)
        call("sfm_pose_setup", ptr(pose), 6 * N, 6, ptr(K), B, N, 0, ptr(self.params), st)
        if isinstance(depth, (list, tuple)):
            call("sfm_loss_forward_pyr", ptr(tgt), ptr(src), *self._pyr_args(depth, trans), ptr(self.params), ptr(self.ws),
                 B, H, W, S, N, int(n_elem_global), int(need_grad), ptr(partials), st)
        else:
            call("sfm_loss_forward", ptr(tgt), ptr(src), ptr(depth), ptr(trans), ptr(self.params), ptr(self.ws), B, H, W, S,
                 N, int(n_elem_global), int(need_grad), ptr(partials), st)
        return partials

    # the three calls of the data-parallel forward, one method each (the CPU test of the host sequence replaces them by
    # the oracle: tests/test_dp_gloo.py)
    def forward_warps(self, tgt, src, K, depth, pose, trans) -> Tensor:
        """both warps; returns this rank's mask sums [S*N] (fp64), to be summed over the ranks"""
        B, H, W, S, N = self.dims
        st = stream_of(tgt)
        msum = torch.empty(S * N, dtype=torch.float64, device=self.device)
        call("sfm_pose_setup", ptr(pose), 6 * N, 6, ptr(K), B, N, 0, ptr(self.params), st)
        if isinstance(depth, (list, tuple)):
            call("sfm_loss_forward_warps_pyr", ptr(tgt), ptr(src), *self._pyr_args(depth, trans), ptr(self.params),
                 ptr(self.ws), B, H, W, S, N, ptr(msum), st)
        else:
            call("sfm_loss_forward_warps", ptr(tgt), ptr(src), ptr(depth), ptr(trans), ptr(self.params), ptr(self.ws),
                 B, H, W, S, N, ptr(msum), st)
        return msum

    def forward_gather(self, tgt, n_elem_global: int, need_grad: bool) -> None:
        B, H, W, S, N = self.dims
        call("sfm_loss_forward_gather", ptr(tgt), ptr(self.ws), B, H, W, S, N, int(n_elem_global), int(need_grad),
             stream_of(tgt))

    def forward_splat(self, src, n_elem_global: int, need_grad: bool, msum_global: Tensor, partials: Tensor) -> None:
        B, H, W, S, N = self.dims
        call("sfm_loss_forward_splat", ptr(src), ptr(self.ws), B, H, W, S, N, int(n_elem_global), int(need_grad),
             ptr(msum_global), ptr(partials), stream_of(src))

    def backward(self, tgt, src, K, depth, pose, trans, partials, n_elem_global: int, gloss: float = 1.0,
                 need_grad: bool = True, d_trans_into: Optional[Sequence[Tensor]] = None, n_extra: int = 0,
                 seeds_guarded: Optional[bool] = None):
        """Second half.  Returns (loss, d_depth, d_pose, d_trans).  Stacked inputs: d_depth [S,B,H,W], d_trans
        [S,B,H,W,3N].  Pyramid LISTS for ``depth`` / ``trans``: the adjoint of the resize runs in the kernel and the
        gradients come back as lists at the pyramid's own sizes (``sfm_loss_backward_pyr``); ``d_trans_into``: maps
        that already hold another term of the translation gradients (the regulariser's), added to in place;
        ``n_extra``: fp64 values behind the sums in ``partials`` that are added to the loss (pyramid path only).
        ``seeds_guarded``: whether the forward has already applied the seed guard for these very sums (what the
        single-call forward of this engine does; default: what the last ``forward_partials`` did) - pass False when
        ``partials`` were changed since (summed with other ranks', edited by a test)."""
        B, H, W, S, N = self.dims
        guarded = getattr(self, "_guarded", False) if seeds_guarded is None else seeds_guarded
        ng = (3 if guarded else 1) if need_grad else 0  # include/sfm_amd.h: bit 1 = guard already applied
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        d_pose = torch.empty_like(pose) if need_grad else None
        if isinstance(depth, (list, tuple)):
            d_depth = d_trans = None
            dd_tab = dt_tab = None
            if need_grad:
                # one zero-filled buffer for every level of d_depth (one fill launch), views per level
                sizes = [t.shape[1] * t.shape[2] for t in trans]
                flat = torch.zeros(B * sum(sizes), dtype=torch.float32, device=self.device)
                d_depth, off = [], 0
                for t, n in zip(trans, sizes):
                    d_depth.append(flat[off:off + B * n].view(B, t.shape[1], t.shape[2], 1))
                    off += B * n
                d_trans = list(d_trans_into) if d_trans_into is not None else [torch.zeros_like(t) for t in trans]
                if any(g.shape != t.shape or not g.is_contiguous() for g, t in zip(d_trans, trans)):
                    raise ValueError("d_trans_into: one contiguous map per level, shaped like the level")
                dd_tab, dt_tab = P._ptrs(d_depth), P._ptrs(d_trans)
            call("sfm_loss_backward_pyr", ptr(tgt), ptr(src), *self._pyr_args(depth, trans), ptr(self.params), ptr(self.ws),
                 B, H, W, S, N, int(n_elem_global), float(gloss), ng, ptr(partials), int(n_extra), ptr(pose), ptr(K),
                 ptr(loss), dd_tab, dt_tab, ptr(d_pose), stream_of(tgt))
            return loss, d_depth, d_pose, d_trans
        d_depth = torch.empty(S, B, H, W, dtype=torch.float32, device=self.device) if need_grad else None
        d_trans = torch.empty(S, B, H, W, 3 * N, dtype=torch.float32, device=self.device) if need_grad else None
        call("sfm_loss_backward", ptr(tgt), ptr(src), ptr(depth), ptr(trans), ptr(self.params), ptr(self.ws), B, H, W,
             S, N, int(n_elem_global), float(gloss), ng, ptr(partials), ptr(pose), ptr(K), ptr(loss),
             ptr(d_depth), ptr(d_trans), ptr(d_pose), stream_of(tgt))
        return loss, d_depth, d_pose, d_trans


_engines = {}


def _world_size(process_group=None) -> int:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(process_group)


def get_engine(B, H, W, S, N, device) -> LossEngine:
    """The cached workspace of a shape, per device AND per stream: the C ABI wants calls that share a workspace
    stream-ordered (include/sfm_amd.h), so work enqueued on two streams (train / val, an eager call beside a graph
    capture) never shares one."""
    device = torch.device(device)
    key = (B, H, W, S, N, str(device), torch.cuda.current_stream(device).cuda_stream)
    eng = _engines.get(key)
    if eng is None:
        eng = _engines[key] = LossEngine(B, H, W, S, N, device)
    return eng


def _evict_engine(eng: LossEngine) -> None:
    """After a failed call between sfm_loss_forward and sfm_loss_backward the overflow accumulator and its flags may be
    dirty (the invariant of sfm_loss_workspace_init): drop the engine, the next step builds a clean one."""
    for k, v in list(_engines.items()):
        if v is eng:
            del _engines[k]


def global_batch(B_local: int, process_group=None, device=None) -> int:
    """Total number of snippets over all ranks (each rank may hold a different count): ONE all-reduce and one host
    synchronisation (``.item()``), every time it is called - nothing is cached, because a rank that skipped the
    exchange while another made it would pair that all-reduce with the wrong collective.  Call it where every rank
    takes part (once per sharding, not per step) and pass the result as ``global_batch_size``.
    ``device``: where the exchanged scalar lives for an RCCL group - the device of the caller's tensors, not the
    process's current device."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return B_local
    if dist.get_world_size(process_group) == 1:
        return B_local
    if dist.get_backend(process_group) == "nccl":
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        t = torch.tensor([B_local], dtype=torch.int64, device=dev)
    else:
        t = torch.tensor([B_local], dtype=torch.int64)
    dist.all_reduce(t, group=process_group)
    return int(t.item())


def _resolve_global_batch(B_local: int, global_batch_size: Optional[int], process_group) -> int:
    if global_batch_size is not None:
        return int(global_batch_size)
    if _world_size(process_group) > 1:
        raise ValueError("data parallel (more than one rank): pass global_batch_size - the number of snippets over all "
                         "ranks, e.g. sfm_amd.loss.global_batch(B, group) evaluated once by every rank")
    return B_local


def allreduce_partials(partials: Tensor, process_group=None) -> Tensor:
    """Sum the fp64 partial sums over the data-parallel ranks (RCCL over xGMI for device
    tensors; gloo in the CPU tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        _all_reduce_sum(partials, process_group)
    return partials


def _all_reduce_sum(t: Tensor, process_group=None, async_op: bool = False):
    """Sum ``t`` over the ranks in place.  RCCL (backend "nccl") reduces the device tensor itself, asynchronously if
    asked (returns the work handle); any other backend with a device tensor - gloo rehearsals of the multi-rank path on
    one GPU - stages through host memory, synchronously (returns None)."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(process_group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=process_group)
        t.copy_(h)
        return None
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op)
    return work if async_op else None


_SIDE_STREAM = os.environ.get("SFM_LOSS_SIDE_STREAM", "1") != "0"
_side_streams = {}


def _side_stream(device):
    """one extra stream per device for work that runs beside the fused step (the regulariser)"""
    key = torch.device(device).index
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class _FusedTotalLoss(torch.autograd.Function):
    """total_loss(pred_poses [B,N,6], pred_depth[0..S), pred_exp_logits[0..S), pred_disp[0..S) or nothing).

    Everything between the network outputs and the scalar runs here as hand-written kernels, forward and
    backward in one go: pyramid upsample (SfMLearner_uni.py:91,113), the fused warp + photometric pipeline
    (:120-168), the smoothness / |t| regulariser (:94-101), and the adjoints back to the pyramid.  The
    gradients are computed with the value and scaled by the upstream scalar in ``backward``."""

    @staticmethod
    def forward(ctx, tgt, src, K, process_group, global_b, smooth_weight, explain_reg_weight, n_scales, engine,
                fuse_resize, poses, *maps):
        S = n_scales
        pred_depth, pred_exp = list(maps[:S]), list(maps[S:2 * S])
        pred_disp = list(maps[2 * S:3 * S])
        B, H, W, _ = tgt.shape
        N = poses.shape[1]
        need = any(ctx.needs_input_grad[10:])
        with torch.cuda.device(tgt.device):
            eng = engine if engine is not None else get_engine(B, H, W, S, N, tgt.device)
            if eng.dims != (B, H, W, S, N) or eng.device != tgt.device:
                raise ValueError(f"engine was built for {eng.dims} on {eng.device}, got {(B, H, W, S, N)} on {tgt.device}")
            # The pyramid kernels take ONE size per level (the translation map's) for both of its maps; the reference
            # resizes each map on its own (SfMLearner_uni.py:91, 113), so levels whose two maps differ in size are legal
            # there: those go through the materialised resize.  Anything else that does not fit raises here - the
            # kernels index with these sizes and check nothing themselves.
            for s_, (dm, tm) in enumerate(zip(pred_depth, pred_exp)):
                if dm.shape[0] != B or tm.shape[0] != B or dm.shape[3] != 1 or tm.shape[3] != 3 * N:
                    raise ValueError(f"scale {s_}: pred_depth {tuple(dm.shape)} / pred_exp_logits {tuple(tm.shape)} do not "
                                     f"match [B={B},h,w,1] / [B,h,w,3N={3 * N}]")
                if dm.device != tgt.device or tm.device != tgt.device:
                    raise ValueError(f"scale {s_}: pyramid maps on {dm.device} / {tm.device}, images on {tgt.device}")
            same = all(dm.shape[1:3] == tm.shape[1:3] and tm.shape[1] <= H and tm.shape[2] <= W
                       for dm, tm in zip(pred_depth, pred_exp))
            fused_maps = fuse_resize and S <= MAX_PYR_SCALES and same
            if fused_maps:
                # the kernels read the pyramid at its native sizes and resize while loading (SfMLearner_uni.py:91, 113
                # inside the warps) and scatter the adjoint of the resize themselves: no [S,B,H,W,1+3N] intermediate
                # in either direction
                depth, trans = pred_depth, pred_exp
            else:
                depth = P.pyramid_upsample(pred_depth, H, W).view(S, B, H, W)
                trans = P.pyramid_upsample(pred_exp, H, W)
            n_elem = global_b * H * W * 3
            g_smooth, sm = None, None
            buf = None
            if fused_maps:
                # ONE fp64 buffer = ONE all-reduce: the S*(3N+1) partial sums, then this rank's share of the regulariser,
                # written there by its kernel and added to the loss by the backward kernel (no torch arithmetic at all)
                buf = torch.empty(eng.n_partials + (1 if smooth_weight > 0 else 0), dtype=torch.float64, device=tgt.device)
            side = None
            if smooth_weight > 0:
                ws_, wa_ = P.smooth_weights(S, smooth_weight, explain_reg_weight)
                if fused_maps and _SIDE_STREAM and torch.cuda.is_current_stream_capturing():
                    # the regulariser (value and gradient: ~40 us of small kernels at c2) on a second stream beside the
                    # forward kernels; joined before the first reader of its outputs (the all-reduce / the backward).
                    # Only while a HIP graph is being captured (GraphedTotalLoss: a parallel branch, 0.384 -> 0.374 ms per
                    # replay); the eager sequence is bound by the host, which the fork and join cost 19 us more
                    main = torch.cuda.current_stream(tgt.device)
                    side = _side_stream(tgt.device)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        sm, g_smooth = P.smooth_terms(pred_disp + pred_exp, ws_, wa_, global_b, need,
                                                      out_f64=buf[eng.n_partials:])
                else:
                    sm, g_smooth = P.smooth_terms(pred_disp + pred_exp, ws_, wa_, global_b, need,
                                                  out_f64=buf[eng.n_partials:] if buf is not None else None)
            try:
                if fused_maps:
                    eng.forward_partials(tgt, src, K, depth, poses, trans, n_elem, need, out=buf, process_group=process_group)
                    if side is not None:
                        torch.cuda.current_stream(tgt.device).wait_stream(side)
                        side = None
                    if global_b != B or process_group is not None:
                        allreduce_partials(buf, process_group)
                    d_into = g_smooth[S:] if g_smooth is not None else None
                    total, d_depth, d_pose, d_trans = eng.backward(tgt, src, K, depth, poses, trans, buf, n_elem, 1.0, need,
                                                                   d_trans_into=d_into, n_extra=buf.numel() - eng.n_partials)
                else:
                    # one buffer, ONE all-reduce: the S*(3N+1) partial sums and this rank's share of the regulariser
                    buf = torch.zeros(eng.n_partials + 1, dtype=torch.float64, device=tgt.device)
                    if sm is not None:
                        buf[-1] = sm
                    eng.forward_partials(tgt, src, K, depth, poses, trans, n_elem, need, out=buf[:-1],
                                         process_group=process_group)
                    if global_b != B or process_group is not None:
                        allreduce_partials(buf, process_group)
                    loss, d_depth, d_pose, d_trans = eng.backward(tgt, src, K, depth, poses, trans, buf[:-1], n_elem, 1.0, need)
                    total = loss if sm is None else loss + buf[-1].float()
            except Exception:
                # a failure between the two halves (rejected argument, failed collective) can leave the splat's overflow
                # accumulator and flags raised: restore the invariant and do not reuse this workspace
                if side is not None:
                    torch.cuda.current_stream(tgt.device).wait_stream(side)
                eng.reinit()
                _evict_engine(eng)
                raise
            if need and fused_maps:
                g_disp = g_smooth[:S] if g_smooth is not None else []
                # (held as a plain attribute and handed over by backward(): tensors returned from save_for_backward
                # are still referenced by the graph, so autograd would CLONE each into .grad - a second pass over
                # every gradient; handed over with no other owner they become the leaves' .grad as they are)
                ctx.owned_grads = [d_pose, *d_depth, *d_trans, *g_disp]
            elif need:
                g_depth = P.pyramid_upsample_grad(d_depth.view(S, B, H, W, 1), [(m.shape[1], m.shape[2]) for m in pred_depth])
                # the regulariser's gradient of the translation maps is already in place: the adjoint adds to it
                g_exp = P.pyramid_upsample_grad(d_trans, [(m.shape[1], m.shape[2]) for m in pred_exp],
                                                accumulate_into=g_smooth[S:] if g_smooth is not None else None)
                g_disp = g_smooth[:S] if g_smooth is not None else []
                ctx.save_for_backward(d_pose, *g_depth, *g_exp, *g_disp)
        return total

    @staticmethod
    def backward(ctx, g):
        if hasattr(ctx, "owned_grads"):
            grads = ctx.owned_grads
            if grads is None:
                raise RuntimeError("multi_scale_loss(fused=True, fuse_resize=True) hands its gradients over once: a second "
                                   "backward through the same graph needs fuse_resize=False")
            ctx.owned_grads = None
            # multiply in place by the upstream scalar (sfm_scale_grads): a launch that exits at once for loss.backward()
            # (g == 1) instead of a pass over every gradient
            g = g.to(dtype=torch.float32, device=grads[0].device).reshape(()).contiguous()
            with torch.cuda.device(g.device):
                n = (ctypes.c_longlong * len(grads))(*[t.numel() for t in grads])
                call("sfm_scale_grads", P._ptrs(grads), n, len(grads), ptr(g), stream_of(g))
            return (None,) * 10 + tuple(grads)
        return (None,) * 10 + tuple(torch._foreach_mul(list(ctx.saved_tensors), g))


def multi_scale_loss(tgt_image: Tensor, src_image_stack: Tensor, intrinsics: Tensor,
                     pred_depth: Sequence[Tensor], pred_poses: Tensor, pred_exp_logits: Sequence[Tensor],
                     pred_disp: Optional[Sequence[Tensor]] = None, smooth_weight: float = 0.0,
                     explain_reg_weight: float = 0.01, fused: bool = True, process_group=None,
                     global_batch_size: Optional[int] = None,
                     engine: Optional[LossEngine] = None, fuse_resize: bool = True) -> Tensor:
    """total_loss of SfMLearner_uni.py:75-176 (exp_loss is identically 0 there).

    tgt_image [B,H,W,3], src_image_stack [B,H,W,3N] in [-1,1]; intrinsics [B,3,3]
    (``intrinsics[:,0]``, used for every scale, :125); pred_depth[s] [B,h,w,1];
    pred_poses [B,N,6]; pred_exp_logits[s] [B,h,w,3N]; pred_disp[s] for smoothness.
    ``len(pred_depth)`` scales (quirk 1).  Under data parallelism pass the process
    group (or have torch.distributed initialised) and the loss is the global one: per step one all-reduce of S*N
    mask sums hidden under the gather-side stencil and one stream-ordered all-reduce of the partial sums.
    ``global_batch_size``: the number of snippets over all ranks - REQUIRED when the group has more than one rank
    (``global_batch(B, group)`` exchanges it: call it once where every rank is known to take part, not per step: a
    per-step exchange that some ranks skip would pair with another rank's all-reduce of the sums).  ``engine``: a ``LossEngine`` to
    run on instead of the cached one of this shape, device and stream (``GraphedTotalLoss`` owns its own).
    ``fuse_resize`` (fused only): the warp kernels read ``pred_depth`` / ``pred_exp_logits`` at their native sizes and
    evaluate the ``resize_bilinear`` of :91, 113 while loading; False materialises the resized maps first
    (``sfm_pyramid_upsample_fwd``) - same numbers bit for bit."""
    if explain_reg_weight <= 0:
        raise NameError("curr_exp_logits is undefined when explain_reg_weight == 0 (SfMLearner_uni.py:127)")
    tgt = U._check(tgt_image, "tgt_image", 4, 3)
    B, H, W, _ = tgt.shape
    poses = U._check(pred_poses, "pred_poses", 3, 6)
    N = poses.shape[1]
    src = U._check(src_image_stack, "src_image_stack", 4, 3 * N)
    K = U._check(intrinsics, "intrinsics", 3, 3)
    S = len(pred_depth)
    if N < 2:
        raise IndexError("the min-reprojection term reads curr_errors[1] (SfMLearner_uni.py:168): need >= 2 sources")
    if smooth_weight > 0 and (pred_disp is None or len(pred_disp) != S):
        raise ValueError("smooth_weight > 0 needs pred_disp, one map per scale (SfMLearner_uni.py:96)")
    if len(pred_exp_logits) != S:
        raise ValueError("one pred_exp_logits map per scale")
    if fused:
        gb = _resolve_global_batch(B, global_batch_size, process_group)
        maps = list(pred_depth) + list(pred_exp_logits) + (list(pred_disp) if smooth_weight > 0 else [])
        maps = [U._check(m, "pyramid map", 4) for m in maps]
        return _FusedTotalLoss.apply(tgt, src, K, process_group, gb, float(smooth_weight), float(explain_reg_weight), S,
                                     engine, bool(fuse_resize), poses, *maps)
    smooth = tgt.new_zeros(())
    if smooth_weight > 0:
        for s in range(S):
            smooth = smooth + smooth_weight / (2 ** s) * compute_smooth_loss(pred_disp[s])
            smooth = smooth + explain_reg_weight / (2 ** s) * (
                compute_smooth_loss(pred_exp_logits[s]) + pred_exp_logits[s].abs().mean())
    depth_full = [resize_bilinear(pred_depth[s], H, W).squeeze(3) for s in range(S)]
    trans_full = [resize_bilinear(pred_exp_logits[s], H, W) for s in range(S)]
    pixel = tgt.new_zeros(())
    for s in range(S):
        errs = []
        for i in range(N):
            t_i = trans_full[s][..., 3 * i:3 * i + 3]
            src_i = src[..., 3 * i:3 * i + 3]
            proj, _ = U.projective_inverse_warp(src_i, depth_full[s], poses[:, i, :], K, add_trans=True, translations=t_i)
            psrc, pmask = U.projective_warp(tgt, depth_full[s], poses[:, i, :], K, add_trans=True, translations=t_i)
            err = ALPHA * (proj - tgt).abs() + (1 - ALPHA) * U.tf_ssim(proj, tgt, size=3, mean_metric=False)
            errs.append(err)
            perr = pmask * (psrc - src_i).abs()
            pssim = pmask * U.tf_ssim(psrc, src_i, size=3, mean_metric=True)
            pixel = pixel + ALPHA * perr.mean() + (1 - ALPHA) * pssim.mean()
        pixel = pixel + 2.0 * torch.where(errs[0] <= errs[1], errs[0], errs[1]).mean()
    return pixel + smooth


class GraphedTotalLoss:
    """``total_loss`` forward + backward as ONE captured HIP graph (the launch-bound case: at 416x128 B=8 the ~25
    kernels of normalise -> upsample -> fused step -> regulariser -> adjoints take 0.43 ms on the GPU but 0.55 ms to
    enqueue from Python; a graph launch takes the host out of the loop).

    Shapes are fixed at construction.  ``__call__(raw_disp, poses, exp_logits)`` copies the network outputs into the
    graph's static inputs, replays, and returns ``(loss, d_raw_disp[s], d_poses, d_exp_logits[s])`` - views of the
    graph's static outputs, overwritten by the next call.  Single GPU or one stream-ordered all-reduce (the process
    group's collectives are captured with the kernels)."""

    def __init__(self, tgt_image: Tensor, src_image_stack: Tensor, intrinsics: Tensor, raw_disp: Sequence[Tensor],
                 pred_poses: Tensor, pred_exp_logits: Sequence[Tensor], smooth_weight: float = 0.0,
                 explain_reg_weight: float = 0.01, process_group=None, global_batch_size: Optional[int] = None,
                 warmup: int = 3):
        dev = tgt_image.device
        if dev.type != "cuda":
            raise TypeError("GraphedTotalLoss needs tensors on a ROCm device (sfm_amd has no CPU path)")
        self.tgt, self.src, self.K = tgt_image.contiguous(), src_image_stack.contiguous(), intrinsics.contiguous()
        self.raw = [t.detach().clone().requires_grad_(True) for t in raw_disp]
        self.exp = [t.detach().clone().requires_grad_(True) for t in pred_exp_logits]
        self.poses = pred_poses.detach().clone().requires_grad_(True)
        gb = _resolve_global_batch(self.tgt.shape[0], global_batch_size, process_group)
        leaves = self.raw + self.exp + [self.poses]
        B, H, W, _ = self.tgt.shape
        # its own workspace: replays must never share Ovf / flags / params with eager calls or other graphs
        self.engine = LossEngine(B, H, W, len(self.raw), self.poses.shape[1], dev)

        def run():
            dn, dp = P.normalize_disp(self.raw)
            loss = multi_scale_loss(self.tgt, self.src, self.K, dp, self.poses, self.exp, pred_disp=dn,
                                    smooth_weight=smooth_weight, explain_reg_weight=explain_reg_weight,
                                    process_group=process_group, global_batch_size=gb, engine=self.engine)
            grads = torch.autograd.grad(loss, leaves)
            return loss, grads

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # allocations and lazy initialisation (workspace, engine) happen before capture
            for _ in range(max(1, warmup)):
                run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.grads = run()

    def __call__(self, raw_disp: Sequence[Tensor], pred_poses: Tensor, pred_exp_logits: Sequence[Tensor]):
        with torch.no_grad():
            torch._foreach_copy_(self.raw + self.exp + [self.poses], list(raw_disp) + list(pred_exp_logits) + [pred_poses])
        self.graph.replay()
        S = len(self.raw)
        return self.loss, list(self.grads[:S]), self.grads[2 * S], list(self.grads[S:2 * S])
