"""The photometric loss loop of ``build_train_graph`` (SfMLearner_uni.py:75-176).

``multi_scale_loss`` mirrors the reference's loop given the same tensors:

    fused=True   one pass of the fused HIP pipeline (sfm_loss_forward / sfm_loss_backward,
                 csrc/loss.hip): geometry evaluated once per (pixel, source), both warps,
                 L1 + SSIM, masks, min-reprojection and every gradient in 8 launches (7 kernels + pose set-up).
    fused=False  the reference's op decomposition, each op a stand-alone HIP operator of
                 ``sfm_amd.utils`` glued by torch device ops and torch autograd.

Data parallelism: the batch of snippets is sharded over ranks;
the only exchange is one all-reduce of S*(3N+1) fp64 partial sums between the two
halves of the fused pipeline (``process_group``), because tf_ssim's global mean is
taken before its clip (utils.py:573-575).  Gradients are per-sample: no collective.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import call, ptr, stream_of
from . import utils as U

Tensor = torch.Tensor
ALPHA = 0.15  # SfMLearner_uni.py:84


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

    def overflow_state(self):
        """(number of raised overflow flags, max |Ovf|) - both 0 between steps (test / debug aid)."""
        B, H, W, S, N = self.dims
        n_acc = S * N * B * H * W * 4
        ovf = self.ws[n_acc:2 * n_acc]                       # workspace map of csrc/loss.hip: Acc, Ovf, ...
        flags = self.ws[-((S * N * B + 3) // 4 * 4):].view(torch.int32)[:S * N * B]
        return int((flags != 0).sum()), float(ovf.abs().max())

    def forward_partials(self, tgt, src, K, depth, pose, trans, n_elem_global: int, need_grad: bool = True) -> Tensor:
        """First half: both warps, the L1/SSIM stencil and the local partial sums.  ``n_elem_global`` =
        B_global*H*W*3 (the reduce_mean denominators are global under data parallelism)."""
        B, H, W, S, N = self.dims
        partials = torch.empty(self.n_partials, dtype=torch.float64, device=self.device)
        st = stream_of(tgt)
        call("sfm_pose_setup", ptr(pose), 6 * N, 6, ptr(K), B, N, 0, ptr(self.params), st)
        call("sfm_loss_forward", ptr(tgt), ptr(src), ptr(depth), ptr(trans), ptr(self.params), ptr(self.ws), B, H, W, S,
             N, int(n_elem_global), int(need_grad), ptr(partials), st)
        return partials

    def backward(self, tgt, src, K, depth, pose, trans, partials, n_elem_global: int, gloss: float = 1.0,
                 need_grad: bool = True):
        B, H, W, S, N = self.dims
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        d_depth = torch.empty_like(depth) if need_grad else None
        d_trans = torch.empty_like(trans) if need_grad else None
        d_pose = torch.empty_like(pose) if need_grad else None
        call("sfm_loss_backward", ptr(tgt), ptr(src), ptr(depth), ptr(trans), ptr(self.params), ptr(self.ws), B, H, W,
             S, N, int(n_elem_global), float(gloss), int(need_grad), ptr(partials), ptr(pose), ptr(K), ptr(loss),
             ptr(d_depth), ptr(d_trans), ptr(d_pose), stream_of(tgt))
        return loss, d_depth, d_pose, d_trans


_engines = {}


def get_engine(B, H, W, S, N, device) -> LossEngine:
    key = (B, H, W, S, N, str(device))
    eng = _engines.get(key)
    if eng is None:
        eng = _engines[key] = LossEngine(B, H, W, S, N, device)
    return eng


def global_batch(B_local: int, process_group=None) -> int:
    """Total number of snippets over all ranks (each rank may hold a different count)."""
    import torch.distributed as dist
    if process_group is None and not (dist.is_available() and dist.is_initialized()):
        return B_local
    t = torch.tensor([B_local], dtype=torch.int64)
    backend = dist.get_backend(process_group)
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=process_group)
    return int(t.item())


def allreduce_partials(partials: Tensor, process_group=None) -> Tensor:
    """Sum the fp64 partial sums over the data-parallel ranks (RCCL over xGMI for device
    tensors; gloo in the CPU tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=process_group)
    return partials


class _FusedPixelLoss(torch.autograd.Function):
    """pixel_loss(depth [S,B,H,W], pose [B,N,6], trans [S,B,H,W,3N]); gradients are
    computed in the forward (one fused pass) and scaled by the upstream scalar."""

    @staticmethod
    def forward(ctx, depth, pose, trans, tgt, src, K, process_group, global_b):
        S, B, H, W = depth.shape
        N = pose.shape[1]
        with torch.cuda.device(tgt.device):
            eng = get_engine(B, H, W, S, N, tgt.device)
            need = any(ctx.needs_input_grad[:3])
            partials = eng.forward_partials(tgt, src, K, depth, pose, trans, global_b * H * W * 3, need)
            if global_b != B or process_group is not None:
                partials = allreduce_partials(partials, process_group)
            loss, d_depth, d_pose, d_trans = eng.backward(tgt, src, K, depth, pose, trans, partials,
                                                          global_b * H * W * 3, 1.0, need)
        if need:
            ctx.save_for_backward(d_depth, d_pose, d_trans)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_depth, d_pose, d_trans = ctx.saved_tensors
        return d_depth * g, d_pose * g, d_trans * g, None, None, None, None, None


def resize_bilinear(x_nhwc: Tensor, out_h: int, out_w: int) -> Tensor:
    """``tf.image.resize_bilinear`` with TF-1 defaults (align_corners=False, no half-pixel
    centres; SfMLearner_uni.py:91,113).  torch device glue, differentiable."""
    _, in_h, in_w, _ = x_nhwc.shape
    if (in_h, in_w) == (out_h, out_w):
        return x_nhwc
    dev, dt = x_nhwc.device, x_nhwc.dtype

    def axis(in_n, out_n):
        scale = torch.tensor(in_n / out_n, dtype=torch.float32, device=dev)
        s = torch.arange(out_n, dtype=torch.float32, device=dev) * scale
        lo = torch.floor(s).long()
        hi = torch.clamp(lo + 1, max=in_n - 1)
        return lo, hi, (s - lo.float()).to(dt)

    ylo, yhi, yl = axis(in_h, out_h)
    xlo, xhi, xl = axis(in_w, out_w)
    top, bot = x_nhwc[:, ylo], x_nhwc[:, yhi]
    xl_, yl_ = xl.reshape(1, 1, out_w, 1), yl.reshape(1, out_h, 1, 1)
    top_i = top[:, :, xlo] + (top[:, :, xhi] - top[:, :, xlo]) * xl_
    bot_i = bot[:, :, xlo] + (bot[:, :, xhi] - bot[:, :, xlo]) * xl_
    return top_i + (bot_i - top_i) * yl_


def compute_smooth_loss(pred: Tensor) -> Tensor:
    """SfMLearner_uni.py:205-219 (torch device glue)."""
    def gradient(p):
        return p[:, :, 1:, :] - p[:, :, :-1, :], p[:, 1:, :, :] - p[:, :-1, :, :]
    beta = 0.25
    dx, dy = gradient(pred)
    dx2, dxdy = gradient(dx)
    dydx, dy2 = gradient(dy)
    gx = dx.abs().mean() + dy.abs().mean()
    hx = dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()
    return beta * gx + (1 - beta) * hx


def multi_scale_loss(tgt_image: Tensor, src_image_stack: Tensor, intrinsics: Tensor,
                     pred_depth: Sequence[Tensor], pred_poses: Tensor, pred_exp_logits: Sequence[Tensor],
                     pred_disp: Optional[Sequence[Tensor]] = None, smooth_weight: float = 0.0,
                     explain_reg_weight: float = 0.01, fused: bool = True, process_group=None,
                     global_batch_size: Optional[int] = None) -> Tensor:
    """total_loss of SfMLearner_uni.py:75-176 (exp_loss is identically 0 there).

    tgt_image [B,H,W,3], src_image_stack [B,H,W,3N] in [-1,1]; intrinsics [B,3,3]
    (``intrinsics[:,0]``, used for every scale, :125); pred_depth[s] [B,h,w,1];
    pred_poses [B,N,6]; pred_exp_logits[s] [B,h,w,3N]; pred_disp[s] for smoothness.
    ``len(pred_depth)`` scales (quirk 1).  Under data parallelism pass the process
    group (or have torch.distributed initialised) and the loss is the global one."""
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
    smooth = tgt.new_zeros(())
    if smooth_weight > 0:
        for s in range(S):
            smooth = smooth + smooth_weight / (2 ** s) * compute_smooth_loss(pred_disp[s])
            smooth = smooth + explain_reg_weight / (2 ** s) * (
                compute_smooth_loss(pred_exp_logits[s]) + pred_exp_logits[s].abs().mean())
    depth_full = [resize_bilinear(pred_depth[s], H, W).squeeze(3) for s in range(S)]
    trans_full = [resize_bilinear(pred_exp_logits[s], H, W) for s in range(S)]
    if fused:
        gb = global_batch_size if global_batch_size is not None else global_batch(B, process_group)
        depth = torch.stack(depth_full, 0).contiguous()
        trans = torch.stack(trans_full, 0).contiguous()
        pixel = _FusedPixelLoss.apply(depth, poses, trans, tgt, src, K, process_group, gb)
        return pixel + smooth
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
