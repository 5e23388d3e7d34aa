"""Drop-in for the view-synthesis functions of the reference's ``utils.py``.

Same names, argument order, tensor conventions (NHWC fp32) and return tuples as the
reference (paths relative to the reference repo):

    projective_inverse_warp  utils.py:207-239      bilinear_sampler  utils.py:241-340
    projective_warp          utils.py:172-205      bilinear_project  utils.py:342-502
    tf_ssim                  utils.py:548-575      pose_vec2mat / euler2mat / meshgrid /
                                                   pixel2cam / cam2pixel  utils.py:27-170

Every function runs hand-written gfx950 kernels through the C ABI of
``libsfm_amd.so`` (include/sfm_amd.h) and is differentiable (``torch.autograd``)
with respect to depth, pose, per-pixel translations, coordinates and - when asked -
the image.  Tensors must live on a ROCm device; there is no CPU path (a CPU tensor
raises ``TypeError``), and a missing library raises ``ImportError``.

Errors: the reference raises nothing itself (TF shape errors at graph build); here
``TypeError`` = wrong type / dtype / device, ``ValueError`` = wrong shape.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream_of

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- checks
def _check(t, name: str, ndim: int, last: Optional[int] = None) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference path is fp32), got {t.dtype}")
    if t.device.type != "cuda":
        raise TypeError(f"{name} is on {t.device}; sfm_amd runs on ROCm devices only (no CPU fallback)")
    if t.dim() != ndim:
        raise ValueError(f"{name} must have {ndim} dimensions, got shape {tuple(t.shape)}")
    if last is not None and t.shape[-1] != last:
        raise ValueError(f"{name} must have last dimension {last}, got shape {tuple(t.shape)}")
    return t.contiguous()


def _same_device(*ts):
    dev = ts[0].device
    for t in ts:
        if t is not None and t.device != dev:
            raise TypeError(f"all tensors must be on one device, got {dev} and {t.device}")
    return dev


def _pose_params(pose: Tensor, intrinsics: Tensor, inverse_pose: bool) -> Tensor:
    bsz = pose.shape[0]
    params = torch.empty(bsz, _lib.PARAM_STRIDE, dtype=torch.float32, device=pose.device)
    call("sfm_pose_setup", ptr(pose), 6, 0, ptr(intrinsics), bsz, 1, int(inverse_pose), ptr(params), stream_of(pose))
    return params


def _pose_grad(pose, intrinsics, partials, nblk, inverse_pose) -> Tensor:
    bsz = pose.shape[0]
    d_pose = torch.empty_like(pose)
    call("sfm_pose_grad", ptr(pose), 6, 0, ptr(intrinsics), ptr(partials), 1, nblk, bsz, 1, int(inverse_pose), 1.0,
         ptr(d_pose), 6, 0, stream_of(pose))
    return d_pose


# ----------------------------------------------------------------------------- gather
class _InverseWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, pose, intrinsics, trans, inverse_pose):
        bsz, h, w, ch = img.shape
        with torch.cuda.device(img.device):
            params = _pose_params(pose, intrinsics, inverse_pose)
            out = torch.empty_like(img)
            mask = torch.empty(bsz, h, w, 1, dtype=torch.float32, device=img.device)
            call("sfm_inverse_warp_fwd", ptr(img), ptr(depth), ptr(trans), ptr(params), bsz, h, w, ch, ptr(out),
                 ptr(mask), None, stream_of(img))
        ctx.save_for_backward(img, depth, pose, intrinsics, trans, params)
        ctx.inverse_pose = inverse_pose
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, gout, _gmask):
        img, depth, pose, intrinsics, trans, params = ctx.saved_tensors
        bsz, h, w, ch = img.shape
        gout = gout.contiguous()
        with torch.cuda.device(img.device):
            nblk = _lib.load().sfm_bwd_blocks(h, w)
            d_depth = torch.empty_like(depth)
            d_trans = torch.empty_like(trans) if trans is not None else None
            partials = torch.empty(bsz, nblk, 12, dtype=torch.float32, device=img.device)
            d_img = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
            call("sfm_inverse_warp_bwd", ptr(img), ptr(depth), ptr(trans), ptr(params), ptr(gout), bsz, h, w, ch,
                 ptr(d_depth), ptr(d_trans), ptr(partials), ptr(d_img), stream_of(img))
            d_pose = _pose_grad(pose, intrinsics, partials, nblk, ctx.inverse_pose)
        return d_img, d_depth, d_pose, None, d_trans, None


def _warp_args(img, depth, pose, intrinsics, add_trans, translations):
    img = _check(img, "img", 4)
    bsz, h, w, _ = img.shape
    depth = _check(depth, "depth", 3)
    pose = _check(pose, "pose", 2, 6)
    intrinsics = _check(intrinsics, "intrinsics", 3, 3)
    if tuple(depth.shape) != (bsz, h, w):
        raise ValueError(f"depth must be [batch, height, width] = {(bsz, h, w)}, got {tuple(depth.shape)}")
    if tuple(pose.shape) != (bsz, 6) or tuple(intrinsics.shape) != (bsz, 3, 3):
        raise ValueError("pose must be [batch, 6] and intrinsics [batch, 3, 3]")
    trans = None
    if add_trans:
        trans = _check(translations, "translations", 4, 3)
        if tuple(trans.shape) != (bsz, h, w, 3):
            raise ValueError(f"translations must be {(bsz, h, w, 3)}, got {tuple(trans.shape)}")
    _same_device(img, depth, pose, intrinsics, trans)
    return img, depth, pose, intrinsics, trans


def projective_inverse_warp(img, depth, pose, intrinsics, inverse_pose=False, add_trans=False, translations=None):
    """Inverse warp a source image to the target image plane (utils.py:207-239).

    img [B,H,W,C] source image; depth [B,H,W] target depth; pose [B,6] = (tx,ty,tz,rx,ry,rz);
    intrinsics [B,3,3].  Returns (output_img [B,H,W,C], output_mask [B,H,W,1])."""
    img, depth, pose, intrinsics, trans = _warp_args(img, depth, pose, intrinsics, add_trans, translations)
    return _InverseWarp.apply(img, depth, pose, intrinsics, trans, bool(inverse_pose))


class _Sampler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, coords):
        bsz, hs, ws, ch = imgs.shape
        _, ht, wt, _ = coords.shape
        with torch.cuda.device(imgs.device):
            out = torch.empty(bsz, ht, wt, ch, dtype=torch.float32, device=imgs.device)
            mask = torch.empty(bsz, ht, wt, 1, dtype=torch.float32, device=imgs.device)
            call("sfm_bilinear_sampler_fwd", ptr(imgs), ptr(coords), bsz, hs, ws, ch, ht, wt, ptr(out), ptr(mask), None,
                 stream_of(imgs))
        ctx.save_for_backward(imgs, coords)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, gout, _gmask):
        imgs, coords = ctx.saved_tensors
        bsz, hs, ws, ch = imgs.shape
        _, ht, wt, _ = coords.shape
        gout = gout.contiguous()
        with torch.cuda.device(imgs.device):
            d_coords = torch.empty_like(coords)
            d_img = torch.zeros_like(imgs) if ctx.needs_input_grad[0] else None
            call("sfm_bilinear_sampler_bwd", ptr(imgs), ptr(coords), ptr(gout), bsz, hs, ws, ch, ht, wt, ptr(d_coords),
                 ptr(d_img), stream_of(imgs))
        return d_img, d_coords


def bilinear_sampler(imgs, coords):
    """4-tap bilinear gather (utils.py:241-340).  imgs [B,Hs,Ws,C], coords [B,Ht,Wt,2] = (x, y).
    Returns (output [B,Ht,Wt,C], mask [B,Ht,Wt,1])."""
    imgs = _check(imgs, "imgs", 4)
    coords = _check(coords, "coords", 4, 2)
    if coords.shape[0] != imgs.shape[0]:
        raise ValueError("imgs and coords must share the batch dimension")
    _same_device(imgs, coords)
    return _Sampler.apply(imgs, coords)


# ----------------------------------------------------------------------------- splat
class _SplatWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, pose, intrinsics, trans, inverse_pose):
        bsz, h, w, ch = img.shape
        with torch.cuda.device(img.device):
            params = _pose_params(pose, intrinsics, inverse_pose)
            acc = torch.empty(bsz, h, w, ch + 1, dtype=torch.float32, device=img.device)
            out = torch.empty_like(img)
            mask = torch.empty_like(img)
            call("sfm_splat_warp_fwd", ptr(img), ptr(depth), ptr(trans), ptr(params), bsz, h, w, ch, ptr(acc), ptr(out),
                 ptr(mask), stream_of(img))
        ctx.save_for_backward(img, depth, pose, intrinsics, trans, params, acc)
        ctx.inverse_pose = inverse_pose
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, gout, _gmask):
        img, depth, pose, intrinsics, trans, params, acc = ctx.saved_tensors
        bsz, h, w, ch = img.shape
        gout = gout.contiguous()
        with torch.cuda.device(img.device):
            nblk = _lib.load().sfm_bwd_blocks(h, w)
            d_depth = torch.empty_like(depth)
            d_trans = torch.empty_like(trans) if trans is not None else None
            partials = torch.empty(bsz, nblk, 12, dtype=torch.float32, device=img.device)
            d_img = torch.empty_like(img) if ctx.needs_input_grad[0] else None
            call("sfm_splat_warp_bwd", ptr(img), ptr(depth), ptr(trans), ptr(params), ptr(acc), ptr(gout), bsz, h, w,
                 ch, ptr(d_depth), ptr(d_trans), ptr(partials), ptr(d_img), stream_of(img))
            d_pose = _pose_grad(pose, intrinsics, partials, nblk, ctx.inverse_pose)
        return d_img, d_depth, d_pose, None, d_trans, None


def projective_warp(img, depth, pose, intrinsics, inverse_pose=False, add_trans=False, translations=None):
    """Forward-splat ``img`` (the target image) into the source view (utils.py:172-205).

    Returns (output_img [B,H,W,C], output_mask [B,H,W,C])."""
    img, depth, pose, intrinsics, trans = _warp_args(img, depth, pose, intrinsics, add_trans, translations)
    return _SplatWarp.apply(img, depth, pose, intrinsics, trans, bool(inverse_pose))


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, coords, z):
        bsz, h, w, ch = imgs.shape
        with torch.cuda.device(imgs.device):
            acc = torch.empty(bsz, h, w, ch + 1, dtype=torch.float32, device=imgs.device)
            out = torch.empty_like(imgs)
            mask = torch.empty_like(imgs)
            call("sfm_bilinear_project_fwd", ptr(imgs), ptr(coords), ptr(z), bsz, h, w, ch, ptr(acc), ptr(out),
                 ptr(mask), stream_of(imgs))
        ctx.save_for_backward(imgs, coords, z, acc)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, gout, _gmask):
        imgs, coords, z, acc = ctx.saved_tensors
        bsz, h, w, ch = imgs.shape
        gout = gout.contiguous()
        with torch.cuda.device(imgs.device):
            d_coords = torch.empty_like(coords)
            d_z = torch.empty_like(z) if z is not None else None
            d_img = torch.empty_like(imgs) if ctx.needs_input_grad[0] else None
            call("sfm_bilinear_project_bwd", ptr(imgs), ptr(coords), ptr(z), ptr(acc), ptr(gout), bsz, h, w, ch,
                 ptr(d_coords), ptr(d_z), ptr(d_img), stream_of(imgs))
        return d_img, d_coords, d_z


def bilinear_project(imgs, coords, use_depth=False, depth=None):
    """Gaussian / depth weighted forward splat (utils.py:342-502).  imgs [B,H,W,C], coords [B,H,W,2],
    depth (if use_depth) any shape with B*H*W elements (utils.py:437 reshapes it).
    Returns (output [B,H,W,C], mask [B,H,W,C])."""
    imgs = _check(imgs, "imgs", 4)
    coords = _check(coords, "coords", 4, 2)
    bsz, h, w, _ = imgs.shape
    if tuple(coords.shape[:3]) != (bsz, h, w):
        raise ValueError("bilinear_project needs coords of the image's size (one coordinate per splatted pixel)")
    z = None
    if use_depth:
        if not isinstance(depth, torch.Tensor):
            raise TypeError("use_depth=True needs depth")
        if depth.numel() != bsz * h * w:
            raise ValueError("depth must have batch*height*width elements")
        z = _check(depth.reshape(bsz, h, w), "depth", 3)
    _same_device(imgs, coords, z)
    return _Project.apply(imgs, coords, z)


# ----------------------------------------------------------------------------- SSIM
class _SsimValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        bsz, h, w, _ = img1.shape
        with torch.cuda.device(img1.device):
            out = torch.empty_like(img1)
            call("sfm_ssim_fwd", ptr(img1), ptr(img2), bsz, h, w, ptr(out), stream_of(img1))
        ctx.save_for_backward(img1, img2)
        return out

    @staticmethod
    def backward(ctx, gv):
        img1, img2 = ctx.saved_tensors
        bsz, h, w, _ = img1.shape
        gv = gv.contiguous()
        with torch.cuda.device(img1.device):
            coef = torch.empty(bsz, h, w, 15, dtype=torch.float32, device=img1.device)
            d1 = torch.empty_like(img1) if ctx.needs_input_grad[0] else None
            d2 = torch.empty_like(img2) if ctx.needs_input_grad[1] else None
            if d1 is None and d2 is None:
                return None, None
            call("sfm_ssim_bwd", ptr(img1), ptr(img2), ptr(gv), bsz, h, w, ptr(coef), ptr(d1), ptr(d2), stream_of(img1))
        return d1, d2


def ssim_value(img1, img2):
    """The raw SSIM map ``value`` of utils.py:572 (3x3 window, sigma 1.5), [B,H,W,3]."""
    img1 = _check(img1, "img1", 4, 3)
    img2 = _check(img2, "img2", 4, 3)
    if img1.shape != img2.shape:
        raise ValueError("img1 and img2 must have the same shape")
    _same_device(img1, img2)
    return _SsimValue.apply(img1, img2)


def tf_ssim(img1, img2, cs_map=False, mean_metric=True, size=3, sigma=1.5):
    """utils.py:548-575: clip((1 - ssim)/2, 0, 1) of the map, or of its global mean."""
    if cs_map:
        raise NotImplementedError("cs_map=True is only reached from the dead tf_ms_ssim (utils.py:578)")
    if size != 3 or sigma != 1.5:
        raise NotImplementedError("only the window the training graph uses (size=3, sigma=1.5) is implemented")
    value = ssim_value(img1, img2)
    if mean_metric:
        value = value.mean()
    return torch.clamp((1 - value) / 2.0, 0, 1)


# ----------------------------------------------------------------------------- geometry helpers
def project_coords(depth, pose, intrinsics, inverse_pose=False, add_trans=False, translations=None):
    """meshgrid -> pixel2cam -> cam2pixel in one launch (utils.py:185-202): returns
    (src_pixel_coords [B,H,W,2], new_depth [B,H,W,1]).  Forward only."""
    depth = _check(depth, "depth", 3)
    bsz, h, w = depth.shape
    pose = _check(pose, "pose", 2, 6)
    intrinsics = _check(intrinsics, "intrinsics", 3, 3)
    trans = _check(translations, "translations", 4, 3) if add_trans else None
    _same_device(depth, pose, intrinsics, trans)
    with torch.cuda.device(depth.device), torch.no_grad():
        params = _pose_params(pose, intrinsics, inverse_pose)
        coords = torch.empty(bsz, h, w, 2, dtype=torch.float32, device=depth.device)
        z = torch.empty(bsz, h, w, 1, dtype=torch.float32, device=depth.device)
        call("sfm_project_coords", ptr(depth), ptr(trans), ptr(params), bsz, h, w, ptr(coords), ptr(z), stream_of(depth))
    return coords, z


def corner_indices(imgs_shape, coords):
    """int32 [B,Ht,Wt,4] = the clipped (x0, x1, y0, y1) of utils.py:286-289 for an image of
    ``imgs_shape`` = (B, Hs, Ws, C) - the integer pixel indices both warps address."""
    coords = _check(coords, "coords", 4, 2)
    bsz, hs, ws, ch = imgs_shape
    _, ht, wt, _ = coords.shape
    with torch.cuda.device(coords.device), torch.no_grad():
        dummy = torch.zeros(bsz, hs, ws, ch, dtype=torch.float32, device=coords.device)
        out = torch.empty(bsz, ht, wt, ch, dtype=torch.float32, device=coords.device)
        mask = torch.empty(bsz, ht, wt, 1, dtype=torch.float32, device=coords.device)
        idx = torch.empty(bsz, ht, wt, 4, dtype=torch.int32, device=coords.device)
        call("sfm_bilinear_sampler_fwd", ptr(dummy), ptr(coords), bsz, hs, ws, ch, ht, wt, ptr(out), ptr(mask), ptr(idx),
             stream_of(coords))
    return idx


# ---- the geometry helpers: hand-written kernels in both directions (csrc/geom.hip).  Forward in the oracle's canonical
# fp32 op order (bit-exact); the adjoints are what tf.gradients derives for a caller that composes utils.py:185-202 from
# the pieces (the fused operators above never call them).

def _geom_ws(B, H, W, device):
    return torch.empty(_lib.load().sfm_geom_bwd_workspace_floats(B, H, W), dtype=torch.float32, device=device)


class _Euler2Mat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, y, x):
        ctx.save_for_backward(z, y, x)
        return _euler2mat_hip(z, y, x)

    @staticmethod
    def backward(ctx, g):
        z, y, x = ctx.saved_tensors
        gz, gy, gx = torch.empty_like(z), torch.empty_like(y), torch.empty_like(x)
        with torch.cuda.device(z.device):
            call("sfm_euler2mat_bwd", ptr(z), ptr(y), ptr(x), ptr(g.contiguous()), z.shape[0], ptr(gz), ptr(gy), ptr(gx),
                 stream_of(z))
        return gz, gy, gx


class _PoseVec2Mat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec):
        ctx.save_for_backward(vec)
        return _pose_vec2mat_hip(vec)

    @staticmethod
    def backward(ctx, g):
        vec, = ctx.saved_tensors
        gvec = torch.empty_like(vec)
        with torch.cuda.device(vec.device):
            call("sfm_pose_vec2mat_bwd", ptr(vec), ptr(g.contiguous()), vec.shape[0], ptr(gvec), stream_of(vec))
        return gvec


class _Pixel2Cam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, is_homogeneous, depth, pixel_coords, intrinsics, translations):
        ctx.homog = bool(is_homogeneous)
        ctx.save_for_backward(depth, pixel_coords, intrinsics, *([translations] if translations is not None else []))
        return _pixel2cam_hip(ctx.homog)(depth, pixel_coords, intrinsics, translations)

    @staticmethod
    def backward(ctx, g):
        depth, pix, K = ctx.saved_tensors[:3]
        trans = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        B, H, W = depth.shape
        need = ctx.needs_input_grad
        dev = depth.device
        with torch.cuda.device(dev):
            g_depth = torch.empty_like(depth) if need[1] else None
            g_pix = torch.empty_like(pix) if need[2] else None
            g_K = torch.empty_like(K) if need[3] else None
            g_t = torch.empty_like(trans) if (trans is not None and need[4]) else None
            call("sfm_pixel2cam_bwd", ptr(depth), ptr(pix), ptr(K), ptr(trans), ptr(g.contiguous()), B, H, W, int(ctx.homog),
                 ptr(g_depth), ptr(g_pix), ptr(g_K), ptr(g_t), ptr(_geom_ws(B, H, W, dev)), stream_of(depth))
        return None, g_depth, g_pix, g_K, g_t


class _Cam2Pixel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam_coords, proj):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(cam_coords, proj)
        return _cam2pixel_hip(cam_coords, proj)

    @staticmethod
    def backward(ctx, g_xy, g_z):
        cam, proj = ctx.saved_tensors
        B, _, H, W = cam.shape
        need = ctx.needs_input_grad
        if (g_xy is None and g_z is None) or not (need[0] or need[1]):
            return None, None
        dev = cam.device
        with torch.cuda.device(dev):
            g_cam = torch.empty_like(cam) if need[0] else None
            g_proj = torch.empty_like(proj) if need[1] else None
            call("sfm_cam2pixel_bwd", ptr(cam), ptr(proj), ptr(g_xy.contiguous() if g_xy is not None else None),
                 ptr(g_z.contiguous() if g_z is not None else None), B, H, W, ptr(g_cam), ptr(g_proj),
                 ptr(_geom_ws(B, H, W, dev)), stream_of(cam))
        return g_cam, g_proj


def _euler2mat_hip(z, y, x):
    bsz = z.shape[0]
    out = torch.empty(bsz, 1, 3, 3, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        call("sfm_euler2mat", ptr(z), ptr(y), ptr(x), bsz, ptr(out), stream_of(z))
    return out


def euler2mat(z, y, x):
    """utils.py:27-75: [B,1] angles -> [B,1,3,3], R = X.Y.Z, angles clipped to +-pi (``sfm_euler2mat``)."""
    z, y, x = (_check(t.reshape(t.shape[0], 1) if isinstance(t, torch.Tensor) and t.dim() == 1 else t, n, 2, 1)
               for t, n in ((z, "z"), (y, "y"), (x, "x")))
    if not (z.shape == y.shape == x.shape):
        raise ValueError("z, y, x must have the same shape [B, 1]")
    _same_device(z, y, x)
    return _Euler2Mat.apply(z, y, x)


def _pose_vec2mat_hip(vec):
    out = torch.empty(vec.shape[0], 4, 4, dtype=torch.float32, device=vec.device)
    with torch.cuda.device(vec.device):
        call("sfm_pose_vec2mat", ptr(vec), vec.shape[0], ptr(out), stream_of(vec))
    return out


def pose_vec2mat(vec):
    """utils.py:77-96: [B,6] = (tx,ty,tz,rx,ry,rz) -> [B,4,4] (``sfm_pose_vec2mat``)."""
    vec = _check(vec, "vec", 2, 6)
    return _PoseVec2Mat.apply(vec)


def meshgrid(batch, height, width, is_homogeneous=True, device=None):
    """utils.py:146-170 -> [B, 3 (2), H, W] with the reference's fp32 linspace values (``sfm_meshgrid``).
    ``device`` (not in the reference: TF places the op): the ROCm device, default the current one."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise TypeError(f"meshgrid on {dev}: sfm_amd runs on ROCm devices only (no CPU fallback)")
    if min(int(batch), int(height), int(width)) < 1:
        raise ValueError("batch, height and width must be positive")
    out = torch.empty(batch, 3 if is_homogeneous else 2, height, width, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        call("sfm_meshgrid", int(batch), int(height), int(width), int(bool(is_homogeneous)), ptr(out),
             stream_of(out))
    return out


def _pixel2cam_hip(is_homogeneous):
    def fn(depth, pixel_coords, intrinsics, translations=None):
        batch, height, width = depth.shape
        out = torch.empty(batch, 4 if is_homogeneous else 3, height, width, dtype=torch.float32, device=depth.device)
        with torch.cuda.device(depth.device):
            kinv = torch.empty(batch, 9, dtype=torch.float32, device=depth.device)
            call("sfm_pixel2cam", ptr(depth), ptr(pixel_coords), ptr(intrinsics), ptr(translations), batch, height, width,
                 int(is_homogeneous), ptr(kinv), ptr(out), stream_of(depth))
        return out
    return fn


def pixel2cam(depth, pixel_coords, intrinsics, is_homogeneous=True, add_trans=False, translations=None):
    """utils.py:98-122: depth [B,H,W], pixel_coords [B,3,H,W], intrinsics [B,3,3] (, translations [B,H,W,3]) ->
    camera points [B, 4 (3), H, W] (``sfm_pixel2cam``)."""
    depth = _check(depth, "depth", 3)
    batch, height, width = depth.shape
    pixel_coords = _check(pixel_coords, "pixel_coords", 4)
    intrinsics = _check(intrinsics, "intrinsics", 3, 3)
    if tuple(pixel_coords.shape) != (batch, 3, height, width) or tuple(intrinsics.shape) != (batch, 3, 3):
        raise ValueError("pixel_coords must be [batch, 3, height, width] and intrinsics [batch, 3, 3]")
    args = [depth, pixel_coords, intrinsics]
    if add_trans:
        trans = _check(translations, "translations", 4, 3)
        if tuple(trans.shape) != (batch, height, width, 3):
            raise ValueError(f"translations must be {(batch, height, width, 3)}, got {tuple(trans.shape)}")
        args.append(trans)
    _same_device(*args)
    return _Pixel2Cam.apply(bool(is_homogeneous), *args, *([None] if len(args) == 3 else []))


def _cam2pixel_hip(cam_coords, proj):
    batch, _, height, width = cam_coords.shape
    coords = torch.empty(batch, height, width, 2, dtype=torch.float32, device=cam_coords.device)
    z = torch.empty(batch, height, width, 1, dtype=torch.float32, device=cam_coords.device)
    with torch.cuda.device(cam_coords.device):
        call("sfm_cam2pixel", ptr(cam_coords), ptr(proj), batch, height, width, ptr(coords), ptr(z), stream_of(cam_coords))
    return coords, z


def cam2pixel(cam_coords, proj) -> Tuple[Tensor, Tensor]:
    """utils.py:124-144: cam_coords [B,4,H,W], proj [B,4,4] -> (coords [B,H,W,2], depth [B,H,W,1]) (``sfm_cam2pixel``)."""
    cam_coords = _check(cam_coords, "cam_coords", 4)
    proj = _check(proj, "proj", 3, 4)
    if cam_coords.shape[1] != 4 or tuple(proj.shape) != (cam_coords.shape[0], 4, 4):
        raise ValueError("cam_coords must be [batch, 4, height, width] and proj [batch, 4, 4]")
    _same_device(cam_coords, proj)
    return _Cam2Pixel.apply(cam_coords, proj)
