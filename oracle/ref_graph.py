"""CPU restatement of the alncat/SFM view-synthesis hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  The product
path (``sfm_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference (``/root/reference``) is TensorFlow-1.x graph code;
TensorFlow is not installed in the build image (``ModuleNotFoundError``), the
reference ships no tests, golden vectors or fixtures, and
there is no network.  This file is therefore an op-for-op restatement of the
reference's Python in PyTorch-CPU, pinned only by analytic known-answer tests
(``tests/test_oracle_*.py``), by an independent NumPy per-pixel restatement
(``oracle/np_forward.py``) and by ``torch.autograd.gradcheck`` in fp64.

Every function cites the reference ``file:line`` it follows (paths relative to
``/root/reference``).  TF-1 semantics that cannot be verified here are marked
ASSUMPTION.

Canonical fp32 op order (this is what "bit-exact projected pixel indices" is
defined against):
  * every ``tf.matmul`` is a k-ordered sum of separately rounded products (Eigen's
    gebp order without FMA contraction - the official TF-1 CPU wheels were built
    without FMA);                                                       ASSUMPTION
  * ``tf.matrix_inverse`` is a partial-pivot LU followed by a unit-lower forward
    and an upper backward substitution against the permuted identity (Eigen
    ``PartialPivLU::inverse``);                                          ASSUMPTION
  * ``tf.sin`` / ``tf.cos`` of the three Euler angles are correctly rounded
    (evaluated in fp64, rounded once to fp32);                           ASSUMPTION
  * ``tf.linspace(a, b, n)[i] = a + i * ((b - a) / (n - 1))`` in fp32 (TF-1
    ``LinSpaceOp``).                                                     ASSUMPTION
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# small dense helpers (canonical op order)
# ----------------------------------------------------------------------------
def matmul_seq(a: Tensor, x: Tensor) -> Tensor:
    """``tf.matmul`` restated: a[B,n,k] @ x[B,k,m], k-ordered, no FMA."""
    rows = []
    for i in range(a.shape[1]):
        acc = a[:, i, 0, None] * x[:, 0, :]
        for k in range(1, a.shape[2]):
            acc = acc + a[:, i, k, None] * x[:, k, :]
        rows.append(acc)
    return torch.stack(rows, dim=1)


def matrix_inverse_lu(a: Tensor) -> Tensor:
    """``tf.matrix_inverse`` restated (utils.py:112, 189, 224): partial-pivot LU,
    then L y = P I (unit lower, forward) and U x = y (backward).  Batched, and
    differentiable through torch autograd (row swaps are gathers)."""
    bsz, n, _ = a.shape
    rows = [a[:, i, :] for i in range(n)]
    eye = torch.eye(n, dtype=a.dtype, device=a.device)
    rhs = [eye[i].expand(bsz, n) for i in range(n)]
    for k in range(n):
        col = torch.stack([rows[i][:, k].abs() for i in range(k, n)], dim=1)
        p = col.argmax(dim=1) + k  # first maximal element wins
        stacked = torch.stack(rows, dim=1)
        stacked_rhs = torch.stack(rhs, dim=1)
        gidx = p[:, None, None].expand(bsz, 1, n)
        rp = stacked.gather(1, gidx).squeeze(1)
        rp_rhs = stacked_rhs.gather(1, gidx).squeeze(1)
        rk, rk_rhs = rows[k], rhs[k]
        for i in range(k + 1, n):
            sel = (p == i)[:, None]
            rows[i] = torch.where(sel, rk, rows[i])
            rhs[i] = torch.where(sel, rk_rhs, rhs[i])
        rows[k], rhs[k] = rp, rp_rhs
        for i in range(k + 1, n):
            l = rows[i][:, k] / rows[k][:, k]
            upd = rows[i][:, k + 1:] - l[:, None] * rows[k][:, k + 1:]
            rows[i] = torch.cat([rows[i][:, :k], l[:, None], upd], dim=1)
    # forward substitution, unit lower triangle
    y: List[Tensor] = []
    for i in range(n):
        acc = rhs[i]
        for j in range(i):
            acc = acc - rows[i][:, j, None] * y[j]
        y.append(acc)
    # backward substitution
    x: List[Optional[Tensor]] = [None] * n
    for i in range(n - 1, -1, -1):
        acc = y[i]
        for j in range(i + 1, n):
            acc = acc - rows[i][:, j, None] * x[j]
        x[i] = acc / rows[i][:, i, None]
    return torch.stack(x, dim=1)


def _sincos(v: Tensor) -> Tuple[Tensor, Tensor]:
    """Correctly rounded sin/cos: evaluate in fp64, round once."""
    if v.dtype == torch.float64:
        return torch.sin(v), torch.cos(v)
    vd = v.double()
    return torch.sin(vd).to(v.dtype), torch.cos(vd).to(v.dtype)


def linspace_tf1(start: float, stop: float, n: int, dtype: torch.dtype) -> Tensor:
    """TF-1 ``LinSpaceOp``: start + i * ((stop - start) / (n - 1)), in ``dtype``."""
    s = torch.tensor(start, dtype=dtype)
    if n == 1:
        return s.reshape(1)
    step = (torch.tensor(stop, dtype=dtype) - s) / torch.tensor(n - 1, dtype=dtype)
    return s + step * torch.arange(n, dtype=dtype)


# ----------------------------------------------------------------------------
# geometry: utils.py:27-170
# ----------------------------------------------------------------------------
def euler2mat(z: Tensor, y: Tensor, x: Tensor) -> Tensor:
    """utils.py:27-75.  z, y, x: [B, 1] -> [B, 1, 3, 3];  R = (X . Y) . Z with the
    angles clipped to [-pi, pi] (np.pi is rounded to the tensor dtype, as a TF
    constant would be)."""
    bsz = z.shape[0]
    z = torch.clamp(z, -math.pi, math.pi).reshape(bsz)
    y = torch.clamp(y, -math.pi, math.pi).reshape(bsz)
    x = torch.clamp(x, -math.pi, math.pi).reshape(bsz)
    zeros = torch.zeros_like(z)
    ones = torch.ones_like(z)
    sinz, cosz = _sincos(z)
    siny, cosy = _sincos(y)
    sinx, cosx = _sincos(x)

    def m3(r0, r1, r2):
        return torch.stack([torch.stack(r0, 1), torch.stack(r1, 1), torch.stack(r2, 1)], 1)

    zmat = m3([cosz, -sinz, zeros], [sinz, cosz, zeros], [zeros, zeros, ones])
    ymat = m3([cosy, zeros, siny], [zeros, ones, zeros], [-siny, zeros, cosy])
    xmat = m3([ones, zeros, zeros], [zeros, cosx, -sinx], [zeros, sinx, cosx])
    rot = matmul_seq(matmul_seq(xmat, ymat), zmat)  # utils.py:74
    return rot[:, None]


def pose_vec2mat(vec: Tensor) -> Tensor:
    """utils.py:77-96.  vec [B,6] = (tx,ty,tz,rx,ry,rz) -> [B,4,4]."""
    bsz = vec.shape[0]
    translation = vec[:, 0:3, None]
    rx, ry, rz = vec[:, 3:4], vec[:, 4:5], vec[:, 5:6]
    rot = euler2mat(rz, ry, rx).squeeze(1)  # utils.py:90
    filler = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=vec.dtype).reshape(1, 1, 4).repeat(bsz, 1, 1)
    top = torch.cat([rot, translation], dim=2)
    return torch.cat([top, filler], dim=1)


def meshgrid(batch: int, height: int, width: int, is_homogeneous: bool = True,
             dtype: torch.dtype = torch.float32) -> Tensor:
    """utils.py:146-170 -> [B, 3 (2), H, W].  The ones-matmuls are exact copies."""
    lin_x = linspace_tf1(-1.0, 1.0, width, dtype)
    lin_y = linspace_tf1(-1.0, 1.0, height, dtype)
    x_t = lin_x[None, :].expand(height, width)
    y_t = lin_y[:, None].expand(height, width)
    x_t = (x_t + 1.0) * 0.5 * torch.tensor(width - 1, dtype=dtype)  # utils.py:162
    y_t = (y_t + 1.0) * 0.5 * torch.tensor(height - 1, dtype=dtype)
    if is_homogeneous:
        coords = torch.stack([x_t, y_t, torch.ones_like(x_t)], dim=0)
    else:
        coords = torch.stack([x_t, y_t], dim=0)
    return coords[None].repeat(batch, 1, 1, 1)


def pixel2cam(depth: Tensor, pixel_coords: Tensor, intrinsics: Tensor, is_homogeneous: bool = True,
              add_trans: bool = False, translations: Optional[Tensor] = None) -> Tensor:
    """utils.py:98-122."""
    batch, height, width = depth.shape
    depth = depth.reshape(batch, 1, -1)
    pixel_coords = pixel_coords.reshape(batch, 3, -1)
    cam_coords = matmul_seq(matrix_inverse_lu(intrinsics), pixel_coords) * depth  # :112
    if add_trans:
        t = translations.permute(0, 3, 1, 2).reshape(batch, 3, -1)  # :114-115
        cam_coords = cam_coords + t * cam_coords  # :116
    if is_homogeneous:
        ones = torch.ones(batch, 1, height * width, dtype=depth.dtype)
        cam_coords = torch.cat([cam_coords, ones], dim=1)
    return cam_coords.reshape(batch, -1, height, width)


def cam2pixel(cam_coords: Tensor, proj: Tensor) -> Tuple[Tensor, Tensor]:
    """utils.py:124-144 -> (coords [B,H,W,2], depth [B,H,W,1])."""
    batch, _, height, width = cam_coords.shape
    cam = cam_coords.reshape(batch, 4, -1)
    unnorm = matmul_seq(proj, cam)  # :135
    x_u, y_u, z_u = unnorm[:, 0:1], unnorm[:, 1:2], unnorm[:, 2:3]
    x_n = x_u / (z_u + 1e-10)
    y_n = y_u / (z_u + 1e-10)
    pix = torch.cat([x_n, y_n], dim=1).reshape(batch, 2, height, width)
    depth = z_u.reshape(batch, 1, height, width)
    return pix.permute(0, 2, 3, 1), depth.permute(0, 2, 3, 1)


def _compose_proj(pose: Tensor, intrinsics: Tensor, inverse_pose: bool) -> Tensor:
    """utils.py:187-201 / 222-236: P = [[K,0],[0,1]] . T(pose)."""
    batch = pose.shape[0]
    t = pose_vec2mat(pose)
    if inverse_pose:
        t = matrix_inverse_lu(t)
    filler = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=pose.dtype).reshape(1, 1, 4).repeat(batch, 1, 1)
    k4 = torch.cat([intrinsics, torch.zeros(batch, 3, 1, dtype=pose.dtype)], dim=2)
    k4 = torch.cat([k4, filler], dim=1)
    return matmul_seq(k4, t)


def project_coords(depth, pose, intrinsics, inverse_pose=False, add_trans=False, translations=None):
    """Shared front half of both warps (utils.py:185-202 == 220-237)."""
    batch, height, width = depth.shape
    proj = _compose_proj(pose, intrinsics, inverse_pose)
    pix = meshgrid(batch, height, width, dtype=depth.dtype)
    cam = pixel2cam(depth, pix, intrinsics, add_trans=add_trans, translations=translations)
    return cam2pixel(cam, proj)


# ----------------------------------------------------------------------------
# the two warps: utils.py:172-502
# ----------------------------------------------------------------------------
def corner_indices(coords: Tensor, height: int, width: int):
    """floor / +1 / clip of utils.py:274-289 == 376-388; returns the four clipped
    corner coordinates as floats and as int32 (x0, x1, y0, y1)."""
    cx, cy = coords[..., 0:1], coords[..., 1:2]
    x0 = torch.floor(cx)
    x1 = x0 + 1
    y0 = torch.floor(cy)
    y1 = y0 + 1
    x_max, y_max = float(width - 1), float(height - 1)
    x0s, x1s = torch.clamp(x0, 0.0, x_max), torch.clamp(x1, 0.0, x_max)
    y0s, y1s = torch.clamp(y0, 0.0, y_max), torch.clamp(y1, 0.0, y_max)
    ints = tuple(t.to(torch.int32) for t in (x0s, x1s, y0s, y1s))
    return (x0s, x1s, y0s, y1s), ints


def bilinear_sampler(imgs: Tensor, coords: Tensor, float_index: bool = False) -> Tuple[Tensor, Tensor]:
    """utils.py:241-340 -> (output [B,H,W,C], mask [B,H,W,1]).

    ``float_index=True`` reproduces the reference's fp32 flat index (quirk 4,
    utils.py:303-317, exact only while B*H*W <= 2**24); the default computes the
    same index in integers (identical wherever the reference is itself exact)."""
    bsz, h_s, w_s, ch = imgs.shape
    _, h_t, w_t, _ = coords.shape
    cx, cy = coords[..., 0:1], coords[..., 1:2]
    (x0s, x1s, y0s, y1s), (ix0, ix1, iy0, iy1) = corner_indices(coords, h_s, w_s)
    x_max, y_max = float(w_s - 1), float(h_s - 1)
    x_out = (cx < -1) | (cx > x_max + 1)  # :283
    y_out = (cy < -1) | (cy > y_max + 1)
    wt_x0 = x1s - cx  # :297-300
    wt_x1 = cx - x0s
    wt_y0 = y1s - cy
    wt_y1 = cy - y0s
    if float_index:
        dim2 = torch.tensor(float(w_s), dtype=torch.float32)
        dim1 = torch.tensor(float(w_s * h_s), dtype=torch.float32)
        base = (torch.arange(bsz, dtype=torch.float32) * dim1).reshape(bsz, 1, 1, 1)
        base_y0 = base + y0s.float() * dim2
        base_y1 = base + y1s.float() * dim2
        idx00 = (x0s.float() + base_y0).to(torch.int64)
        idx01 = (x0s.float() + base_y1).to(torch.int64)
        idx10 = (x1s.float() + base_y0).to(torch.int64)
        idx11 = (x1s.float() + base_y1).to(torch.int64)
    else:
        base = (torch.arange(bsz, dtype=torch.int64) * (w_s * h_s)).reshape(bsz, 1, 1, 1)
        base_y0 = base + iy0.long() * w_s
        base_y1 = base + iy1.long() * w_s
        idx00, idx01 = ix0.long() + base_y0, ix0.long() + base_y1
        idx10, idx11 = ix1.long() + base_y0, ix1.long() + base_y1
    flat = imgs.reshape(-1, ch)
    out_size = (bsz, h_t, w_t, ch)
    im00 = flat[idx00.reshape(-1)].reshape(out_size)  # :322-325
    im01 = flat[idx01.reshape(-1)].reshape(out_size)
    im10 = flat[idx10.reshape(-1)].reshape(out_size)
    im11 = flat[idx11.reshape(-1)].reshape(out_size)
    w00, w01, w10, w11 = wt_x0 * wt_y0, wt_x0 * wt_y1, wt_x1 * wt_y0, wt_x1 * wt_y1
    output = ((w00 * im00 + w01 * im01) + w10 * im10) + w11 * im11  # add_n, :332-335
    mask = 1.0 - (x_out | y_out).to(imgs.dtype)  # :338
    return output, mask


def avg_pool3_same(x_nhwc: Tensor) -> Tensor:
    """``slim.avg_pool2d(x, 3, 1, 'SAME')`` (utils.py:479): mean over the
    in-bounds taps only.                                            ASSUMPTION"""
    x = x_nhwc.permute(0, 3, 1, 2)
    y = F.avg_pool2d(x, 3, stride=1, padding=1, count_include_pad=False)
    return y.permute(0, 2, 3, 1)


def bilinear_project(imgs: Tensor, coords: Tensor, use_depth: bool = False,
                     depth: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """utils.py:342-502 forward splat -> (output [B,H,W,C], mask [B,H,W,C])."""
    bsz, h_s, w_s, ch = imgs.shape
    _, h_t, w_t, _ = coords.shape
    cx, cy = coords[..., 0:1], coords[..., 1:2]
    (x0s, x1s, y0s, y1s), (ix0, ix1, iy0, iy1) = corner_indices(coords, h_s, w_s)
    wt_x1 = x1s - cx  # :396-399
    wt_x0 = cx - x0s
    wt_y1 = y1s - cy
    wt_y0 = cy - y0s
    n_pix = bsz * h_s * w_s
    bidx = (torch.arange(n_pix, dtype=torch.int64) // (h_s * w_s))  # :424-425
    def flat_idx(iy, ix):
        return (bidx * h_t + iy.reshape(-1).long()) * w_t + ix.reshape(-1).long()
    idx00, idx01 = flat_idx(iy0, ix0), flat_idx(iy1, ix0)
    idx10, idx11 = flat_idx(iy0, ix1), flat_idx(iy1, ix1)
    dist00 = torch.exp(-(wt_x0 * wt_x0 + wt_y0 * wt_y0) / 2)  # :431-434
    dist01 = torch.exp(-(wt_x0 * wt_x0 + wt_y1 * wt_y1) / 2)
    dist10 = torch.exp(-(wt_x1 * wt_x1 + wt_y0 * wt_y0) / 2)
    dist11 = torch.exp(-(wt_x1 * wt_x1 + wt_y1 * wt_y1) / 2)
    if use_depth:
        d = depth.reshape(bsz, h_t, w_t, 1)
        depth_w = torch.exp(-torch.clamp(d / 4, 0, 10))  # :438
        dist00, dist01 = depth_w * dist00, depth_w * dist01
        dist10, dist11 = depth_w * dist10, depth_w * dist11
    flat = imgs.reshape(-1, ch)
    d00, d01 = dist00.reshape(-1, 1), dist01.reshape(-1, 1)
    d10, d11 = dist10.reshape(-1, 1), dist11.reshape(-1, 1)
    im00, im01, im10, im11 = flat * d00, flat * d01, flat * d10, flat * d11  # :453-456

    def scatter(idx, upd, width):
        return torch.zeros(bsz * h_t * w_t, width, dtype=imgs.dtype).index_add(0, idx, upd)

    s00, s01 = scatter(idx00, d00, 1), scatter(idx01, d01, 1)  # :458-461
    s10, s11 = scatter(idx10, d10, 1), scatter(idx11, d11, 1)
    dists = s00 + s01 + s10 + s11 + 1e-10  # :463
    c00, c01 = scatter(idx00, im00, ch), scatter(idx01, im01, ch)  # :465-468
    c10, c11 = scatter(idx10, im10, ch), scatter(idx11, im11, ch)
    output = (((c00 + c01) + c10) + c11) / dists  # :470
    output = output.reshape(bsz, h_t, w_t, ch)
    mask = 1 - (output == 0).to(imgs.dtype)  # :478
    mask = avg_pool3_same(mask)  # :479
    return output, mask


def projective_warp(img, depth, pose, intrinsics, inverse_pose=False, add_trans=False, translations=None):
    """utils.py:172-205: forward splat of ``img`` (the target image)."""
    coords, new_depth = project_coords(depth, pose, intrinsics, inverse_pose, add_trans, translations)
    return bilinear_project(img, coords, use_depth=True, depth=new_depth)


def projective_inverse_warp(img, depth, pose, intrinsics, inverse_pose=False, add_trans=False,
                            translations=None, float_index=False):
    """utils.py:207-239: 4-tap gather of ``img`` (a source image)."""
    coords, _ = project_coords(depth, pose, intrinsics, inverse_pose, add_trans, translations)
    return bilinear_sampler(img, coords, float_index=float_index)


# ----------------------------------------------------------------------------
# SSIM: utils.py:504-519, 548-575
# ----------------------------------------------------------------------------
def fspecial_gauss(size: int, sigma: float, dtype=torch.float32) -> Tensor:
    """utils.py:504-519 -> [size, size]."""
    r = torch.arange(-size // 2 + 1, size // 2 + 1, dtype=dtype)
    x = r[:, None].expand(size, size)
    y = r[None, :].expand(size, size)
    g = torch.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))
    return g / g.sum()


def _dwconv_same(x_nhwc: Tensor, window: Tensor) -> Tensor:
    """``tf.nn.depthwise_conv2d(..., 'SAME')`` with one shared window (zero pad)."""
    ch = x_nhwc.shape[3]
    pad = window.shape[0] // 2
    w = window[None, None].repeat(ch, 1, 1, 1)
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w, padding=pad, groups=ch)
    return y.permute(0, 2, 3, 1)


def ssim_value(img1: Tensor, img2: Tensor, size: int = 3, sigma: float = 1.5) -> Tensor:
    """The un-clipped SSIM map ``value`` of utils.py:572."""
    window = fspecial_gauss(size, sigma, img1.dtype)
    c1, c2 = (0.01 * 1) ** 2, (0.03 * 1) ** 2
    mu1 = _dwconv_same(img1, window)
    mu2 = _dwconv_same(img2, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq = _dwconv_same(img1 * img1, window) - mu1_sq
    sigma2_sq = _dwconv_same(img2 * img2, window) - mu2_sq
    sigma12 = _dwconv_same(img1 * img2, window) - mu1_mu2
    return ((2 * mu1_mu2 + c1) * (2 * sigma12 + c2)) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2))


def tf_ssim(img1, img2, cs_map=False, mean_metric=True, size=3, sigma=1.5):
    """utils.py:548-575 (``cs_map=True`` is dead code in the reference; not restated)."""
    if cs_map:
        raise NotImplementedError("cs_map=True is only reached from the dead tf_ms_ssim (utils.py:578)")
    value = ssim_value(img1, img2, size, sigma)
    if mean_metric:
        value = value.mean()
    return torch.clamp((1 - value) / 2.0, 0, 1)


# ----------------------------------------------------------------------------
# loss assembly: SfMLearner_uni.py:75-176, 205-219
# ----------------------------------------------------------------------------
def normalize_disp(pred_disp: Sequence[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
    """SfMLearner_uni.py:40, 47: ``pred_disp = [d / tf.reduce_mean(d, axis=[1,2,3], keep_dims=True) ...]``,
    ``pred_depth = [1. / d for d in pred_disp]``."""
    dn = [d / d.mean(dim=(1, 2, 3), keepdim=True) for d in pred_disp]
    return dn, [1.0 / d for d in dn]


def resize_bilinear_tf1(x_nhwc: Tensor, out_h: int, out_w: int) -> Tensor:
    """``tf.image.resize_bilinear(x, [h, w])`` with TF-1 defaults (align_corners=False,
    no half-pixel centres): src = dst * in/out, top-left anchored.  ASSUMPTION"""
    bsz, in_h, in_w, ch = x_nhwc.shape
    if (in_h, in_w) == (out_h, out_w):
        return x_nhwc
    dt = x_nhwc.dtype

    def axis(in_n, out_n):
        scale = torch.tensor(in_n / out_n, dtype=torch.float32)  # TF computes the scale in fp32
        src = torch.arange(out_n, dtype=torch.float32) * scale
        lo = torch.floor(src).long()
        hi = torch.clamp(lo + 1, max=in_n - 1)
        lerp = (src - lo.float()).to(dt)
        return lo, hi, lerp

    ylo, yhi, yl = axis(in_h, out_h)
    xlo, xhi, xl = axis(in_w, out_w)
    top = x_nhwc[:, ylo]
    bot = x_nhwc[:, yhi]
    xl_ = xl.reshape(1, 1, out_w, 1)
    yl_ = yl.reshape(1, out_h, 1, 1)
    top_i = top[:, :, xlo] + (top[:, :, xhi] - top[:, :, xlo]) * xl_
    bot_i = bot[:, :, xlo] + (bot[:, :, xhi] - bot[:, :, xlo]) * xl_
    return top_i + (bot_i - top_i) * yl_


def compute_smooth_loss(pred: Tensor) -> Tensor:
    """SfMLearner_uni.py:205-219."""
    def gradient(p):
        d_dy = p[:, 1:, :, :] - p[:, :-1, :, :]
        d_dx = p[:, :, 1:, :] - p[:, :, :-1, :]
        return d_dx, d_dy
    beta = 0.25
    dx, dy = gradient(pred)
    dx2, dxdy = gradient(dx)
    dydx, dy2 = gradient(dy)
    gx = dx.abs().mean() + dy.abs().mean()
    hx = dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()
    return beta * gx + (1 - beta) * hx


def tf_minimum(a: Tensor, b: Tensor) -> Tensor:
    """``tf.minimum`` with TF's tie rule for the gradient: ties go to the first
    argument (torch.minimum would split them)."""
    return torch.where(a <= b, a, b)


def multi_scale_loss(tgt_image: Tensor, src_image_stack: Tensor, intrinsics: Tensor,
                     pred_depth: Sequence[Tensor], pred_poses: Tensor,
                     pred_exp_logits: Sequence[Tensor], pred_disp: Optional[Sequence[Tensor]] = None,
                     smooth_weight: float = 0.0, explain_reg_weight: float = 0.01,
                     alpha: float = 0.15, float_index: bool = False, return_parts: bool = False):
    """The loss loop of ``build_train_graph`` (SfMLearner_uni.py:75-176).

    tgt_image [B,H,W,3], src_image_stack [B,H,W,3*N] (already in [-1,1]);
    intrinsics [B,3,3] (the reference passes ``intrinsics[:,0]`` for every scale,
    :125,133); pred_depth[s] [B,h_s,w_s,1]; pred_poses [B,N,6]; pred_exp_logits[s]
    [B,h_s,w_s,3*N] (the per-pixel ``translations``); pred_disp[s] only for the
    smoothness term.  The number of scales is ``len(pred_depth)`` (quirk 1)."""
    bsz, height, width, _ = tgt_image.shape
    num_source = pred_poses.shape[1]
    if explain_reg_weight <= 0:
        raise NameError("curr_exp_logits is undefined when explain_reg_weight == 0 (SfMLearner_uni.py:127)")
    pixel_loss = tgt_image.new_zeros(())
    smooth_loss = tgt_image.new_zeros(())
    parts = []
    for s in range(len(pred_depth)):
        depth_s = resize_bilinear_tf1(pred_depth[s], height, width)  # :91
        if smooth_weight > 0:
            smooth_loss = smooth_loss + smooth_weight / (2 ** s) * compute_smooth_loss(pred_disp[s])
            smooth_loss = smooth_loss + explain_reg_weight / (2 ** s) * (
                compute_smooth_loss(pred_exp_logits[s]) + pred_exp_logits[s].abs().mean())  # :99-101
        curr_errors = []
        for i in range(num_source):
            trans = resize_bilinear_tf1(pred_exp_logits[s][..., 3 * i:3 * i + 3], height, width)  # :109-113
            src_i = src_image_stack[..., 3 * i:3 * i + 3]
            proj_image, _ = projective_inverse_warp(src_i, depth_s.squeeze(3), pred_poses[:, i, :], intrinsics,
                                                    add_trans=True, translations=trans,
                                                    float_index=float_index)  # :120
            proj_src_image, proj_src_mask = projective_warp(tgt_image, depth_s.squeeze(3), pred_poses[:, i, :],
                                                            intrinsics, add_trans=True, translations=trans)  # :128
            curr_error = alpha * (proj_image - tgt_image).abs()  # :137
            curr_proj_error = proj_src_mask * (proj_src_image - src_i).abs()  # :139
            curr_error = curr_error + (1 - alpha) * tf_ssim(proj_image, tgt_image, size=3, mean_metric=False)  # :142
            curr_errors.append(curr_error)
            curr_ssim_error = proj_src_mask * tf_ssim(proj_src_image, src_i, size=3, mean_metric=True)  # :144
            pixel_loss = pixel_loss + alpha * curr_proj_error.mean() + (1 - alpha) * curr_ssim_error.mean()  # :149
            if return_parts:
                parts.append(dict(s=s, i=i, proj_image=proj_image, proj_src_image=proj_src_image,
                                  proj_src_mask=proj_src_mask))
        pixel_loss = pixel_loss + 2.0 * tf_minimum(curr_errors[0], curr_errors[1]).mean()  # :168
    total = pixel_loss + smooth_loss  # exp_loss stays 0 (:76, :176)
    if return_parts:
        return total, pixel_loss, smooth_loss, parts
    return total


# ----------------------------------------------------------------------------
# ASPP dilated convolution
# ----------------------------------------------------------------------------
def dilated_conv3x3(x_nhwc: Tensor, w_hwio: Tensor, rate: int) -> Tensor:
    """``slim.conv2d(x, Cout, [3, 3], rate=rate)`` of ``disp_bottleneck_dense`` (nets.py:457, 458, 463)
    without the arg-scope's normaliser / activation (nets.py:653-656): NHWC x HWIO, stride 1, SAME
    padding.  For a 3x3 kernel with dilation ``rate`` TF's SAME pads ``rate`` zero pixels on every
    side (effective kernel 2*rate + 1, stride 1).                                      ASSUMPTION"""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    return F.conv2d(x, w, padding=rate, dilation=rate).permute(0, 2, 3, 1).contiguous()


def batch_norm_elu(x_nhwc: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Optional[Tensor] = None,
                   moving_var: Optional[Tensor] = None, decay: float = 0.9997, epsilon: float = 1e-5,
                   activation: bool = True) -> Tensor:
    """``slim.batch_norm(x, decay, epsilon, scale=True, fused=True, is_training=True, updates_collections=None)``
    followed by ``tf.nn.elu`` - what the arg-scope of ``disp_aspp_u_dense`` (nets.py:653-656 ->
    resnet/resnet_utils.py:232-251) appends to every ``slim.conv2d``.  Training mode: batch statistics over
    (B, H, W), population variance for the normalisation; the moving averages (updated in place) take the
    variance with Bessel's correction, as ``tf.nn.fused_batch_norm`` does.                      ASSUMPTION"""
    c = x_nhwc.shape[-1]
    flat = x_nhwc.reshape(-1, c)
    mean = flat.mean(dim=0)
    var = ((flat - mean) ** 2).mean(dim=0)
    # moving_averages.assign_moving_average(zero_debias=False): variable -= (variable - value) * cast(1.0 - decay)
    f = torch.tensor(1.0 - decay, dtype=x_nhwc.dtype)
    if moving_mean is not None:
        with torch.no_grad():
            moving_mean.sub_((moving_mean - mean.detach()) * f)
    if moving_var is not None:
        with torch.no_grad():
            m = flat.shape[0]
            moving_var.sub_((moving_var - var.detach() * (m / max(m - 1, 1))) * f)
    z = gamma * ((x_nhwc - mean) * torch.rsqrt(var + epsilon)) + beta
    return F.elu(z) if activation else z


def batch_norm_infer(x_nhwc: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Tensor, moving_var: Tensor,
                     epsilon: float = 1e-5, activation: bool = True) -> Tensor:
    """``slim.batch_norm(..., is_training=False)`` + ``tf.nn.elu``: the moving statistics instead of the batch's
    (FusedBatchNorm, is_training=False: y = (x - mean) * rsqrt(var + eps) * gamma + beta)."""
    z = gamma * ((x_nhwc - moving_mean) * torch.rsqrt(moving_var + epsilon)) + beta
    return F.elu(z) if activation else z


def conv1x1(x_nhwc: Tensor, w_hwio: Tensor) -> Tensor:
    """``slim.conv2d(x, Cout, [1, 1])`` without normaliser / activation (nets.py:455, 456, 461, 466, 470)."""
    return x_nhwc @ w_hwio[0, 0]


def disp_bottleneck_dense(params: dict, concat: Tensor, skip_feat: Optional[Tensor] = None, rates=(2, 4),
                          use_skip: bool = True, use_skip2: bool = True, skip2: Optional[Tensor] = None,
                          is_training: bool = True):
    """nets.py:451-471 op for op; every slim.conv2d = convolution (no bias) + batch norm [+ ELU unless
    activation_fn=None] through the arg-scope of nets.py:653-656.  ``params`` as in sfm_amd.aspp.  ``is_training=False``
    (the inference graphs, test_kitti_depth_con.py:19-65): every batch norm on its moving statistics."""
    def bn(y, q, activation=True):
        if is_training:
            return batch_norm_elu(y, q["gamma"], q["beta"], activation=activation)
        return batch_norm_infer(y, q["gamma"], q["beta"], q["moving_mean"], q["moving_var"], activation=activation)

    def sc(x, name, rate=None, activation=True):
        q = params[name]
        y = conv1x1(x, q["w"]) if rate is None else dilated_conv3x3(x, q["w"], rate)
        return bn(y, q, activation)
    h, w = concat.shape[1], concat.shape[2]
    upnet = sc(resize_bilinear_tf1(concat, 2 * h, 2 * w), "up", activation=False)
    supnet = sc(upnet, "sup")
    aspp_up = sc(upnet, "aspp_up", rate=rates[0])
    aspp_up2 = sc(torch.cat([supnet, aspp_up], dim=3), "aspp_up2", rate=rates[1])
    if use_skip:
        q = params["skip_bn"]
        skip = sc(bn(skip_feat, q), "skip")
    else:
        skip = sc(torch.cat([aspp_up, aspp_up2], dim=3), "skip", rate=rates[1] * 2)
    parts = [supnet, aspp_up, aspp_up2, skip]
    if use_skip2:
        parts.append(sc(resize_bilinear_tf1(skip2, 2 * h, 2 * w), "skip2"))
    return sc(torch.cat(parts, dim=3), "out", activation=False), upnet


DISP_SCALING = 10    # nets.py:11
MIN_DISP = 0.01      # nets.py:12


def disp_head(x_nhwc: Tensor, w_hwio: Tensor, bias: Tensor) -> Tensor:
    """``DISP_SCALING * slim.conv2d(x, 1, [3, 3], activation_fn=tf.sigmoid, normalizer_fn=None) + MIN_DISP``
    (nets.py:678, 683, 689): no normaliser, so slim.conv2d adds its bias (zeros-initialised variable); SAME padding."""
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2).contiguous(), w_hwio.permute(3, 2, 0, 1).contiguous(), bias=bias,
                 padding=1).permute(0, 2, 3, 1)
    return DISP_SCALING * torch.sigmoid(y) + MIN_DISP


def disp_aspp_u_dense_decoder(end_points: dict, params: dict, is_training: bool = True):
    """nets.py:666-692 after the ResNet: batch norm + ELU + 1x1 (no activation) on block4 (:672-673), five
    ``disp_bottleneck_dense`` levels (:675-686), three heads (:678, 683, 689).  Returns [disp4, disp3, disp2].
    ``end_points`` / ``params`` as in sfm_amd.aspp.disp_aspp_u_dense_decoder."""
    def bn(y, q, activation=True):
        if is_training:
            return batch_norm_elu(y, q["gamma"], q["beta"], activation=activation)
        return batch_norm_infer(y, q["gamma"], q["beta"], q["moving_mean"], q["moving_var"], activation=activation)
    lvl = lambda *a, **kw: disp_bottleneck_dense(*a, is_training=is_training, **kw)  # noqa: E731
    upnet = bn(end_points["block4"], params["stem_bn"])
    q = params["stem"]
    upnet = bn(conv1x1(upnet, q["w"]), q, activation=False)
    concat, _ = lvl(params["concat"], upnet, end_points["block3"], use_skip2=False)
    concat1, skip1 = lvl(params["concat1"], concat, end_points["block2"], use_skip2=False)
    xconcat2, skip2 = lvl(params["concat2"], concat1, end_points["block1"], skip2=skip1)
    disp2 = disp_head(xconcat2, params["disp2"]["w"], params["disp2"]["b"])
    xconcat3, skip3 = lvl(params["concat3"], xconcat2, end_points["root_block"], rates=(3, 6), skip2=skip2)
    disp3 = disp_head(xconcat3, params["disp3"]["w"], params["disp3"]["b"])
    xconcat4, _ = lvl(params["concat4"], xconcat3, None, rates=(3, 6), use_skip=False, skip2=skip3)
    disp4 = disp_head(xconcat4, params["disp4"]["w"], params["disp4"]["b"])
    return [disp4, disp3, disp2]


# ----------------------------------------------------------------------------
# data-parallel decomposition of the pixel loss
# ----------------------------------------------------------------------------
def loss_partials(tgt_image, src_image_stack, intrinsics, pred_depth, pred_poses, pred_exp_logits,
                  alpha: float = 0.15) -> Tensor:
    """The S*(3N+1) fp64 sums one data-parallel rank contributes: per scale and source
    (sum m|Sp-src|, sum m, sum ssim(Sp,src)) and per scale sum min(e_0,e_1).  These are
    exactly the quantities behind every ``reduce_mean`` of SfMLearner_uni.py:149,168 and
    utils.py:574; summed over ranks they give the single-device loss (tested with gloo)."""
    bsz, height, width, _ = tgt_image.shape
    num_source = pred_poses.shape[1]
    out = []
    for s in range(len(pred_depth)):
        depth_s = resize_bilinear_tf1(pred_depth[s], height, width)
        errs = []
        for i in range(num_source):
            trans = resize_bilinear_tf1(pred_exp_logits[s][..., 3 * i:3 * i + 3], height, width)
            src_i = src_image_stack[..., 3 * i:3 * i + 3]
            proj_image, _ = projective_inverse_warp(src_i, depth_s.squeeze(3), pred_poses[:, i, :], intrinsics,
                                                    add_trans=True, translations=trans)
            sp, m = projective_warp(tgt_image, depth_s.squeeze(3), pred_poses[:, i, :], intrinsics, add_trans=True,
                                    translations=trans)
            errs.append(alpha * (proj_image - tgt_image).abs()
                        + (1 - alpha) * tf_ssim(proj_image, tgt_image, size=3, mean_metric=False))
            out += [(m * (sp - src_i).abs()).double().sum(), m.double().sum(), ssim_value(sp, src_i).double().sum()]
        out.append(tf_minimum(errs[0], errs[1]).double().sum())
    return torch.stack(out)


def loss_from_partials(partials: Tensor, num_scales: int, num_source: int, n_elem_global: int,
                       alpha: float = 0.15) -> Tensor:
    """pixel_loss of SfMLearner_uni.py:149,168 from the (globally summed) partials."""
    p = partials.reshape(num_scales, 3 * num_source + 1).double()
    n = float(n_elem_global)
    loss = torch.zeros((), dtype=torch.float64)
    for s in range(num_scales):
        for i in range(num_source):
            a_sum, m_sum, v_sum = p[s, 3 * i], p[s, 3 * i + 1], p[s, 3 * i + 2]
            sc = torch.clamp((1 - (v_sum / n).float()) / 2.0, 0, 1).double()  # utils.py:573-575 on the global mean
            loss = loss + alpha * a_sum / n + (1 - alpha) * sc * m_sum / n
        loss = loss + 2.0 * p[s, 3 * num_source] / n
    return loss
