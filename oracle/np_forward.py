"""Independent per-pixel NumPy restatement of the two warps (forward only).
TEST INFRASTRUCTURE ONLY - see the header of oracle/ref_graph.py for who may import it.

Written as scalar loops over pixels in np.float32 arithmetic, i.e. in a different
style from the tensorised torch restatement, so that a transcription slip in one of
them shows up as a disagreement (tests/test_oracle.py).  Sizes: tiny images only.

Reference lines followed (paths relative to the reference repo):
  euler2mat utils.py:27-75, pose_vec2mat 77-96, meshgrid 146-170, pixel2cam 98-122,
  cam2pixel 124-144, bilinear_sampler 241-340, bilinear_project 342-502.
PARITY UNPINNED by the reference itself (no TensorFlow here, no fixtures there).
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


def _dot(a, b):
    """k-ordered dot product, one rounding per multiply and per add."""
    acc = f32(a[0]) * f32(b[0])
    for k in range(1, len(a)):
        acc = f32(acc + f32(f32(a[k]) * f32(b[k])))
    return f32(acc)


def _matmul(a, b):
    n, kk = a.shape
    m = b.shape[1]
    o = np.zeros((n, m), dtype=f32)
    for i in range(n):
        for j in range(m):
            o[i, j] = _dot(a[i, :], b[:, j])
    return o


def _lu_inverse(a):
    """Partial-pivot LU inverse in fp32 (the canonical order of ref_graph.matrix_inverse_lu)."""
    n = a.shape[0]
    lu = a.astype(f32).copy()
    rhs = np.eye(n, dtype=f32)
    for k in range(n):
        p = k + int(np.argmax(np.abs(lu[k:, k])))
        if p != k:
            lu[[k, p]] = lu[[p, k]]
            rhs[[k, p]] = rhs[[p, k]]
        for i in range(k + 1, n):
            l = f32(lu[i, k] / lu[k, k])
            lu[i, k] = l
            for j in range(k + 1, n):
                lu[i, j] = f32(lu[i, j] - f32(l * lu[k, j]))
    y = np.zeros((n, n), dtype=f32)
    for i in range(n):
        for c in range(n):
            acc = rhs[i, c]
            for j in range(i):
                acc = f32(acc - f32(lu[i, j] * y[j, c]))
            y[i, c] = acc
    x = np.zeros((n, n), dtype=f32)
    for i in range(n - 1, -1, -1):
        for c in range(n):
            acc = y[i, c]
            for j in range(i + 1, n):
                acc = f32(acc - f32(lu[i, j] * x[j, c]))
            x[i, c] = f32(acc / lu[i, i])
    return x


def pose_matrix(vec):
    """pose_vec2mat for one item: (tx,ty,tz,rx,ry,rz) -> 4x4, R = (X.Y).Z, angles clipped to +-pi."""
    pi = f32(math.pi)
    rx, ry, rz = (min(max(f32(v), -pi), pi) for v in vec[3:6])
    sx, cx = f32(math.sin(float(rx))), f32(math.cos(float(rx)))
    sy, cy = f32(math.sin(float(ry))), f32(math.cos(float(ry)))
    sz, cz = f32(math.sin(float(rz))), f32(math.cos(float(rz)))
    X = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=f32)
    Y = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=f32)
    Z = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=f32)
    R = _matmul(_matmul(X, Y), Z)
    T = np.eye(4, dtype=f32)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(vec[:3], dtype=f32)
    return T


def grid_value(i, n):
    """meshgrid coordinate i of n: ((linspace(-1,1,n)[i] + 1) * 0.5) * (n-1), fp32."""
    if n == 1:
        lin = f32(-1.0)
    else:
        step = f32(f32(2.0) / f32(n - 1))
        lin = f32(f32(-1.0) + f32(step * f32(i)))
    return f32(f32(f32(lin + f32(1.0)) * f32(0.5)) * f32(n - 1))


def project_coords(depth, pose, K, inverse_pose=False, trans=None):
    """depth [B,H,W], pose [B,6], K [B,3,3], trans [B,H,W,3] or None -> coords [B,H,W,2], z [B,H,W,1]."""
    B, H, W = depth.shape
    coords = np.zeros((B, H, W, 2), dtype=f32)
    z = np.zeros((B, H, W, 1), dtype=f32)
    for b in range(B):
        T = pose_matrix(pose[b])
        if inverse_pose:
            T = _lu_inverse(T)
        K4 = np.eye(4, dtype=f32)
        K4[:3, :3] = K[b]
        P = _matmul(K4, T)
        Ki = _lu_inverse(K[b])
        for v in range(H):
            gy = grid_value(v, H)
            for u in range(W):
                gx = grid_value(u, W)
                pix = (gx, gy, f32(1.0))
                cam = [f32(_dot(Ki[r], pix) * f32(depth[b, v, u])) for r in range(3)]
                if trans is not None:
                    cam = [f32(cam[r] + f32(f32(trans[b, v, u, r]) * cam[r])) for r in range(3)]
                cam4 = (cam[0], cam[1], cam[2], f32(1.0))
                q = [_dot(P[r], cam4) for r in range(3)]
                den = f32(q[2] + f32(1e-10))
                coords[b, v, u, 0] = f32(q[0] / den)
                coords[b, v, u, 1] = f32(q[1] / den)
                z[b, v, u, 0] = q[2]
    return coords, z


def _corners(x, y, Hs, Ws):
    x0 = f32(math.floor(x)); x1 = f32(x0 + f32(1)); y0 = f32(math.floor(y)); y1 = f32(y0 + f32(1))
    cl = lambda v, hi: f32(min(max(v, f32(0)), f32(hi)))
    return cl(x0, Ws - 1), cl(x1, Ws - 1), cl(y0, Hs - 1), cl(y1, Hs - 1)


def bilinear_sampler(img, coords):
    B, Hs, Ws, C = img.shape
    _, Ht, Wt, _ = coords.shape
    out = np.zeros((B, Ht, Wt, C), dtype=f32)
    mask = np.zeros((B, Ht, Wt, 1), dtype=f32)
    idx = np.zeros((B, Ht, Wt, 4), dtype=np.int32)
    for b in range(B):
        for v in range(Ht):
            for u in range(Wt):
                x, y = f32(coords[b, v, u, 0]), f32(coords[b, v, u, 1])
                x0, x1, y0, y1 = _corners(x, y, Hs, Ws)
                idx[b, v, u] = (int(x0), int(x1), int(y0), int(y1))
                wx0, wx1, wy0, wy1 = f32(x1 - x), f32(x - x0), f32(y1 - y), f32(y - y0)
                w00, w01, w10, w11 = f32(wx0 * wy0), f32(wx0 * wy1), f32(wx1 * wy0), f32(wx1 * wy1)
                for c in range(C):
                    a = f32(w00 * img[b, int(y0), int(x0), c])
                    a = f32(a + f32(w01 * img[b, int(y1), int(x0), c]))
                    a = f32(a + f32(w10 * img[b, int(y0), int(x1), c]))
                    a = f32(a + f32(w11 * img[b, int(y1), int(x1), c]))
                    out[b, v, u, c] = a
                oob = (x < -1) or (x > Ws) or (y < -1) or (y > Hs)
                mask[b, v, u, 0] = 0.0 if oob else 1.0
    return out, mask, idx


def bilinear_project(img, coords, z=None):
    """Forward splat; accumulation in fp64 (the sums are order-dependent in fp32), rounded at the end."""
    B, H, W, C = img.shape
    S = np.zeros((B, H, W, C), dtype=np.float64)
    D = np.zeros((B, H, W), dtype=np.float64)
    for b in range(B):
        for v in range(H):
            for u in range(W):
                x, y = f32(coords[b, v, u, 0]), f32(coords[b, v, u, 1])
                x0, x1, y0, y1 = _corners(x, y, H, W)
                dx0, dx1, dy0, dy1 = f32(x - x0), f32(x1 - x), f32(y - y0), f32(y1 - y)
                dw = f32(1.0)
                if z is not None:
                    zq = f32(f32(z[b, v, u, 0]) / f32(4))
                    dw = f32(np.exp(-f32(min(max(zq, f32(0)), f32(10)))))
                for (dx, dy, xx, yy) in ((dx0, dy0, x0, y0), (dx0, dy1, x0, y1), (dx1, dy0, x1, y0), (dx1, dy1, x1, y1)):
                    g = f32(np.exp(f32(-f32(f32(dx * dx) + f32(dy * dy)) / f32(2))))
                    g = f32(dw * g) if z is not None else g
                    D[b, int(yy), int(xx)] += float(g)
                    for c in range(C):
                        S[b, int(yy), int(xx), c] += float(f32(img[b, v, u, c] * g))
    out = (S / (D[..., None] + 1e-10)).astype(f32)
    nz = (out != 0).astype(np.float64)
    mask = np.zeros_like(out)
    for v in range(H):
        for u in range(W):
            ys, xs = slice(max(v - 1, 0), min(v + 2, H)), slice(max(u - 1, 0), min(u + 2, W))
            mask[:, v, u, :] = nz[:, ys, xs, :].mean(axis=(1, 2))
    return out, mask
