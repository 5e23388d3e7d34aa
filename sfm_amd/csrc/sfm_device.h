// Device-side building blocks shared by every kernel of the view-synthesis path.
// gfx950 only.  This translation unit family is compiled with -ffp-contract=off:
// the coordinate arithmetic below must keep the reference's fp32 rounding sequence
// (one rounding per multiply and per add, k-ordered dot products) so that the
// projected integer pixel indices are bit-exact against the canonical restatement
// (oracle/ref_graph.py).  Where a fused multiply-add is wanted it is written as
// __builtin_fmaf explicitly.
//
// Reference arithmetic followed here (paths relative to the reference repo):
//   meshgrid    utils.py:146-170     pixel2cam  utils.py:98-122
//   cam2pixel   utils.py:124-144     corners    utils.py:274-289 == 376-388
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SFM_PARAM_STRIDE 32  // floats per (source, batch item): Kinv[9], P[12], pad

namespace sfm {

constexpr int kBlock = 256;  // 4 waves of 64

struct __attribute__((packed, aligned(4))) f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 ld3(const float* __restrict__ p) { return *reinterpret_cast<const f3*>(p); }
__device__ __forceinline__ void st3(float* __restrict__ p, float a, float b, float c) {
  f3 v{a, b, c};
  *reinterpret_cast<f3*>(p) = v;
}

// Per-(source, batch item) constants, wave-uniform (scalar loads).
struct Pose {
  float ki[9];  // inverse intrinsics, row-major
  float p[12];  // rows 0..2 of [[K,0],[0,1]] * T, row-major 3x4
};

__device__ __forceinline__ Pose load_pose(const float* __restrict__ prm) {
  Pose o;
#pragma unroll
  for (int k = 0; k < 9; ++k) o.ki[k] = prm[k];
#pragma unroll
  for (int k = 0; k < 12; ++k) o.p[k] = prm[9 + k];
  return o;
}

// meshgrid (utils.py:157-163): ((linspace(-1,1,n)[i] + 1) * 0.5) * (n-1), with
// linspace[i] = -1 + i*step, step = fp32(2/(n-1)) computed on the host.
__device__ __forceinline__ float grid_coord(int i, float step, float nm1) {
  float lin = -1.0f + step * (float)i;
  return ((lin + 1.0f) * 0.5f) * nm1;
}

struct Geom {
  float x, y, z;     // projected pixel coordinates and depth in the other view
  float den;         // z + 1e-10
  float c0, c1, c2;  // camera-frame point after the optional per-pixel rescale
  float r0, r1, r2;  // Kinv * (gx, gy, 1)
};

// pixel2cam + cam2pixel for one pixel.  add_trans: c += t (.) c  (utils.py:116).
__device__ __forceinline__ Geom geom_fwd(const Pose& ps, float gx, float gy, float d, bool add_trans, float t0,
                                         float t1, float t2) {
  Geom g;
  g.r0 = (ps.ki[0] * gx + ps.ki[1] * gy) + ps.ki[2] * 1.0f;
  g.r1 = (ps.ki[3] * gx + ps.ki[4] * gy) + ps.ki[5] * 1.0f;
  g.r2 = (ps.ki[6] * gx + ps.ki[7] * gy) + ps.ki[8] * 1.0f;
  float c0 = g.r0 * d, c1 = g.r1 * d, c2 = g.r2 * d;
  if (add_trans) {
    c0 = c0 + t0 * c0;
    c1 = c1 + t1 * c1;
    c2 = c2 + t2 * c2;
  }
  g.c0 = c0;
  g.c1 = c1;
  g.c2 = c2;
  float q0 = ((ps.p[0] * c0 + ps.p[1] * c1) + ps.p[2] * c2) + ps.p[3] * 1.0f;
  float q1 = ((ps.p[4] * c0 + ps.p[5] * c1) + ps.p[6] * c2) + ps.p[7] * 1.0f;
  float q2 = ((ps.p[8] * c0 + ps.p[9] * c1) + ps.p[10] * c2) + ps.p[11] * 1.0f;
  g.den = q2 + 1e-10f;
  g.x = q0 / g.den;
  g.y = q1 / g.den;
  g.z = q2;
  return g;
}

// Chain rule of geom_fwd.  (gx_, gy_, gz_) = dL/d(x, y, z).  Accumulates the 12
// entries of dL/dP into dp[], returns dL/d depth and dL/d translations.
__device__ __forceinline__ void geom_bwd(const Pose& ps, const Geom& g, float d, bool add_trans, float t0, float t1,
                                         float t2, float gx_, float gy_, float gz_, float& dd, float& dt0,
                                         float& dt1, float& dt2, float* __restrict__ dp) {
  float inv = 1.0f / g.den;
  float dq0 = gx_ * inv;
  float dq1 = gy_ * inv;
  float dq2 = gz_ - (gx_ * g.x + gy_ * g.y) * inv;
  dp[0] += dq0 * g.c0; dp[1] += dq0 * g.c1; dp[2] += dq0 * g.c2; dp[3] += dq0;
  dp[4] += dq1 * g.c0; dp[5] += dq1 * g.c1; dp[6] += dq1 * g.c2; dp[7] += dq1;
  dp[8] += dq2 * g.c0; dp[9] += dq2 * g.c1; dp[10] += dq2 * g.c2; dp[11] += dq2;
  float dc0 = (ps.p[0] * dq0 + ps.p[4] * dq1) + ps.p[8] * dq2;
  float dc1 = (ps.p[1] * dq0 + ps.p[5] * dq1) + ps.p[9] * dq2;
  float dc2 = (ps.p[2] * dq0 + ps.p[6] * dq1) + ps.p[10] * dq2;
  if (add_trans) {
    dt0 = dc0 * (g.r0 * d);
    dt1 = dc1 * (g.r1 * d);
    dt2 = dc2 * (g.r2 * d);
    dd = (dc0 * (1.0f + t0) * g.r0 + dc1 * (1.0f + t1) * g.r1) + dc2 * (1.0f + t2) * g.r2;
  } else {
    dt0 = dt1 = dt2 = 0.0f;
    dd = (dc0 * g.r0 + dc1 * g.r1) + dc2 * g.r2;
  }
}

// floor / +1 / clip (utils.py:274-289).  fminf/fmaxf drop NaNs, so the integer
// corners are always inside the image whatever the coordinates are.
struct Corners {
  float fx0, fx1, fy0, fy1;  // clipped, as floats
  int x0, x1, y0, y1;
};

__device__ __forceinline__ Corners corners(float x, float y, float xmax, float ymax) {
  Corners c;
  float x0 = floorf(x), y0 = floorf(y);
  float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
  c.fx0 = fminf(fmaxf(x0, 0.0f), xmax);
  c.fx1 = fminf(fmaxf(x1, 0.0f), xmax);
  c.fy0 = fminf(fmaxf(y0, 0.0f), ymax);
  c.fy1 = fminf(fmaxf(y1, 0.0f), ymax);
  c.x0 = (int)c.fx0;
  c.x1 = (int)c.fx1;
  c.y0 = (int)c.fy0;
  c.y1 = (int)c.fy1;
  return c;
}

// Gaussian corner weights of the forward splat (utils.py:396-399, 431-442).
// Order: g[0]=00 -> (y0,x0), g[1]=01 -> (y1,x0), g[2]=10 -> (y0,x1), g[3]=11 -> (y1,x1).
struct SplatW {
  float dx0, dx1, dy0, dy1;
  float g[4];
  float zin;  // 1 where the depth weight has a gradient (0 <= z/4 <= 10)
};

__device__ __forceinline__ SplatW splat_weights(float x, float y, const Corners& c, bool use_depth, float z) {
  SplatW s;
  s.dx0 = x - c.fx0;
  s.dx1 = c.fx1 - x;
  s.dy0 = y - c.fy0;
  s.dy1 = c.fy1 - y;
  float ax0 = s.dx0 * s.dx0, ax1 = s.dx1 * s.dx1, ay0 = s.dy0 * s.dy0, ay1 = s.dy1 * s.dy1;
  s.g[0] = expf(-(ax0 + ay0) / 2.0f);
  s.g[1] = expf(-(ax0 + ay1) / 2.0f);
  s.g[2] = expf(-(ax1 + ay0) / 2.0f);
  s.g[3] = expf(-(ax1 + ay1) / 2.0f);
  s.zin = 0.0f;
  if (use_depth) {
    float zq = z / 4.0f;
    float dw = expf(-fminf(fmaxf(zq, 0.0f), 10.0f));
    s.zin = (zq >= 0.0f && zq <= 10.0f) ? 1.0f : 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.g[k] = dw * s.g[k];
  }
  return s;
}

// tf.image.resize_bilinear, TF-1 defaults (align_corners=False, no half-pixel centres), resize_bilinear_op.cc:
//   scale = in / (float)out;  src = dst * scale;  lo = floor(src);  hi = min(lo + 1, in - 1);  lerp = src - lo
//   top = tl + (tr - tl) * xl;  bot = bl + (br - bl) * xl;  out = top + (bot - top) * yl
// Shared by the stand-alone resize kernels (pyramid.hip) and the fused loss kernels that read the prediction pyramid
// directly (loss.hip): one arithmetic, so the two ways of feeding the warps agree bit for bit.
struct Lerp {
  int lo, hi;
  float l;
};
__device__ __forceinline__ Lerp lerp_of(int dst, float scale, int in_n) {
  const float src = (float)dst * scale;
  const float f = floorf(src);
  Lerp o;
  o.lo = min((int)f, in_n - 1);
  o.hi = min(o.lo + 1, in_n - 1);
  o.l = src - f;
  return o;
}
__device__ __forceinline__ float lerp4(float tl, float tr, float bl, float br, float xl, float yl) {
  const float top = tl + (tr - tl) * xl;
  const float bot = bl + (br - bl) * xl;
  return top + (bot - top) * yl;
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks L and L + 8 share an XCD and its L2;
// MI355X_MICROARCH.md "Workgroup dispatch").  xcd_contiguous(L, T) renumbers a 1-D grid of T blocks so
// that each XCD works on one contiguous range of the renumbered ids: neighbouring tiles then share an L2
// and their overlapping halos are fetched from HBM once.  A bijection on [0, T); speed only, never
// correctness.
__device__ __forceinline__ int xcd_contiguous(int L, int T) {
  const int per = T >> 3, rem = T & 7, x = L & 7;
  return x * per + min(x, rem) + (L >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Wave sum on the DPP network (6 VALU instructions with a cross-lane operand each; the shuffle tree above goes through
// ds_bpermute: an LDS round trip per step).  Lane 63 ends up with the total.  Order: quads, rows of 16 by rotation,
// then row 0 -> 1, 2 -> 3 and rows {0,1} -> {2,3} by broadcast; fixed, so results are reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float x) {  // 0 where the source lane is invalid or the row is masked off
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp_lane63(float v) {
  v += dpp_mov<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x124, 0xf>(v);  // row_ror:4
  v += dpp_mov<0x128, 0xf>(v);  // row_ror:8   -> every lane holds its row's sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 = total
  return v;
}

// Sum NV per-thread values over a 256-thread block; thread k < NV of wave 0 ends
// up holding the k-th total and stores it to out[k].  red must hold 4*NV floats.
template <int NV>
__device__ __forceinline__ void block_sum_store(float* __restrict__ vals, float* __restrict__ red,
                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float s = wave_sum_dpp_lane63(vals[k]);
    if (lane == 63) red[wave * NV + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = (red[threadIdx.x] + red[NV + threadIdx.x]) + (red[2 * NV + threadIdx.x] + red[3 * NV + threadIdx.x]);
    out[threadIdx.x] = s;
  }
}

// 3x3 Gaussian window of tf_ssim (utils.py:504-519, size 3, sigma 1.5), fp32:
// exp(-(i*i+j*j)/4.5) normalised by the sum of the nine taps.
struct SsimWin {
  float w[9];
};
__device__ __forceinline__ SsimWin ssim_window() {
  const float e1 = 0.800737402916808f;   // exp(-1/4.5)
  const float e2 = 0.6411803884299546f;  // exp(-2/4.5)
  const float sum = (((((((e2 + e1) + e2) + e1) + 1.0f) + e1) + e2) + e1) + e2;
  SsimWin o;
  o.w[0] = e2 / sum; o.w[1] = e1 / sum; o.w[2] = e2 / sum;
  o.w[3] = e1 / sum; o.w[4] = 1.0f / sum; o.w[5] = e1 / sum;
  o.w[6] = e2 / sum; o.w[7] = e1 / sum; o.w[8] = e2 / sum;
  return o;
}

constexpr float kSsimC1 = 0.0001f;  // (0.01*1)^2
constexpr float kSsimC2 = 0.0009f;  // (0.03*1)^2

// SSIM value from the five windowed raw moments, and (optionally) its partial
// derivatives with respect to m1 (total), m11 and m12 (image 1 differentiated).
__device__ __forceinline__ float ssim_from_moments(float m1, float m2, float m11, float m22, float m12) {
  float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
  float s11 = m11 - mu1_sq, s22 = m22 - mu2_sq, s12 = m12 - mu12;
  return ((2.0f * mu12 + kSsimC1) * (2.0f * s12 + kSsimC2)) / ((mu1_sq + mu2_sq + kSsimC1) * (s11 + s22 + kSsimC2));
}

__device__ __forceinline__ float ssim_with_grad(float m1, float m2, float m11, float m22, float m12, float& a,
                                                float& b, float& c) {
  float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
  float s11 = m11 - mu1_sq, s22 = m22 - mu2_sq, s12 = m12 - mu12;
  float A1 = 2.0f * mu12 + kSsimC1, A2 = 2.0f * s12 + kSsimC2;
  float B1 = mu1_sq + mu2_sq + kSsimC1, B2 = s11 + s22 + kSsimC2;
  float iB = 1.0f / (B1 * B2);
  float v = (A1 * A2) * iB;
  float dv_ds12 = 2.0f * A1 * iB;
  float dv_ds11 = -v / B2;
  float dv_dm1 = (2.0f * m2 * A2) * iB - v * (2.0f * m1) / B1;
  a = dv_dm1 + dv_ds11 * (-2.0f * m1) + dv_ds12 * (-m2);
  b = dv_ds11;
  c = dv_ds12;
  return v;
}

}  // namespace sfm
