// The two view-synthesis warps as stand-alone operators (one launch per call):
//   gather  : bilinear_sampler            utils.py:241-340   (+ projective_inverse_warp 207-239)
//   splat   : bilinear_project            utils.py:342-502   (+ projective_warp 172-205)
// Each kernel exists in two flavours selected at compile time:
//   GEOM = true   coordinates come from depth / pose / intrinsics, computed in registers
//                 (meshgrid -> pixel2cam -> cam2pixel never touch memory)
//   GEOM = false  coordinates are an input tensor (the reference's bilinear_* signatures)
// and CT = 3 (12-byte NHWC pixel vectors, one dwordx3 per tap) or CT = 0 (any channel
// count, scalar loop).  One thread per target pixel, blockIdx.y = batch item so the
// 21 pose/intrinsics constants are scalar loads.  HBM-bound: no LDS staging is needed
// here because every tensor element is touched once and taps hit L2.
#include <cstdint>
#include <cstdlib>
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {

struct CoordSrc {
  const float* coords;  // [B,Ht,Wt,2]        (GEOM = false)
  const float* zin;     // [B,Ht,Wt] or null  (GEOM = false; projected depth for the splat)
  const float* depth;   // [B,H,W]            (GEOM = true)
  const float* trans;   // [B,H,W,3] or null  (GEOM = true, add_trans)
  const float* params;  // [B,SFM_PARAM_STRIDE]
  float step_x, step_y;
};

struct Pix {
  float x, y, z;
  Geom g;
  float d, t0, t1, t2;
};

template <bool GEOM>
__device__ __forceinline__ Pix fetch(const CoordSrc& cs, const Pose& ps, int b, int p, int H, int W) {
  Pix o;
  const long gi = (long)b * H * W + p;
  if constexpr (GEOM) {
    int v = p / W, u = p - v * W;
    float gx = grid_coord(u, cs.step_x, (float)(W - 1));
    float gy = grid_coord(v, cs.step_y, (float)(H - 1));
    o.d = cs.depth[gi];
    o.t0 = o.t1 = o.t2 = 0.0f;
    if (cs.trans) {
      f3 t = ld3(cs.trans + gi * 3);
      o.t0 = t.x; o.t1 = t.y; o.t2 = t.z;
    }
    o.g = geom_fwd(ps, gx, gy, o.d, cs.trans != nullptr, o.t0, o.t1, o.t2);
    o.x = o.g.x; o.y = o.g.y; o.z = o.g.z;
  } else {
    float2 xy = *reinterpret_cast<const float2*>(cs.coords + gi * 2);
    o.x = xy.x; o.y = xy.y;
    o.z = cs.zin ? cs.zin[gi] : 0.0f;
  }
  return o;
}

// ---------------------------------------------------------------- coords only
__global__ __launch_bounds__(kBlock) void project_coords_kernel(CoordSrc cs, int H, int W, float* __restrict__ coords,
                                                                float* __restrict__ zout) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  Pose ps = load_pose(cs.params + (long)b * SFM_PARAM_STRIDE);
  if (p >= H * W) return;
  Pix px = fetch<true>(cs, ps, b, p, H, W);
  const long gi = (long)b * H * W + p;
  *reinterpret_cast<float2*>(coords + gi * 2) = make_float2(px.x, px.y);
  if (zout) zout[gi] = px.z;
}

// ---------------------------------------------------------------- gather forward
template <bool GEOM, int CT>
__global__ __launch_bounds__(kBlock) void sampler_fwd_kernel(CoordSrc cs, const float* __restrict__ img, int Hs,
                                                             int Ws, int C, int Ht, int Wt, float* __restrict__ out,
                                                             float* __restrict__ mask, int* __restrict__ idx_dump) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  Pose ps;
  if constexpr (GEOM) ps = load_pose(cs.params + (long)b * SFM_PARAM_STRIDE);
  if (p >= Ht * Wt) return;
  Pix px = fetch<GEOM>(cs, ps, b, p, Ht, Wt);
  const float xmax = (float)(Ws - 1), ymax = (float)(Hs - 1);
  Corners c = corners(px.x, px.y, xmax, ymax);
  float wx0 = c.fx1 - px.x, wx1 = px.x - c.fx0, wy0 = c.fy1 - px.y, wy1 = px.y - c.fy0;  // utils.py:297-300
  float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
  const float* ib = img + (long)b * Hs * Ws * C;
  const int i00 = c.y0 * Ws + c.x0, i01 = c.y1 * Ws + c.x0, i10 = c.y0 * Ws + c.x1, i11 = c.y1 * Ws + c.x1;
  const long gi = (long)b * Ht * Wt + p;
  float* o = out + gi * C;
  if constexpr (CT == 3) {
    f3 a = ld3(ib + i00 * 3), bb = ld3(ib + i01 * 3), cc = ld3(ib + i10 * 3), dd = ld3(ib + i11 * 3);
    st3(o, ((w00 * a.x + w01 * bb.x) + w10 * cc.x) + w11 * dd.x, ((w00 * a.y + w01 * bb.y) + w10 * cc.y) + w11 * dd.y,
        ((w00 * a.z + w01 * bb.z) + w10 * cc.z) + w11 * dd.z);
  } else {
    for (int ch = 0; ch < C; ++ch)
      o[ch] = ((w00 * ib[i00 * C + ch] + w01 * ib[i01 * C + ch]) + w10 * ib[i10 * C + ch]) + w11 * ib[i11 * C + ch];
  }
  const bool oob = (px.x < -1.0f) || (px.x > xmax + 1.0f) || (px.y < -1.0f) || (px.y > ymax + 1.0f);  // :283-284
  mask[gi] = 1.0f - (oob ? 1.0f : 0.0f);
  if (idx_dump) *reinterpret_cast<int4*>(idx_dump + gi * 4) = make_int4(c.x0, c.x1, c.y0, c.y1);
}

// ---------------------------------------------------------------- gather backward
// PPT pixels per thread so the 12 dL/dP partials are reduced once per 1024 pixels.
constexpr int kPPT = 4;

template <bool GEOM, int CT>
__global__ __launch_bounds__(kBlock) void sampler_bwd_kernel(CoordSrc cs, const float* __restrict__ img,
                                                             const float* __restrict__ gout, int Hs, int Ws, int C,
                                                             int Ht, int Wt, float* __restrict__ d_coords,
                                                             float* __restrict__ d_depth, float* __restrict__ d_trans,
                                                             float* __restrict__ partials, float* __restrict__ d_img) {
  __shared__ float red[4 * 12];
  const int b = blockIdx.y;
  Pose ps;
  if constexpr (GEOM) ps = load_pose(cs.params + (long)b * SFM_PARAM_STRIDE);
  float dp[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) dp[k] = 0.0f;
  const float xmax = (float)(Ws - 1), ymax = (float)(Hs - 1);
  const float* ib = img + (long)b * Hs * Ws * C;
  float* dib = d_img ? d_img + (long)b * Hs * Ws * C : nullptr;
  // (whole waves run every iteration - kBlock * kPPT pixels per block, lanes beyond the image only skip their
  // loads and stores - so that the lane exchange below always has its neighbour)
  for (int it = 0; it < kPPT; ++it) {
    const int p_raw = (blockIdx.x * kPPT + it) * kBlock + threadIdx.x;
    if ((blockIdx.x * kPPT + it) * kBlock >= Ht * Wt) break;  // block-uniform
    const bool live = p_raw < Ht * Wt;
    const int p = live ? p_raw : Ht * Wt - 1;
    Pix px = fetch<GEOM>(cs, ps, b, p, Ht, Wt);
    Corners c = corners(px.x, px.y, xmax, ymax);
    float wx0 = c.fx1 - px.x, wx1 = px.x - c.fx0, wy0 = c.fy1 - px.y, wy1 = px.y - c.fy0;
    const int i00 = c.y0 * Ws + c.x0, i01 = c.y1 * Ws + c.x0, i10 = c.y0 * Ws + c.x1, i11 = c.y1 * Ws + c.x1;
    const long gi = (long)b * Ht * Wt + p;
    float gx_ = 0.0f, gy_ = 0.0f;
    // d_img: wave-aggregated scatter.  For a smooth warp lane l's right-hand taps (x1, y0 / y1) are lane l + 1's
    // left-hand ones: a lane hands the contributions of its right column to the next lane when the addresses agree
    // (one DPP wave shift each way for the test, one per value) and issues atomics only for what it keeps - half the
    // atomics of the per-lane form on such warps, the same sums in any case.
    bool keep10 = live, keep11 = live, take00 = false, take01 = false;
#ifndef SFM_NO_DIMG_MERGE
    if (dib) {
      const int lane = threadIdx.x & 63;
      const int m00 = live ? i00 : -1, m01 = live ? i01 : -1, m10 = live ? i10 : -2, m11 = live ? i11 : -2;
      const int prev10 = __builtin_amdgcn_update_dpp(-3, m10, 0x138, 0xf, 0xf, false);  // wave_shr:1 - lane l reads l - 1
      const int prev11 = __builtin_amdgcn_update_dpp(-3, m11, 0x138, 0xf, 0xf, false);
      const int next00 = __builtin_amdgcn_update_dpp(-4, m00, 0x130, 0xf, 0xf, false);  // wave_shl:1 - lane l reads l + 1
      const int next01 = __builtin_amdgcn_update_dpp(-4, m01, 0x130, 0xf, 0xf, false);
      take00 = lane > 0 && prev10 == m00;
      take01 = lane > 0 && prev11 == m01;
      keep10 = live && !(lane < 63 && next00 == m10);
      keep11 = live && !(lane < 63 && next01 == m11);
    }
#endif
    for (int ch = 0; ch < (CT == 3 ? 3 : C); ++ch) {
      float g = live ? gout[gi * C + ch] : 0.0f;
      float a = ib[i00 * C + ch], bb = ib[i01 * C + ch], cc = ib[i10 * C + ch], dd = ib[i11 * C + ch];
      gx_ += g * (wy0 * (cc - a) + wy1 * (dd - bb));
      gy_ += g * (wx0 * (bb - a) + wx1 * (dd - cc));
      if (dib) {
        const float v10 = (wx1 * wy0) * g, v11 = (wx1 * wy1) * g;
        float v00 = (wx0 * wy0) * g, v01 = (wx0 * wy1) * g;
        const float p10 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v10), 0x138, 0xf, 0xf, false));
        const float p11 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v11), 0x138, 0xf, 0xf, false));
        if (take00) v00 += p10;
        if (take01) v01 += p11;
        if (live) {
          atomicAdd(dib + i00 * C + ch, v00);
          atomicAdd(dib + i01 * C + ch, v01);
        }
        if (keep10) atomicAdd(dib + i10 * C + ch, v10);
        if (keep11) atomicAdd(dib + i11 * C + ch, v11);
      }
    }
    if (!live) continue;
    if constexpr (GEOM) {
      float dd_, dt0, dt1, dt2;
      geom_bwd(ps, px.g, px.d, cs.trans != nullptr, px.t0, px.t1, px.t2, gx_, gy_, 0.0f, dd_, dt0, dt1, dt2, dp);
      d_depth[gi] = dd_;
      if (d_trans) st3(d_trans + gi * 3, dt0, dt1, dt2);
    } else {
      *reinterpret_cast<float2*>(d_coords + gi * 2) = make_float2(gx_, gy_);
    }
  }
  if constexpr (GEOM) block_sum_store<12>(dp, red, partials + ((long)b * gridDim.x + blockIdx.x) * 12);
}

// ---------------------------------------------------------------- splat forward
// acc [B,H,W,C+1]: channels 0..C-1 = sum of g*img, channel C = sum of g.  Pre-zeroed.
template <bool GEOM, int CT>
__global__ __launch_bounds__(kBlock) void splat_fwd_kernel(CoordSrc cs, const float* __restrict__ img, int H, int W,
                                                           int C, int use_depth, float* __restrict__ acc) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  Pose ps;
  if constexpr (GEOM) ps = load_pose(cs.params + (long)b * SFM_PARAM_STRIDE);
  if (p >= H * W) return;
  Pix px = fetch<GEOM>(cs, ps, b, p, H, W);
  Corners c = corners(px.x, px.y, (float)(W - 1), (float)(H - 1));
  SplatW sw = splat_weights(px.x, px.y, c, use_depth != 0, px.z);
  const int A = C + 1;
  float* ab = acc + (long)b * H * W * A;
  const int idx[4] = {c.y0 * W + c.x0, c.y1 * W + c.x0, c.y0 * W + c.x1, c.y1 * W + c.x1};
  const float* ip = img + ((long)b * H * W + p) * C;
  if constexpr (CT == 3) {
    f3 v = ld3(ip);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* a = ab + idx[k] * 4;
      atomicAdd(a + 0, v.x * sw.g[k]);
      atomicAdd(a + 1, v.y * sw.g[k]);
      atomicAdd(a + 2, v.z * sw.g[k]);
      atomicAdd(a + 3, sw.g[k]);
    }
  } else {
    for (int k = 0; k < 4; ++k) {
      float* a = ab + idx[k] * A;
      for (int ch = 0; ch < C; ++ch) atomicAdd(a + ch, ip[ch] * sw.g[k]);
      atomicAdd(a + C, sw.g[k]);
    }
  }
}

// out = S / (D + 1e-10)   (utils.py:463, 470)
__global__ __launch_bounds__(kBlock) void splat_normalise_kernel(const float* __restrict__ acc, long npix, int C,
                                                                 float* __restrict__ out) {
  long p = (long)blockIdx.x * kBlock + threadIdx.x;
  if (p >= npix) return;
  const float* a = acc + p * (C + 1);
  float den = a[C] + 1e-10f;
  for (int ch = 0; ch < C; ++ch) out[p * C + ch] = a[ch] / den;
}

// mask = avg_pool3x3_SAME(1 - [out == 0]), mean over the in-bounds taps (utils.py:478-479)
__global__ __launch_bounds__(kBlock) void splat_mask_kernel(const float* __restrict__ out, int H, int W, int C,
                                                            float* __restrict__ mask) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int v = p / W, u = p - v * W;
  const float* ob = out + (long)b * H * W * C;
  const int y0 = max(v - 1, 0), y1 = min(v + 1, H - 1), x0 = max(u - 1, 0), x1 = min(u + 1, W - 1);
  const float cnt = (float)((y1 - y0 + 1) * (x1 - x0 + 1));
  for (int ch = 0; ch < C; ++ch) {
    float s = 0.0f;
    for (int yy = y0; yy <= y1; ++yy)
      for (int xx = x0; xx <= x1; ++xx) s += (ob[(yy * W + xx) * C + ch] == 0.0f) ? 0.0f : 1.0f;
    mask[((long)b * H * W + p) * C + ch] = s / cnt;
  }
}

// out and mask of the C = 3 splat in one pass over the accumulator: out = S / (D + 1e-10) of the pixel, mask = the 3x3
// SAME average of [out != 0] with the neighbours' quotients recomputed from their accumulators (L2 hits) - the separate
// normalise and mask kernels read `out` back from memory.
__global__ __launch_bounds__(kBlock) void splat_finish3_kernel(const float* __restrict__ acc, int H, int W,
                                                               float* __restrict__ out, float* __restrict__ mask) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int v = p / W, u = p - v * W;
  const float4* ab = reinterpret_cast<const float4*>(acc) + (long)b * H * W;
  const int y0 = max(v - 1, 0), y1 = min(v + 1, H - 1), x0 = max(u - 1, 0), x1 = min(u + 1, W - 1);
  const float cnt = (float)((y1 - y0 + 1) * (x1 - x0 + 1));
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  f3 mine{0.f, 0.f, 0.f};
  for (int yy = y0; yy <= y1; ++yy)
    for (int xx = x0; xx <= x1; ++xx) {
      const float4 a = ab[yy * W + xx];
      const float den = a.w + 1e-10f;
      const float o0 = a.x / den, o1 = a.y / den, o2 = a.z / den;
      s0 += (o0 == 0.0f) ? 0.0f : 1.0f;
      s1 += (o1 == 0.0f) ? 0.0f : 1.0f;
      s2 += (o2 == 0.0f) ? 0.0f : 1.0f;
      if (yy == v && xx == u) mine = f3{o0, o1, o2};
    }
  const long gi = (long)b * H * W + p;
  st3(out + gi * 3, mine.x, mine.y, mine.z);
  st3(mask + gi * 3, s0 / cnt, s1 / cnt, s2 / cnt);
}

// ---------------------------------------------------------------- splat backward
// gout = dL/d out.  dL/dg_ab = sum_c gout[p_ab,c] (img[n,c] - out[p_ab,c]) / (D[p_ab] + 1e-10)
template <bool GEOM, int CT>
__global__ __launch_bounds__(kBlock) void splat_bwd_kernel(CoordSrc cs, const float* __restrict__ img,
                                                           const float* __restrict__ acc,
                                                           const float* __restrict__ gout, int H, int W, int C,
                                                           int use_depth, float* __restrict__ d_coords,
                                                           float* __restrict__ d_z, float* __restrict__ d_depth,
                                                           float* __restrict__ d_trans, float* __restrict__ partials,
                                                           float* __restrict__ d_img) {
  __shared__ float red[4 * 12];
  const int b = blockIdx.y;
  Pose ps;
  if constexpr (GEOM) ps = load_pose(cs.params + (long)b * SFM_PARAM_STRIDE);
  float dp[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) dp[k] = 0.0f;
  const int A = C + 1;
  const float* ab = acc + (long)b * H * W * A;
  const float* gb = gout + (long)b * H * W * C;
  for (int it = 0; it < kPPT; ++it) {
    const int p = (blockIdx.x * kPPT + it) * kBlock + threadIdx.x;
    if (p >= H * W) break;
    Pix px = fetch<GEOM>(cs, ps, b, p, H, W);
    Corners c = corners(px.x, px.y, (float)(W - 1), (float)(H - 1));
    SplatW sw = splat_weights(px.x, px.y, c, use_depth != 0, px.z);
    const int idx[4] = {c.y0 * W + c.x0, c.y1 * W + c.x0, c.y0 * W + c.x1, c.y1 * W + c.x1};
    const long gi = (long)b * H * W + p;
    const float* ip = img + gi * C;
    float inv[4], dg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      inv[k] = 1.0f / (ab[idx[k] * A + C] + 1e-10f);
      dg[k] = 0.0f;
    }
    for (int ch = 0; ch < (CT == 3 ? 3 : C); ++ch) {
      const float v = ip[ch];
      float di = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float go = gb[idx[k] * C + ch] * inv[k];
        dg[k] += go * (v - ab[idx[k] * A + ch] * inv[k]);
        di += sw.g[k] * go;
      }
      if (d_img) d_img[gi * C + ch] = di;
    }
    // dg/dx: corner a=0 uses dx0 = x - x0s (+1), corner a=1 uses dx1 = x1s - x (-1)
    float gx_ = -(dg[0] * sw.g[0] + dg[1] * sw.g[1]) * sw.dx0 + (dg[2] * sw.g[2] + dg[3] * sw.g[3]) * sw.dx1;
    float gy_ = -(dg[0] * sw.g[0] + dg[2] * sw.g[2]) * sw.dy0 + (dg[1] * sw.g[1] + dg[3] * sw.g[3]) * sw.dy1;
    float gz_ = -0.25f * sw.zin * (((dg[0] * sw.g[0] + dg[1] * sw.g[1]) + dg[2] * sw.g[2]) + dg[3] * sw.g[3]);
    if constexpr (GEOM) {
      float dd_, dt0, dt1, dt2;
      geom_bwd(ps, px.g, px.d, cs.trans != nullptr, px.t0, px.t1, px.t2, gx_, gy_, gz_, dd_, dt0, dt1, dt2, dp);
      d_depth[gi] = dd_;
      if (d_trans) st3(d_trans + gi * 3, dt0, dt1, dt2);
    } else {
      *reinterpret_cast<float2*>(d_coords + gi * 2) = make_float2(gx_, gy_);
      if (d_z) d_z[gi] = gz_;
    }
  }
  if constexpr (GEOM) block_sum_store<12>(dp, red, partials + ((long)b * gridDim.x + blockIdx.x) * 12);
}

static inline bool shape_ok(int B, int H, int W, int C) {
  return B > 0 && H > 0 && W > 0 && C > 0 && B <= 65535 && sfm_fits(H, W, C + 1);
}

}  // namespace sfm

using namespace sfm;

extern "C" int sfm_bwd_blocks(int H, int W) { return (H * W + kBlock * kPPT - 1) / (kBlock * kPPT); }

extern "C" int sfm_project_coords(const float* depth, const float* trans, const float* params, int B, int H, int W,
                                  float* coords, float* z, void* stream) {
  if (!depth || !params || !coords || !shape_ok(B, H, W, 3)) return SFM_ERR_ARG;
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  dim3 grid((H * W + kBlock - 1) / kBlock, B);
  hipLaunchKernelGGL(project_coords_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, cs, H, W, coords, z);
  return sfm_last_error();
}

#define SFM_DISPATCH_C(KERNEL, GEOM, C, ...)                                                          \
  do {                                                                                                \
    if ((C) == 3)                                                                                     \
      hipLaunchKernelGGL((KERNEL<GEOM, 3>), grid, dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); \
    else                                                                                              \
      hipLaunchKernelGGL((KERNEL<GEOM, 0>), grid, dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); \
  } while (0)

extern "C" int sfm_bilinear_sampler_fwd(const float* img, const float* coords, int B, int Hs, int Ws, int C, int Ht,
                                        int Wt, float* out, float* mask, int* idx_dump, void* stream) {
  if (!img || !coords || !out || !mask || !shape_ok(B, Hs, Ws, C) || !shape_ok(B, Ht, Wt, C)) return SFM_ERR_ARG;
  CoordSrc cs{coords, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  dim3 grid((Ht * Wt + kBlock - 1) / kBlock, B);
  SFM_DISPATCH_C(sampler_fwd_kernel, false, C, cs, img, Hs, Ws, C, Ht, Wt, out, mask, idx_dump);
  return sfm_last_error();
}

extern "C" int sfm_bilinear_sampler_bwd(const float* img, const float* coords, const float* gout, int B, int Hs,
                                        int Ws, int C, int Ht, int Wt, float* d_coords, float* d_img, void* stream) {
  if (!img || !coords || !gout || !d_coords || !shape_ok(B, Hs, Ws, C) || !shape_ok(B, Ht, Wt, C)) return SFM_ERR_ARG;
  CoordSrc cs{coords, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
  dim3 grid(sfm_bwd_blocks(Ht, Wt), B);
  SFM_DISPATCH_C(sampler_bwd_kernel, false, C, cs, img, gout, Hs, Ws, C, Ht, Wt, d_coords, (float*)nullptr,
                 (float*)nullptr, (float*)nullptr, d_img);
  return sfm_last_error();
}

extern "C" int sfm_inverse_warp_fwd(const float* img, const float* depth, const float* trans, const float* params,
                                    int B, int H, int W, int C, float* out, float* mask, int* idx_dump, void* stream) {
  if (!img || !depth || !params || !out || !mask || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  dim3 grid((H * W + kBlock - 1) / kBlock, B);
  SFM_DISPATCH_C(sampler_fwd_kernel, true, C, cs, img, H, W, C, H, W, out, mask, idx_dump);
  return sfm_last_error();
}

extern "C" int sfm_inverse_warp_bwd(const float* img, const float* depth, const float* trans, const float* params,
                                    const float* gout, int B, int H, int W, int C, float* d_depth, float* d_trans,
                                    float* partials, float* d_img, void* stream) {
  if (!img || !depth || !params || !gout || !d_depth || !partials || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  if (trans && !d_trans) return SFM_ERR_ARG;
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  dim3 grid(sfm_bwd_blocks(H, W), B);
  SFM_DISPATCH_C(sampler_bwd_kernel, true, C, cs, img, gout, H, W, C, H, W, (float*)nullptr, d_depth,
                 trans ? d_trans : (float*)nullptr, partials, d_img);
  return sfm_last_error();
}

// Forward splat.  acc is a caller-provided [B,H,W,C+1] workspace that this call
// zeroes (hipMemsetAsync on the stream) and that the backward needs again.
static int splat_fwd_common(bool geom, CoordSrc cs, const float* img, int B, int H, int W, int C, int use_depth,
                            float* acc, float* out, float* mask, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(acc, 0, (size_t)B * H * W * (C + 1) * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  dim3 grid((H * W + kBlock - 1) / kBlock, B);
  // projective_warp itself (geometry in registers, RGB, depth-weighted): the tile-owned LDS splat of the fused loss -
  // workgroups own 64x32 destination tiles and accumulate in LDS, global atomics only for the tile flush and the rare
  // corner no owner scans.  Its scan-window table lives in `mask`, which is not written before the last kernel.
  static const bool tile_owned = [] { const char* e = getenv("SFM_SPLAT_TILE"); return !e || e[0] != '0'; }();  // 0: experiments
  if (tile_owned && geom && C == 3 && use_depth && ((uintptr_t)mask & 15) == 0 && ((uintptr_t)acc & 15) == 0 &&
      splat_solo_table_int4s(B, H, W) * 4 <= (long)B * H * W * 3 &&
      launch_splat_solo(img, cs.depth, cs.trans, cs.params, B, H, W, acc, mask, stream)) {
    hipLaunchKernelGGL(splat_finish3_kernel, grid, dim3(kBlock), 0, stream, acc, H, W, out, mask);
    return sfm_last_error();
  }
  if (geom)
    SFM_DISPATCH_C(splat_fwd_kernel, true, C, cs, img, H, W, C, use_depth, acc);
  else
    SFM_DISPATCH_C(splat_fwd_kernel, false, C, cs, img, H, W, C, use_depth, acc);
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(splat_normalise_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     acc, npix, C, out);
  hipLaunchKernelGGL(splat_mask_kernel, grid, dim3(kBlock), 0, stream, out, H, W, C, mask);
  return sfm_last_error();
}

extern "C" int sfm_bilinear_project_fwd(const float* img, const float* coords, const float* z, int B, int H, int W,
                                        int C, float* acc, float* out, float* mask, void* stream) {
  if (!img || !coords || !acc || !out || !mask || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  CoordSrc cs{coords, z, nullptr, nullptr, nullptr, 0.f, 0.f};
  return splat_fwd_common(false, cs, img, B, H, W, C, z != nullptr, acc, out, mask, (hipStream_t)stream);
}

extern "C" int sfm_splat_warp_fwd(const float* img, const float* depth, const float* trans, const float* params,
                                  int B, int H, int W, int C, float* acc, float* out, float* mask, void* stream) {
  if (!img || !depth || !params || !acc || !out || !mask || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  return splat_fwd_common(true, cs, img, B, H, W, C, 1, acc, out, mask, (hipStream_t)stream);
}

// ---- the allocation-free variant of sfm_splat_warp_fwd (no counterpart in the reference) ----------------------
// For callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
// [B,H,W,C+1] accumulator lives in a file-scope device array instead of a caller-provided workspace.  Each GPU the
// library runs on holds its own copy of the array (at its own address) until the process ends; the host only ever
// uses the address hipGetSymbolAddress returned for the CURRENT device - never the symbol itself, which in host code
// is a stand-in, not a device address.
namespace sfm {
constexpr size_t kStaticAccFloats = (size_t)4 << 20;  // 16 MiB per GPU: B*H*W*(C+1) <= 4 Mi floats (c2: 1.7 Mi)
static __device__ __align__(256) float g_static_acc[kStaticAccFloats];
constexpr int kMaxDevices = 64;
static float* g_static_acc_ptr[kMaxDevices];  // per device, filled on the first call made with that device current
}  // namespace sfm

extern "C" long long sfm_splat_warp_fwd_static_max_floats(void) { return (long long)kStaticAccFloats; }

extern "C" int sfm_splat_warp_fwd_static(const float* img, const float* depth, const float* trans, const float* params,
                                         int B, int H, int W, int C, float* out, float* mask, void* stream) {
  if (!img || !depth || !params || !out || !mask || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  if ((size_t)B * H * W * (C + 1) > kStaticAccFloats) return SFM_ERR_TOO_LARGE;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return SFM_ERR_ARG;
  float* acc = g_static_acc_ptr[dev];
  if (!acc) {  // once per device, with that device current (racing first calls resolve the same address)
    e = hipGetSymbolAddress((void**)&acc, HIP_SYMBOL(g_static_acc));
    if (e != hipSuccess) return (int)e;
    if (!acc) return SFM_ERR_ARG;
    g_static_acc_ptr[dev] = acc;
  }
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  // splat_fwd_common clears the part of the array this call uses (the last call's sums are still in it)
  return splat_fwd_common(true, cs, img, B, H, W, C, 1, acc, out, mask, (hipStream_t)stream);
}

extern "C" int sfm_bilinear_project_bwd(const float* img, const float* coords, const float* z, const float* acc,
                                        const float* gout, int B, int H, int W, int C, float* d_coords, float* d_z,
                                        float* d_img, void* stream) {
  if (!img || !coords || !acc || !gout || !d_coords || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  CoordSrc cs{coords, z, nullptr, nullptr, nullptr, 0.f, 0.f};
  dim3 grid(sfm_bwd_blocks(H, W), B);
  SFM_DISPATCH_C(splat_bwd_kernel, false, C, cs, img, acc, gout, H, W, C, z != nullptr, d_coords, z ? d_z : (float*)nullptr,
                 (float*)nullptr, (float*)nullptr, (float*)nullptr, d_img);
  return sfm_last_error();
}

extern "C" int sfm_splat_warp_bwd(const float* img, const float* depth, const float* trans, const float* params,
                                  const float* acc, const float* gout, int B, int H, int W, int C, float* d_depth,
                                  float* d_trans, float* partials, float* d_img, void* stream) {
  if (!img || !depth || !params || !acc || !gout || !d_depth || !partials || !shape_ok(B, H, W, C)) return SFM_ERR_ARG;
  if (trans && !d_trans) return SFM_ERR_ARG;
  CoordSrc cs{nullptr, nullptr, depth, trans, params, sfm_lin_step(W), sfm_lin_step(H)};
  dim3 grid(sfm_bwd_blocks(H, W), B);
  SFM_DISPATCH_C(splat_bwd_kernel, true, C, cs, img, acc, gout, H, W, C, 1, (float*)nullptr, (float*)nullptr, d_depth,
                 trans ? d_trans : (float*)nullptr, partials, d_img);
  return sfm_last_error();
}
