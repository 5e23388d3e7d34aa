// Streaming L1 + SSIM stencil of the fused loss (SfMLearner_uni.py:137-149, 168; tf_ssim
// utils.py:548-575), forward sums AND backward seeds in ONE pass over the data.
//
// Why one pass is possible: the backward through  mean(m) * clip((1 - mean ssim(Sp, src)) / 2)
// needs the GLOBAL sums M = sum m and V = sum ssim (utils.py:573-575 takes the mean before the
// clip).  Everything downstream of this kernel is linear in dL/dSp, so the kernel writes the two
// parts separately - hA (masked-L1 part, coefficient alpha/n known up front) and hB (SSIM part,
// to be scaled by kV = (1-alpha) (M/n) (-1/2) [clip passes] / n once the sums are global) - and
// the backward warp combines them.  The gather side (min-reprojection, :168) has no such
// dependence: dL/dG is final here.
//
// Work decomposition: one WAVE owns a strip of 60 image columns (64 lanes, 2 halo lanes on each
// side) and marches down `rows` image rows.  Horizontal neighbours come from the adjacent lanes
// by DPP wave shifts (v_*_dpp wave_shr:1 / wave_shl:1: no LDS, half the plain VALU rate), vertical neighbours from
// a 3-row sliding window held in registers; the 3x3 Gaussian window of tf_ssim is separable
// (w_ij = g_i g_j), so every windowed moment is one horizontal and one vertical 3-tap pass.
// The chained second stencil of the backward (dSSIM/dx = conv(a) + 2 x conv(b) + y conv(c), the
// transpose of the window applied to the per-pixel derivative coefficients) runs two rows
// behind in the same sweep.  No LDS, no barriers, every input element is loaded once per strip
// (x 64/60) and chunk (x (rows+4)/rows) with fully coalesced 12/16-byte-per-lane loads.
//
// This file is compiled WITH fp contraction (FMA) and uses v_rcp_f32 for the SSIM quotients and
// the splat normalisation: nothing here feeds an integer pixel index, the tolerance is the
// loss's 1e-5 / the gradients' 1e-4, and the kernel is VALU-issue bound.
#include "loss_common.h"
#include <type_traits>

namespace sfm {

namespace {

// tf_ssim window (utils.py:504-519, size 3, sigma 1.5) in separable form w_ij = g_|i| * g_|j|,
// g = (e1, 1, e1) / (1 + 2 e1), e1 = exp(-1/4.5).  The fp32 values are the pair of neighbouring
// floats whose 2-D weights sum to exactly 1, as the reference's normalised fp32 window does: a
// weight sum of 1 + eps biases every variance by eps * mean^2, which tf_ssim's C2 = 9e-4 turns
// into a systematic ~eps / C2 relative error of the SSIM gradients (6e-8 -> 7e-5).
constexpr float kG0 = 0.38439732789993286f;
constexpr float kG1 = 0.30780133605003357f;

__device__ __forceinline__ float lane_left(float v) {  // lane i <- lane i-1 (0 into lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {  // lane i <- lane i+1 (0 into lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float hsum3(float v) { return kG0 * v + kG1 * (lane_left(v) + lane_right(v)); }
__device__ __forceinline__ float vsum3(float top, float mid, float bot) { return kG0 * mid + kG1 * (top + bot); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

__device__ __forceinline__ float clip01_half(float v) {  // clip((1 - v)/2, 0, 1), utils.py:575
  return fminf(fmaxf((1.0f - v) / 2.0f, 0.0f), 1.0f);
}
__device__ __forceinline__ float clip01_half_pass(float v) {
  const float h = (1.0f - v) / 2.0f;
  return (h >= 0.0f && h <= 1.0f) ? 1.0f : 0.0f;
}
__device__ __forceinline__ float sgn(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }

// SSIM value (utils.py:572) from the windowed raw moments m1 = E[x], m2 = E[y], mss = E[x*x + y*y]
// (the denominator only needs sigma_x^2 + sigma_y^2, so the two squares share one filter) and
// m12 = E[x*y]; if GRAD also its partial derivatives with respect to E[x] (a), E[x*x] (b) and
// E[x*y] (c), x being the differentiated image.  One reciprocal per factor of the denominator.
template <bool GRAD>
__device__ __forceinline__ float ssim_point(float m1, float m2, float mss, float m12, float& a, float& b, float& c) {
  const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
  const float musum = mu1_sq + mu2_sq;
  const float A1 = 2.0f * mu12 + kSsimC1, A2 = 2.0f * (m12 - mu12) + kSsimC2;
  const float B1 = musum + kSsimC1, B2 = (mss - musum) + kSsimC2;
  const float iB1 = fast_rcp(B1), iB2 = fast_rcp(B2);
  const float iB = iB1 * iB2;
  const float v = (A1 * A2) * iB;
  if constexpr (GRAD) {
    c = 2.0f * A1 * iB;                                   // dv / d s12
    b = -v * iB2;                                         // dv / d s11
    const float dv_dm1 = (2.0f * m2 * A2) * iB - v * (2.0f * m1) * iB1;
    a = dv_dm1 - 2.0f * m1 * b - m2 * c;
  }
  return v;
}

template <int V>
using IC = std::integral_constant<int, V>;

struct Item {
  int img, chunk, strip;
};
// 32-bit arithmetic (the item count is checked on the host): a 64-bit division pair here was ~7 % of a wave's instructions
__device__ __forceinline__ Item split_item(long item, const StencilPlan& p) {
  const unsigned u = (unsigned)item, strips = (unsigned)p.strips, chunks = (unsigned)p.chunks;
  Item it;
  const unsigned r = u / strips;
  it.strip = (int)(u - r * strips);
  const unsigned q = r / chunks;
  it.chunk = (int)(r - q * chunks);
  it.img = (int)q;
  return it;
}

__device__ __forceinline__ float rows_in(int q, int H) {  // in-bounds taps of a 3-tap window at q
  return (float)(min(q + 1, H - 1) - max(q - 1, 0) + 1);
}

// ------------------------------------------------------------------ splat side
// x = Sp_i = Acc.rgb / (Acc.w + 1e-10) (utils.py:463,470), y = src_i.  Sums A = sum m |x - y|,
// M = sum m, V = sum ssim(x, y) with m = avgpool3x3_SAME(1 - [x == 0]) (utils.py:478-479).
//
// Register windows are 3-slot rings indexed by (row mod 3); the row loop is unrolled by 3 with the
// phase as a compile-time constant, so sliding a window is a renaming, not a move.
template <bool GRAD>
__device__ __forceinline__ void splat_item(const StencilArgs& a, long item, float* __restrict__ wavep_out) {
  const LossDims d = a.d;
  const Item it = split_item(item, a.p);
  const int lane = threadIdx.x & 63;
  const int b = it.img % d.B, i = (it.img / d.B) % d.N;
  const int HW = d.H * d.W, C3 = 3 * d.N;
  const int gx = it.strip * kStripOwn - 2 + lane;
  const bool col_in = gx >= 0 && gx < d.W;
  const bool own_col = lane >= 2 && lane < 62 && gx < d.W;
  const int y0 = it.chunk * a.p.rows, y1 = min(y0 + a.p.rows, d.H);
  const float* __restrict__ acc = a.Acc + (size_t)it.img * HW * 4;
  const float* __restrict__ ovf = a.flags[it.img] ? a.Ovf + (size_t)it.img * HW * 4 : nullptr;  // wave-uniform
  const float* __restrict__ sp = a.src + (size_t)b * HW * C3 + 3 * i;
  float* __restrict__ hb = a.Hb + (size_t)it.img * HW * 4;  // part A; part B one plane further
  const size_t hb_plane = (size_t)d.S * d.N * d.B * HW * 4;
  const float nx = (float)(min(gx + 1, d.W - 1) - max(gx - 1, 0) + 1);
  constexpr int HAL = GRAD ? 2 : 1;
  const int r_first = y0 - HAL, r_last = y1 - 1 + HAL;

  // rings, slot = (row - r_first) mod 3.  moments per channel: x, y, xx + yy, xy
  float Hm[3][3][4], Nz[3][3], Ch[3][3][3], X[3][3], Y[3][3], Iv[3], Mk[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Iv[k] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      Nz[k][ch] = X[k][ch] = Y[k][ch] = Mk[k][ch] = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) Hm[k][ch][m] = 0.f;
#pragma unroll
      for (int m = 0; m < 3; ++m) Ch[k][ch][m] = 0.f;
    }
  }
  float sA = 0.f, sM = 0.f, sV = 0.f;

  float4 na = make_float4(0.f, 0.f, 0.f, 0.f);  // software prefetch of the next row's raw values
  f3 ns{0.f, 0.f, 0.f};
  auto fetch = [&](int r) {
    na = make_float4(0.f, 0.f, 0.f, 0.f);
    ns = f3{0.f, 0.f, 0.f};
    if (r >= 0 && r < d.H && col_in) {
      const size_t p = (size_t)r * d.W + gx;
      na = *reinterpret_cast<const float4*>(acc + p * 4);
      if (ovf) {
        const float4 o = *reinterpret_cast<const float4*>(ovf + p * 4);
        na.x += o.x; na.y += o.y; na.z += o.z; na.w += o.w;
      }
      ns = ld3(sp + p * C3);
    }
  };

  auto step = [&](int r, auto ph) {
    constexpr int S0 = decltype(ph)::value;            // slot of row r
    constexpr int S1 = (S0 + 2) % 3, S2 = (S0 + 1) % 3;  // slots of rows r-1, r-2 (r-3 shares S0)
    const float4 ca = na;
    const f3 cs = ns;
    if (r + 1 <= r_last) fetch(r + 1);
    // ---- row r: values, horizontal sums
    if (r >= 0 && r < d.H) {  // wave-uniform
      const float inv = col_in ? fast_rcp(ca.w + 1e-10f) : 0.f;
      Iv[S0] = inv;
      const float xv[3] = {ca.x * inv, ca.y * inv, ca.z * inv}, yv[3] = {cs.x, cs.y, cs.z};  // 0 outside the image
      // all products first, then all cross-lane sums: a DPP operand needs two wait states after the VALU write of
      // its source, and the compiler pads with s_nop when the product is formed right in front of its hsum3
      float sq[3], xy[3], nz[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        X[S0][ch] = xv[ch];
        Y[S0][ch] = yv[ch];
        sq[ch] = xv[ch] * xv[ch] + yv[ch] * yv[ch];
        xy[ch] = xv[ch] * yv[ch];
        nz[ch] = (xv[ch] == 0.0f) ? 0.0f : 1.0f;
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        Hm[S0][ch][0] = hsum3(xv[ch]);
        Hm[S0][ch][1] = hsum3(yv[ch]);
        Hm[S0][ch][2] = hsum3(sq[ch]);
        Hm[S0][ch][3] = hsum3(xy[ch]);
        Nz[S0][ch] = nz[ch] + (lane_left(nz[ch]) + lane_right(nz[ch]));
      }
    } else {
      Iv[S0] = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        X[S0][ch] = Y[S0][ch] = Nz[S0][ch] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) Hm[S0][ch][m] = 0.f;
      }
    }
    // ---- row q = r - 1: windowed moments -> SSIM value, derivative coefficients, mask, sums
    const int q = r - 1;
    if (r - r_first >= 2) {
      const bool pix_in = q >= 0 && q < d.H && col_in;
      const bool own = own_col && q >= y0 && q < y1;
      const float rcnt = 1.0f / (rows_in(q, d.H) * nx);  // k * fl(1/k) == 1 exactly for k = 4, 6, 9
      float cf[3][3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float ca_ = 0.f, cb_ = 0.f, cc_ = 0.f;
        const float v = ssim_point<GRAD>(vsum3(Hm[S2][ch][0], Hm[S1][ch][0], Hm[S0][ch][0]),
                                         vsum3(Hm[S2][ch][1], Hm[S1][ch][1], Hm[S0][ch][1]),
                                         vsum3(Hm[S2][ch][2], Hm[S1][ch][2], Hm[S0][ch][2]),
                                         vsum3(Hm[S2][ch][3], Hm[S1][ch][3], Hm[S0][ch][3]), ca_, cb_, cc_);
        const float m = ((Nz[S2][ch] + Nz[S1][ch]) + Nz[S0][ch]) * rcnt;
        Mk[S1][ch] = m;
        if (own) {
          sV += v;
          sA += m * fabsf(X[S1][ch] - Y[S1][ch]);
          sM += m;
        }
        if constexpr (GRAD) {
          cf[ch][0] = pix_in ? ca_ : 0.f;
          cf[ch][1] = pix_in ? cb_ : 0.f;
          cf[ch][2] = pix_in ? cc_ : 0.f;
        }
      }
      if constexpr (GRAD) {  // coefficients of row q overwrite those of row q - 3 (same slot); sums after all values (DPP hazard)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int m = 0; m < 3; ++m) Ch[S1][ch][m] = hsum3(cf[ch][m]);
      }
    }
    // ---- row p = r - 2: backward seeds (coefficient rows p-1, p, p+1 sit in slots S0, S2, S1)
    if constexpr (GRAD) {
      const int p = r - 2;
      if (own_col && p >= y0 && p < y1) {
        float hA[3], hB[3], wA = 0.f, wB = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float xv = X[S2][ch], yv = Y[S2][ch];
          const float rr = vsum3(Ch[S0][ch][0], Ch[S2][ch][0], Ch[S1][ch][0]) +
                           2.0f * vsum3(Ch[S0][ch][1], Ch[S2][ch][1], Ch[S1][ch][1]) * xv +
                           vsum3(Ch[S0][ch][2], Ch[S2][ch][2], Ch[S1][ch][2]) * yv;
          hA[ch] = (a.kL1 * Mk[S2][ch] * sgn(xv - yv)) * Iv[S2];
          hB[ch] = rr * Iv[S2];
          wA += hA[ch] * xv;
          wB += hB[ch] * xv;
        }
        float* o = hb + ((size_t)p * d.W + gx) * 4;
        *reinterpret_cast<float4*>(o) = make_float4(hA[0], hA[1], hA[2], wA);
        *reinterpret_cast<float4*>(o + hb_plane) = make_float4(hB[0], hB[1], hB[2], wB);
      }
    }
  };

  fetch(r_first);
  for (int r = r_first; r <= r_last; r += 3) {
    step(r, IC<0>{});
    if (r + 1 <= r_last) step(r + 1, IC<1>{});
    if (r + 2 <= r_last) step(r + 2, IC<2>{});
  }
  sA = wave_sum_dpp_lane63(sA); sM = wave_sum_dpp_lane63(sM); sV = wave_sum_dpp_lane63(sV);
  if (lane == 63) *reinterpret_cast<float4*>(wavep_out) = make_float4(sA, sM, sV, 0.f);
}

// ------------------------------------------------------------------ gather side
// e_i = alpha |G_i - tgt| + (1 - alpha) clip((1 - ssim(G_i, tgt)) / 2, 0, 1)  (:137,142), E = sum min(e_0, e_1)
// (:168).  dG_i = kE [ alpha sel_i sgn(G_i - tgt) + (1 - alpha)(-1/2) (ssim backward of the selected side) ].
template <bool GRAD>
__device__ __forceinline__ void gather_item(const StencilArgs& a, long item, float* __restrict__ wavep_out) {
  const LossDims d = a.d;
  const Item it = split_item(item, a.p);
  const int lane = threadIdx.x & 63;
  const int s = it.img / d.B, b = it.img - s * d.B;
  const int HW = d.H * d.W;
  const int gx = it.strip * kStripOwn - 2 + lane;
  const bool col_in = gx >= 0 && gx < d.W;
  const bool own_col = lane >= 2 && lane < 62 && gx < d.W;
  const int y0 = it.chunk * a.p.rows, y1 = min(y0 + a.p.rows, d.H);
  const size_t gpix = (size_t)d.B * HW;
  const size_t o0 = (((size_t)s * 2 + 0) * gpix + (size_t)b * HW) * 3;
  const size_t o1 = (((size_t)s * 2 + 1) * gpix + (size_t)b * HW) * 3;
  const float* __restrict__ g0p = a.G + o0;
  const float* __restrict__ g1p = a.G + o1;
  const float* __restrict__ tp = a.tgt + (size_t)b * HW * 3;
  constexpr int HAL = GRAD ? 2 : 1;
  const int r_first = y0 - HAL, r_last = y1 - 1 + HAL;
  const float ku = a.kE * (1.0f - kAlpha) * (-0.5f);
  const float kl = a.kE * kAlpha;

  // rings, slot = (row - r_first) mod 3.  moments per channel: t, x0, x1, x0x0 + tt, x1x1 + tt, x0t, x1t
  float Hm[3][3][7];
  float Ch[3][3][6];  // (a0, b0, c0, a1, b1, c1), already scaled by the selection
  float X0[3][3], X1[3][3], T[3][3], Sel[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      X0[k][ch] = X1[k][ch] = T[k][ch] = Sel[k][ch] = 0.f;
#pragma unroll
      for (int m = 0; m < 7; ++m) Hm[k][ch][m] = 0.f;
#pragma unroll
      for (int m = 0; m < 6; ++m) Ch[k][ch][m] = 0.f;
    }
  float sE = 0.f;

  f3 n0{0.f, 0.f, 0.f}, n1{0.f, 0.f, 0.f}, nt{0.f, 0.f, 0.f};
  auto fetch = [&](int r) {
    n0 = n1 = nt = f3{0.f, 0.f, 0.f};
    if (r >= 0 && r < d.H && col_in) {
      const size_t p = ((size_t)r * d.W + gx) * 3;
      n0 = ld3(g0p + p);
      n1 = ld3(g1p + p);
      nt = ld3(tp + p);
    }
  };

  auto step = [&](int r, auto ph) {
    constexpr int S0 = decltype(ph)::value;
    constexpr int S1 = (S0 + 2) % 3, S2 = (S0 + 1) % 3;
    const float x0[3] = {n0.x, n0.y, n0.z}, x1[3] = {n1.x, n1.y, n1.z}, t[3] = {nt.x, nt.y, nt.z};
    if (r + 1 <= r_last) fetch(r + 1);
    if (r >= 0 && r < d.H) {
      float pr[3][4];  // products first, cross-lane sums after (DPP hazard: see splat_item)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        X0[S0][ch] = x0[ch]; X1[S0][ch] = x1[ch]; T[S0][ch] = t[ch];
        const float tt = t[ch] * t[ch];
        pr[ch][0] = x0[ch] * x0[ch] + tt;
        pr[ch][1] = x1[ch] * x1[ch] + tt;
        pr[ch][2] = x0[ch] * t[ch];
        pr[ch][3] = x1[ch] * t[ch];
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        Hm[S0][ch][0] = hsum3(t[ch]);
        Hm[S0][ch][1] = hsum3(x0[ch]);
        Hm[S0][ch][2] = hsum3(x1[ch]);
        Hm[S0][ch][3] = hsum3(pr[ch][0]);
        Hm[S0][ch][4] = hsum3(pr[ch][1]);
        Hm[S0][ch][5] = hsum3(pr[ch][2]);
        Hm[S0][ch][6] = hsum3(pr[ch][3]);
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        X0[S0][ch] = X1[S0][ch] = T[S0][ch] = 0.f;
#pragma unroll
        for (int m = 0; m < 7; ++m) Hm[S0][ch][m] = 0.f;
      }
    }
    const int q = r - 1;
    if (r - r_first >= 2) {
      const bool pix_in = q >= 0 && q < d.H && col_in;
      const bool own = own_col && q >= y0 && q < y1;
      float cf[3][6];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float mm[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) mm[m] = vsum3(Hm[S2][ch][m], Hm[S1][ch][m], Hm[S0][ch][m]);
        float a0 = 0.f, b0 = 0.f, c0 = 0.f, a1 = 0.f, b1 = 0.f, c1 = 0.f;
        const float v0 = ssim_point<GRAD>(mm[1], mm[0], mm[3], mm[5], a0, b0, c0);
        const float v1 = ssim_point<GRAD>(mm[2], mm[0], mm[4], mm[6], a1, b1, c1);
        const float e0 = kAlpha * fabsf(X0[S1][ch] - T[S1][ch]) + (1.0f - kAlpha) * clip01_half(v0);
        const float e1 = kAlpha * fabsf(X1[S1][ch] - T[S1][ch]) + (1.0f - kAlpha) * clip01_half(v1);
        const bool first = e0 <= e1;  // tf.minimum ties -> either (equal values)
        Sel[S1][ch] = first ? 1.0f : 0.0f;
        if (own) sE += first ? e0 : e1;
        if constexpr (GRAD) {
          const float u0 = (pix_in && first) ? ku * clip01_half_pass(v0) : 0.f;
          const float u1 = (pix_in && !first) ? ku * clip01_half_pass(v1) : 0.f;
          cf[ch][0] = a0 * u0; cf[ch][1] = b0 * u0; cf[ch][2] = c0 * u0;
          cf[ch][3] = a1 * u1; cf[ch][4] = b1 * u1; cf[ch][5] = c1 * u1;
        }
      }
      if constexpr (GRAD) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int m = 0; m < 6; ++m) Ch[S1][ch][m] = hsum3(cf[ch][m]);
      }
    }
    if constexpr (GRAD) {
      const int p = r - 2;
      if (own_col && p >= y0 && p < y1) {
        float d0[3], d1[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float tv = T[S2][ch], xa = X0[S2][ch], xb = X1[S2][ch];
          d0[ch] = kl * Sel[S2][ch] * sgn(xa - tv) +
                   (vsum3(Ch[S0][ch][0], Ch[S2][ch][0], Ch[S1][ch][0]) + 2.0f * vsum3(Ch[S0][ch][1], Ch[S2][ch][1], Ch[S1][ch][1]) * xa +
                    vsum3(Ch[S0][ch][2], Ch[S2][ch][2], Ch[S1][ch][2]) * tv);
          d1[ch] = kl * (1.0f - Sel[S2][ch]) * sgn(xb - tv) +
                   (vsum3(Ch[S0][ch][3], Ch[S2][ch][3], Ch[S1][ch][3]) + 2.0f * vsum3(Ch[S0][ch][4], Ch[S2][ch][4], Ch[S1][ch][4]) * xb +
                    vsum3(Ch[S0][ch][5], Ch[S2][ch][5], Ch[S1][ch][5]) * tv);
        }
        const size_t pp = ((size_t)p * d.W + gx) * 3;
        st3(a.dG + o0 + pp, d0[0], d0[1], d0[2]);
        st3(a.dG + o1 + pp, d1[0], d1[1], d1[2]);
      }
    }
  };

  fetch(r_first);
  for (int r = r_first; r <= r_last; r += 3) {
    step(r, IC<0>{});
    if (r + 1 <= r_last) step(r + 1, IC<1>{});
    if (r + 2 <= r_last) step(r + 2, IC<2>{});
  }
  sE = wave_sum_dpp_lane63(sE);
  if (lane == 63) *reinterpret_cast<float4*>(wavep_out) = make_float4(sE, 0.f, 0.f, 0.f);
}

// One wave per work item, 4 independent waves per block (plain round-robin block order: the
// XCD-contiguous order measured 25 % MORE fetched bytes here - profiles/r01_traffic.md).  Two kernels because the register
// budgets differ (the gather side carries two SSIMs per channel).
template <bool GRAD>
__global__ __launch_bounds__(kBlock) void loss_stencil_gather_kernel(StencilArgs a) {
  const long w = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w < a.p.n_gather) gather_item<GRAD>(a, w, a.wavep + (a.p.n_splat + w) * kWaveTerms);
}
template <bool GRAD>
__global__ __launch_bounds__(kBlock) void loss_stencil_splat_kernel(StencilArgs a) {
  const long w = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w < a.p.n_splat) splat_item<GRAD>(a, w, a.wavep + w * kWaveTerms);
}

}  // namespace

int launch_loss_stencil(const StencilArgs& a, bool need_grad, int which, hipStream_t stream) {
  constexpr int WPB = kBlock / 64;
  const dim3 gs((unsigned)((a.p.n_splat + WPB - 1) / WPB)), gg((unsigned)((a.p.n_gather + WPB - 1) / WPB));
  const bool gather = (which & kStencilGather) && a.p.n_gather, splat = (which & kStencilSplat) != 0;
  if (need_grad) {
    if (gather) hipLaunchKernelGGL(loss_stencil_gather_kernel<true>, gg, dim3(kBlock), 0, stream, a);
    if (splat) hipLaunchKernelGGL(loss_stencil_splat_kernel<true>, gs, dim3(kBlock), 0, stream, a);
  } else {
    if (gather) hipLaunchKernelGGL(loss_stencil_gather_kernel<false>, gg, dim3(kBlock), 0, stream, a);
    if (splat) hipLaunchKernelGGL(loss_stencil_splat_kernel<false>, gs, dim3(kBlock), 0, stream, a);
  }
  return sfm_last_error();
}

}  // namespace sfm
