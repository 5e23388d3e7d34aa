// Streaming L1 + SSIM stencil of the fused loss (SfMLearner_uni.py:137-149, 168; tf_ssim
// utils.py:548-575), forward sums AND backward seeds in ONE pass over the data.
//
// Why one pass is possible: the backward through  mean(m) * clip((1 - mean ssim(Sp, src)) / 2)
// needs the GLOBAL sums M = sum m and V = sum ssim (utils.py:573-575 takes the mean before the
// clip).  M is known before this kernel runs (it needs no window: loss_kv0 in loss_common.h; the
// splat tiles count it, and under data parallelism it is all-reduced while the gather side runs),
// so the seed of the masked-L1 part hA (coefficient alpha/n) and of the SSIM part hB (coefficient
// kV0 = (1-alpha) (M/n) (-1/2) / n) leave as ONE plane h = hA + kV0 hB.  V only decides whether
// the clip passes, which it does unless the mean SSIM leaves [-1, 1]; for that case the REDO form
// of the kernel rewrites the seeds with kV0 = 0 once the sums are global (exits at once otherwise).
// The gather side (min-reprojection, :168) has no such dependence: dL/dG is final here.
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

// occupancy targets (waves per SIMD): see the Makefile / DESIGN.md section 5
#ifndef SFM_WPE_GATHER
#define SFM_WPE_GATHER
#endif
#ifndef SFM_WPE_SPLAT
#define SFM_WPE_SPLAT
#endif

namespace sfm {

namespace {

// tf_ssim window (utils.py:504-519, size 3, sigma 1.5) in separable form w_ij = g_|i| * g_|j|,
// g = (e1, 1, e1) / (1 + 2 e1), e1 = exp(-1/4.5).  The fp32 values are the pair of neighbouring
// floats whose 2-D weights sum to exactly 1, as the reference's normalised fp32 window does: a
// weight sum of 1 + eps biases every variance by eps * mean^2, which tf_ssim's C2 = 9e-4 turns
// into a systematic ~eps / C2 relative error of the SSIM gradients (6e-8 -> 7e-5).
constexpr float kG0 = 0.38439732789993286f;
constexpr float kG1 = 0.30780133605003357f;

// ---- horizontal 3-tap sums across the lanes of a wave --------------------------------------------------------
// out[k] = g0 in[k] + g1 (in[k] of lane-1 + in[k] of lane+1), 0 beyond the ends of the wave.  Written as ONE asm
// statement per group of values: N plain multiplies, then 2N v_fmac_f32_dpp (the DPP operand folded into the FMA: 3
// instructions per sum where the compiler's v_mov_dpp / v_add_dpp / mul / fmac form took 5 plus hazard s_nops, and DPP
// forms issue at half rate - tools/valu_bench.hip).  gfx950 wants 2 wait states between the VALU write of a register
// and a DPP read of it: the N >= 3 leading multiplies are those wait states for anything written just before the
// statement, and inside it the DPP operands are inputs only.  The taps live in VGPRs (DPP encodings take no literal).
#define SFM_DPP_L " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"  // lane i <- lane i-1
#define SFM_DPP_R " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"  // lane i <- lane i+1
#define HS_M(o, i) "v_mul_f32 %" #o ", %[g0], %" #i "\n"
#define HS_L(o, i) "v_fmac_f32_dpp %" #o ", %" #i ", %[g1]" SFM_DPP_L
#define HS_R(o, i) "v_fmac_f32_dpp %" #o ", %" #i ", %[g1]" SFM_DPP_R
struct Taps {
  float g0, g1;
};
__device__ __forceinline__ void hsum3(const float (&in)[3], float (&o)[3], const Taps t) {
  asm(HS_M(0, 3) HS_M(1, 4) HS_M(2, 5) HS_L(0, 3) HS_L(1, 4) HS_L(2, 5) HS_R(0, 3) HS_R(1, 4) HS_R(2, 5)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), [g0] "v"(t.g0), [g1] "v"(t.g1));
}
__device__ __forceinline__ void hsum3(const float (&in)[4], float (&o)[4], const Taps t) {
  asm(HS_M(0, 4) HS_M(1, 5) HS_M(2, 6) HS_M(3, 7) HS_L(0, 4) HS_L(1, 5) HS_L(2, 6) HS_L(3, 7) HS_R(0, 4) HS_R(1, 5)
          HS_R(2, 6) HS_R(3, 7)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), [g0] "v"(t.g0), [g1] "v"(t.g1));
}
__device__ __forceinline__ void hsum3(const float (&in)[6], float (&o)[6], const Taps t) {
  asm(HS_M(0, 6) HS_M(1, 7) HS_M(2, 8) HS_M(3, 9) HS_M(4, 10) HS_M(5, 11) HS_L(0, 6) HS_L(1, 7) HS_L(2, 8) HS_L(3, 9)
          HS_L(4, 10) HS_L(5, 11) HS_R(0, 6) HS_R(1, 7) HS_R(2, 8) HS_R(3, 9) HS_R(4, 10) HS_R(5, 11)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), "v"(in[4]), "v"(in[5]), [g0] "v"(t.g0), [g1] "v"(t.g1));
}
__device__ __forceinline__ void hsum3(const float (&in)[7], float (&o)[7], const Taps t) {
  asm(HS_M(0, 7) HS_M(1, 8) HS_M(2, 9) HS_M(3, 10) HS_M(4, 11) HS_M(5, 12) HS_M(6, 13) HS_L(0, 7) HS_L(1, 8) HS_L(2, 9)
          HS_L(3, 10) HS_L(4, 11) HS_L(5, 12) HS_L(6, 13) HS_R(0, 7) HS_R(1, 8) HS_R(2, 9) HS_R(3, 10) HS_R(4, 11)
              HS_R(5, 12) HS_R(6, 13)
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), "v"(in[4]), "v"(in[5]), "v"(in[6]), [g0] "v"(t.g0),
        [g1] "v"(t.g1));
}
// unweighted: o[k] = in[k] + its two lane neighbours (the 3-tap half of avg_pool 3x3)
__device__ __forceinline__ void nsum3(const float (&in)[3], float (&o)[3]) {
  asm("v_mov_b32 %0, %3\nv_mov_b32 %1, %4\nv_mov_b32 %2, %5\n"
      "v_add_f32_dpp %0, %3, %0" SFM_DPP_L "v_add_f32_dpp %1, %4, %1" SFM_DPP_L "v_add_f32_dpp %2, %5, %2" SFM_DPP_L
      "v_add_f32_dpp %0, %3, %0" SFM_DPP_R "v_add_f32_dpp %1, %4, %1" SFM_DPP_R "v_add_f32_dpp %2, %5, %2" SFM_DPP_R
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]));
}
// ---- vertical 3-tap sums in scatter form ------------------------------------------------------------------------
// A row's horizontal sum h is pushed into two running accumulators per quantity (p1: the window centred on the previous
// row, complete after this push; p0: the window centred on this row) instead of being kept in a 3-row ring: the state
// carried from row to row is 2 registers per quantity instead of 3, which is what lets another wave fit on the SIMD.
// Returns the completed sum g1 h[r-2] + g0 h[r-1] + g1 h[r] for row r-1.
__device__ __forceinline__ float vpush(float& p1, float& p0, float h) {
  const float out = __builtin_fmaf(kG1, h, p1);
  p1 = __builtin_fmaf(kG0, h, p0);
  p0 = kG1 * h;
  return out;
}
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
__device__ __forceinline__ Item split_item(long item, int n_strips, int n_chunks) {
  const unsigned u = (unsigned)item, strips = (unsigned)n_strips, chunks = (unsigned)n_chunks;
  Item it;
  const unsigned r = u / strips;
  it.strip = (int)(u - r * strips);
  const unsigned q = r / chunks;
  it.chunk = (int)(r - q * chunks);
  it.img = (int)q;
  return it;
}


// ------------------------------------------------------------------ splat side
// x = Sp_i = Acc.rgb / (Acc.w + 1e-10) (utils.py:463,470), y = src_i.  Sums A = sum m |x - y|,
// M = sum m, V = sum ssim(x, y) with m = avgpool3x3_SAME(1 - [x == 0]) (utils.py:478-479).
//
// Per step (image row r): the row's values and their horizontal sums are pushed into the vertical accumulators, which
// completes the windowed moments of row q = r - 1 -> SSIM value, mask, sums and derivative coefficients of row q; the
// coefficients' horizontal sums are pushed in turn, which completes the transposed window of row p = r - 2 -> seeds.
// The raw values of the last three rows sit in 3-slot rings indexed by (row mod 3); the row loop is unrolled by 3 with
// the phase as a compile-time constant, so sliding a ring is a renaming, not a move.
// The wave's constants and running state live in a struct with force-inlined member functions (not in lambdas
// capturing by reference: with the closure form the compiler inlined the bodies only after its last scalar-replacement
// pass and left every captured local - all the state below - in scratch memory, 7x slower).
template <bool GRAD>
struct SplatWave {
  LossDims d;
  int gx, y0, y1, r_first, r_last;
  bool col_in, own_col;
  const float* __restrict__ acc;
  const float* __restrict__ ovf;
  const float* __restrict__ sp;
  float* __restrict__ hb;
  int C3;
  float kL1, kV0;
  Taps taps;
  float P1[3][4], P0[3][4];  // vertical accumulators of the moments x, y, xx + yy, xy per channel
  float N1[3], N0[3];        // ... of the non-zero counts
  float C1[3][3], C0[3][3];  // ... of the derivative coefficients (a, b, c)
  float X[3][3], Y[3][3], Iv[3], Mk[3][3];  // rings, slot = (row - r_first) mod 3
  float sA, sM, sV;
  float na[4], ns[3];  // software prefetch of the next row's raw values (plain floats: a float4 member is assigned by
                       // memcpy, which kept the whole struct in scratch)

  __device__ __forceinline__ void fetch(int r) {
    na[0] = na[1] = na[2] = na[3] = 0.f;
    ns[0] = ns[1] = ns[2] = 0.f;
    if (r >= 0 && r < d.H && col_in) {
      const size_t p = (size_t)r * d.W + gx;
      const float4 t = *reinterpret_cast<const float4*>(acc + p * 4);
      na[0] = t.x; na[1] = t.y; na[2] = t.z; na[3] = t.w;
      if (ovf) {
        const float4 o = *reinterpret_cast<const float4*>(ovf + p * 4);
        na[0] += o.x; na[1] += o.y; na[2] += o.z; na[3] += o.w;
      }
      const f3 u = ld3(sp + p * C3);
      ns[0] = u.x; ns[1] = u.y; ns[2] = u.z;
    }
  }

  // S0: ring slot of row r.  rc3 / rc2 / rc1 = 1 / (taps of the 3x3 SAME window inside the image) for 3, 2, 1 rows in
  // bounds: by value - a select between struct MEMBERS becomes a select between their addresses, which keeps the whole
  // struct (all the wave's state) in scratch memory
  template <int S0>
  __device__ __forceinline__ void step(int r, float rc3, float rc2, float rc1) {
    constexpr int S1 = (S0 + 2) % 3, S2 = (S0 + 1) % 3;  // slots of rows r-1, r-2
    const float ca[4] = {na[0], na[1], na[2], na[3]}, yv[3] = {ns[0], ns[1], ns[2]};
    if (r + 1 <= r_last) fetch(r + 1);
    // ---- row r: values (0 outside the image: the prefetch loaded nothing), horizontal sums, vertical push
    const float inv = (col_in && r >= 0 && r < d.H) ? fast_rcp(ca[3] + 1e-10f) : 0.f;
    Iv[S0] = inv;
    const float xv[3] = {ca[0] * inv, ca[1] * inv, ca[2] * inv};
    float mom[3][4], nz[3], nzh[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      X[S0][ch] = xv[ch];
      Y[S0][ch] = yv[ch];
      nz[ch] = (xv[ch] == 0.0f) ? 0.0f : 1.0f;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#ifdef SFM_ABL_SHARED_MOMENTS  // timing ablation only (wrong numbers): the window moments of src as if precomputed once per step
      const float in3[3] = {xv[ch], xv[ch] * xv[ch] + yv[ch] * yv[ch], xv[ch] * yv[ch]};
      float h3[3];
      hsum3(in3, h3, taps);
      mom[ch][0] = vpush(P1[ch][0], P0[ch][0], h3[0]);
      mom[ch][1] = Y[S1][ch];
      mom[ch][2] = vpush(P1[ch][2], P0[ch][2], h3[1]);
      mom[ch][3] = vpush(P1[ch][3], P0[ch][3], h3[2]);
#else
      const float in4[4] = {xv[ch], yv[ch], xv[ch] * xv[ch] + yv[ch] * yv[ch], xv[ch] * yv[ch]};
      float h4[4];
      hsum3(in4, h4, taps);
#pragma unroll
      for (int m = 0; m < 4; ++m) mom[ch][m] = vpush(P1[ch][m], P0[ch][m], h4[m]);
#endif
    }
    nsum3(nz, nzh);
    // ---- row q = r - 1: windowed moments -> SSIM value, derivative coefficients, mask, sums
    const int q = r - 1;
    if (r - r_first >= 2) {
      const bool pix_in = q >= 0 && q < d.H && col_in;
      const bool own = own_col && q >= y0 && q < y1;
      const int rin = min(q + 1, d.H - 1) - max(q - 1, 0) + 1;
      const float rcnt = rin == 3 ? rc3 : (rin == 2 ? rc2 : rc1);
      float cf[3][3], hc[3][3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float ca_ = 0.f, cb_ = 0.f, cc_ = 0.f;
        const float v = ssim_point<GRAD>(mom[ch][0], mom[ch][1], mom[ch][2], mom[ch][3], ca_, cb_, cc_);
        const float m = (N1[ch] + nzh[ch]) * rcnt;
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
      // ---- row p = r - 2: the transposed window of the coefficients -> backward seeds
      if constexpr (GRAD) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) hsum3(cf[ch], hc[ch], taps);
        float rr[3][3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int m = 0; m < 3; ++m) rr[ch][m] = vpush(C1[ch][m], C0[ch][m], hc[ch][m]);
        const int p = r - 2;
        if (own_col && p >= y0 && p < y1) {
          float h[3], wh = 0.f;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float xq = X[S2][ch], yq = Y[S2][ch];
            const float g = rr[ch][0] + 2.0f * rr[ch][1] * xq + rr[ch][2] * yq;
            h[ch] = (kL1 * Mk[S2][ch] * sgn(xq - yq) + kV0 * g) * Iv[S2];
            wh += h[ch] * xq;
          }
          *reinterpret_cast<float4*>(hb + ((size_t)p * d.W + gx) * 4) = make_float4(h[0], h[1], h[2], wh);
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {  // slide the count window
      N1[ch] = N0[ch] + nzh[ch];
      N0[ch] = nzh[ch];
    }
  }
};

// REDO: the seeds of one item again with kV0 = 0 (its clip did not pass); no sums are written
template <bool GRAD, bool REDO = false>
__device__ __forceinline__ void splat_item(const StencilArgs& a, long item, float* __restrict__ wavep_out,
                                           const double* __restrict__ partials = nullptr) {
  SplatWave<GRAD> w;
  const LossDims d = a.d;
  w.d = d;
  // launch order: the N sources of one strip are consecutive waves (same workgroup, same time): `src` is channel-
  // interleaved over the sources ([.., 3N]), so source i's 12 bytes pull the whole 12N-byte pixel through the caches - with
  // the sources side by side those lines leave HBM once instead of N times.  The wave's slot in `wavep` keeps the
  // (scale, source)-major order the reduction sums in.
  const unsigned un = (unsigned)d.N, uw = (unsigned)item;
  const int i = (int)(uw % un);
  Item it = split_item((long)(uw / un), a.p.strips, a.p.chunks);  // it.img = s * B + b here
  const int b = it.img % d.B, s_ = it.img / d.B;
  it.img = (s_ * d.N + i) * d.B + b;
  if constexpr (REDO) {
    if (partials && loss_clip_passes(partials[s_ * (3 * d.N + 1) + 3 * i + 2], a.inv_n)) return;  // wave-uniform
  }
  wavep_out += ((size_t)it.img * a.p.per_img + (size_t)it.chunk * a.p.strips + it.strip) * kWaveTerms;
  const int lane = threadIdx.x & 63;
  const int HW = d.H * d.W;
  w.C3 = 3 * d.N;
  w.gx = it.strip * kStripOwn - 2 + lane;
  w.col_in = w.gx >= 0 && w.gx < d.W;
  w.own_col = lane >= 2 && lane < 62 && w.gx < d.W;
  w.y0 = it.chunk * a.p.rows;
  w.y1 = min(w.y0 + a.p.rows, d.H);
  w.acc = a.Acc + (size_t)it.img * HW * 4;
  // side accumulator: only if one of the splat tiles under this strip (halo included) holds overflow contributions.
  // Wave-uniform: the strip's first column and its rows are.
  bool any_ovf = false;
  {
    const int c0 = max(it.strip * kStripOwn - 2, 0), c1 = min(it.strip * kStripOwn + 61, d.W - 1);
    const int r0 = max(w.y0 - 2, 0), r1 = min(w.y1 + 1, d.H - 1);
    const int* __restrict__ fl = a.flags + (size_t)it.img * (a.stiles_x * a.stiles_y);
    for (int ty = r0 / STH; ty <= r1 / STH; ++ty)
      for (int tx = c0 / STW; tx <= c1 / STW; ++tx) any_ovf |= fl[ty * a.stiles_x + tx] != 0;
  }
  w.ovf = any_ovf ? a.Ovf + (size_t)it.img * HW * 4 : nullptr;
  w.sp = a.src + (size_t)b * HW * w.C3 + 3 * i;
  w.hb = a.Hb + (size_t)it.img * HW * 4;
  w.kV0 = 0.0f;
  if constexpr (GRAD && !REDO) {
    double M;
    if (a.msum) {
      M = a.msum[s_ * d.N + i];
    } else {
      // single device: the tile shares of this (scale, source) - B * tiles consecutive floats - added up here, by every
      // wave in the same order (the fp64 sum of these fp32 values is exact anyway: loss_msum_reduce_kernel)
      const int per = d.B * a.stiles_x * a.stiles_y;
      const float* __restrict__ mt = a.msumt + (size_t)(s_ * d.N + i) * per;
      M = 0.0;
      for (int k = lane; k < per; k += 64) M += (double)mt[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) M += __shfl_xor(M, o, 64);
    }
    w.kV0 = loss_kv0(M, a.inv_n);
  }
  const float nx = (float)(min(w.gx + 1, d.W - 1) - max(w.gx - 1, 0) + 1);
  // 1 / (taps of the 3x3 SAME window inside the image): k * fl(1/k) == 1 exactly for k = 4, 6, 9
  const float rc3 = 1.0f / (3.0f * nx), rc2 = 1.0f / (2.0f * nx), rc1 = 1.0f / nx;
  w.kL1 = a.kL1;
  w.taps = Taps{kG0, kG1};
  constexpr int HAL = GRAD ? 2 : 1;
  w.r_first = w.y0 - HAL;
  w.r_last = w.y1 - 1 + HAL;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    w.N1[ch] = w.N0[ch] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) w.P1[ch][m] = w.P0[ch][m] = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) w.C1[ch][m] = w.C0[ch][m] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    w.Iv[k] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) w.X[k][ch] = w.Y[k][ch] = w.Mk[k][ch] = 0.f;
  }
  w.sA = w.sM = w.sV = 0.f;

  w.fetch(w.r_first);
  for (int r = w.r_first; r <= w.r_last; r += 3) {
    w.template step<0>(r, rc3, rc2, rc1);
    if (r + 1 <= w.r_last) w.template step<1>(r + 1, rc3, rc2, rc1);
    if (r + 2 <= w.r_last) w.template step<2>(r + 2, rc3, rc2, rc1);
  }
  if constexpr (REDO) return;
  const float sA = wave_sum_dpp_lane63(w.sA), sM = wave_sum_dpp_lane63(w.sM), sV = wave_sum_dpp_lane63(w.sV);
  if (lane == 63) *reinterpret_cast<float4*>(wavep_out) = make_float4(sA, sM, sV, 0.f);
}

// ------------------------------------------------------------------ gather side
// e_i = alpha |G_i - tgt| + (1 - alpha) clip((1 - ssim(G_i, tgt)) / 2, 0, 1)  (:137,142), E = sum min(e_0, e_1)
// (:168).  dG_i = kE [ alpha sel_i sgn(G_i - tgt) + (1 - alpha)(-1/2) (ssim backward of the selected side) ].
template <bool GRAD>
__device__ __forceinline__ void gather_item(const StencilArgs& a, long item, float* __restrict__ wavep_out) {
  const LossDims d = a.d;
  const Item it = split_item(item, a.p.strips, a.p.chunks_g);
  const int lane = threadIdx.x & 63;
  const int s = it.img / d.B, b = it.img - s * d.B;
  const int HW = d.H * d.W;
  const int gx = it.strip * kStripOwn - 2 + lane;
  const bool col_in = gx >= 0 && gx < d.W;
  const bool own_col = lane >= 2 && lane < 62 && gx < d.W;
  const int y0 = it.chunk * a.p.rows_g, y1 = min(y0 + a.p.rows_g, d.H);
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
  const Taps taps{kG0, kG1};

  float P1[3][7], P0[3][7];  // vertical accumulators of the moments t, x0, x1, x0x0 + tt, x1x1 + tt, x0t, x1t per channel
  float C1[3][6], C0[3][6];  // ... of the coefficients (a0, b0, c0, a1, b1, c1), already scaled by the selection
  float X0[3][3], X1[3][3], T[3][3], Sel[3][3];  // rings, slot = (row - r_first) mod 3
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
    for (int m = 0; m < 7; ++m) P1[ch][m] = P0[ch][m] = 0.f;
#pragma unroll
    for (int m = 0; m < 6; ++m) C1[ch][m] = C0[ch][m] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) X0[k][ch] = X1[k][ch] = T[k][ch] = Sel[k][ch] = 0.f;
  }
  float sE = 0.f;

  f3 n0{0.f, 0.f, 0.f}, n1{0.f, 0.f, 0.f}, nt{0.f, 0.f, 0.f};
  auto fetch = [&](int r) __attribute__((always_inline)) {
    n0 = n1 = nt = f3{0.f, 0.f, 0.f};
    if (r >= 0 && r < d.H && col_in) {
      const size_t p = ((size_t)r * d.W + gx) * 3;
      n0 = ld3(g0p + p);
      n1 = ld3(g1p + p);
      nt = ld3(tp + p);
    }
  };

  auto step = [&](int r, auto ph) __attribute__((always_inline)) {
    constexpr int S0 = decltype(ph)::value;
    constexpr int S1 = (S0 + 2) % 3, S2 = (S0 + 1) % 3;
    const float x0[3] = {n0.x, n0.y, n0.z}, x1[3] = {n1.x, n1.y, n1.z}, t[3] = {nt.x, nt.y, nt.z};  // 0 outside the image
    if (r + 1 <= r_last) fetch(r + 1);
    float mm[3][7];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      X0[S0][ch] = x0[ch]; X1[S0][ch] = x1[ch]; T[S0][ch] = t[ch];
      const float tt = t[ch] * t[ch];
#ifdef SFM_ABL_SHARED_MOMENTS  // timing ablation only (wrong numbers): the window moments of tgt as if precomputed once per step
      const float in6[6] = {x0[ch], x1[ch], x0[ch] * x0[ch] + tt, x1[ch] * x1[ch] + tt, x0[ch] * t[ch], x1[ch] * t[ch]};
      float h6[6];
      hsum3(in6, h6, taps);
      mm[ch][0] = T[S1][ch];
#pragma unroll
      for (int m = 0; m < 6; ++m) mm[ch][m + 1] = vpush(P1[ch][m + 1], P0[ch][m + 1], h6[m]);
#else
      const float in7[7] = {t[ch], x0[ch], x1[ch], x0[ch] * x0[ch] + tt, x1[ch] * x1[ch] + tt, x0[ch] * t[ch], x1[ch] * t[ch]};
      float h7[7];
      hsum3(in7, h7, taps);
#pragma unroll
      for (int m = 0; m < 7; ++m) mm[ch][m] = vpush(P1[ch][m], P0[ch][m], h7[m]);
#endif
    }
    const int q = r - 1;
    if (r - r_first >= 2) {
      const bool pix_in = q >= 0 && q < d.H && col_in;
      const bool own = own_col && q >= y0 && q < y1;
      float cf[3][6];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float a0 = 0.f, b0 = 0.f, c0 = 0.f, a1 = 0.f, b1 = 0.f, c1 = 0.f;
        const float v0 = ssim_point<GRAD>(mm[ch][1], mm[ch][0], mm[ch][3], mm[ch][5], a0, b0, c0);
        const float v1 = ssim_point<GRAD>(mm[ch][2], mm[ch][0], mm[ch][4], mm[ch][6], a1, b1, c1);
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
        float rr[3][6];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          float hc[6];
          hsum3(cf[ch], hc, taps);
#pragma unroll
          for (int m = 0; m < 6; ++m) rr[ch][m] = vpush(C1[ch][m], C0[ch][m], hc[m]);
        }
        const int p = r - 2;
        if (own_col && p >= y0 && p < y1) {
          float d0[3], d1[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float tv = T[S2][ch], xa = X0[S2][ch], xb = X1[S2][ch];
            d0[ch] = kl * Sel[S2][ch] * sgn(xa - tv) + (rr[ch][0] + 2.0f * rr[ch][1] * xa + rr[ch][2] * tv);
            d1[ch] = kl * (1.0f - Sel[S2][ch]) * sgn(xb - tv) + (rr[ch][3] + 2.0f * rr[ch][4] * xb + rr[ch][5] * tv);
          }
          const size_t pp = ((size_t)p * d.W + gx) * 3;
          st3(a.dG + o0 + pp, d0[0], d0[1], d0[2]);
          st3(a.dG + o1 + pp, d1[0], d1[1], d1[2]);
        }
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
__global__ __launch_bounds__(kBlock) SFM_WPE_GATHER void loss_stencil_gather_kernel(StencilArgs a, int gather_blocks) {
  if ((int)blockIdx.x >= gather_blocks) {  // the extra workgroups of a single-device launch: mask-sum fix-up, one wave per tile
    const long t = (long)(blockIdx.x - gather_blocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int ntile = a.stiles_x * a.stiles_y;
    if (t < (long)a.d.S * a.d.N * a.d.B * ntile)
      msum_fix_tile(t, threadIdx.x & 63, a.Acc, a.Ovf, a.flags, a.ovl, a.msumt, a.d.H, a.d.W, a.stiles_x, ntile);
    return;
  }
  const long w = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w < a.p.n_gather) gather_item<GRAD>(a, w, a.wavep + (a.p.n_splat + w) * kWaveTerms);
}
template <bool GRAD>
__global__ __launch_bounds__(kBlock) SFM_WPE_SPLAT void loss_stencil_splat_kernel(StencilArgs a) {
  const long w = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w < a.p.n_splat) splat_item<GRAD>(a, w, a.wavep);  // the item computes its own slot
}

// The guard of the combined seed plane.  A small fixed grid: every wave first looks at the S*N global SSIM sums and
// leaves if every clip passes (always, for real images); otherwise the waves share the items of the failing
// (scale, source) pairs and rewrite their seeds with kV0 = 0.
// One wave per workgroup: the item index is then a scalar from the start (blockIdx), with no v_readfirstlane next to the
// inline-asm DPP blocks (tests/test_build_invariants.py).
constexpr int kRedoWaves = 1024;
__global__ __launch_bounds__(64) SFM_WPE_SPLAT void loss_stencil_splat_redo_kernel(StencilArgs a,
                                                                                     const double* __restrict__ partials) {
  const int nterm = 3 * a.d.N + 1;
  bool any = false;
  for (int k = 0; k < a.d.S * a.d.N; ++k)
    any |= !loss_clip_passes(partials[(k / a.d.N) * nterm + 3 * (k % a.d.N) + 2], a.inv_n);
  if (!any) return;
  for (long w = blockIdx.x; w < a.p.n_splat; w += gridDim.x) splat_item<true, true>(a, w, a.wavep, partials);
}

// per-wave partial sums of the stencil kernels -> partials fp64 [S][3N+1], fixed summation order (deterministic).
// grid (S, 3N+1): workgroup (s, t) owns one term.  GUARD (single device, where these sums are the global ones): the
// workgroup that has formed V(s, i) checks the clip and, in the degenerate case that it does not pass, rewrites the seeds
// of its (scale, source) with kV0 = 0 itself (four waves: slow, and never taken for real images).
template <bool GUARD>
__global__ __launch_bounds__(kBlock) SFM_WPE_SPLAT void loss_reduce_kernel(StencilArgs a, double* __restrict__ partials) {
  __shared__ double red[kBlock];
  const LossDims d = a.d;
  const StencilPlan p = a.p;
  const int s = blockIdx.x, t = blockIdx.y;
  const int nterm = 3 * d.N + 1;
  long per = (long)d.B * p.per_img;  // waves per (scale, source) image set
  long first;
  int comp;
  if (t < 3 * d.N) {
    first = ((long)s * d.N + t / 3) * per;
    comp = t % 3;
  } else {
    per = (long)d.B * p.per_img_g;
    first = p.n_splat + (long)s * per;
    comp = 0;
  }
  double acc = 0.0;
  if (t < 3 * d.N || p.n_gather > 0)
    for (long k = threadIdx.x; k < per; k += kBlock) acc += (double)a.wavep[(first + k) * kWaveTerms + comp];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(size_t)s * nterm + t] = red[0];
  if constexpr (GUARD) {
    if (t < 3 * d.N && comp == 2 && !loss_clip_passes(red[0], a.inv_n)) {  // workgroup-uniform
      const int i = t / 3;
      const long per_s = (long)d.B * p.chunks * p.strips;  // items of one scale and source
      for (long q = threadIdx.x >> 6; q < per_s; q += kBlock / 64)
        splat_item<true, true>(a, ((long)s * per_s + q) * d.N + i, a.wavep, nullptr);
    }
  }
}

}  // namespace

int launch_loss_reduce(const StencilArgs& a, double* partials, bool guard, hipStream_t stream) {
  const dim3 grid(a.d.S, 3 * a.d.N + 1);
  if (guard) hipLaunchKernelGGL(loss_reduce_kernel<true>, grid, dim3(kBlock), 0, stream, a, partials);
  else hipLaunchKernelGGL(loss_reduce_kernel<false>, grid, dim3(kBlock), 0, stream, a, partials);
  return sfm_last_error();
}

int launch_loss_stencil_redo(const StencilArgs& a, const double* partials, hipStream_t stream) {
  hipLaunchKernelGGL(loss_stencil_splat_redo_kernel, dim3(kRedoWaves), dim3(64), 0, stream, a, partials);
  return sfm_last_error();
}

// Both kernels go to the caller's stream, back to back.  Running them side by side (the gather side on a second stream,
// forked and joined with events) was measured at c2 and is SLOWER: 116 us for the pair instead of 96 (gather 93 us
// instead of 43, splat 80 instead of 56; profiles/r02_c2_experiments.txt) - together they want more VALU issue slots
// and cache than the chip has, so they only get in each other's way.
int launch_loss_stencil(const StencilArgs& a, bool need_grad, int which, hipStream_t stream) {
  constexpr int WPB = kBlock / 64;
  const int gb = (int)((a.p.n_gather + WPB - 1) / WPB);
  const dim3 gs((unsigned)((a.p.n_splat + WPB - 1) / WPB)), gg((unsigned)(gb + (need_grad ? a.fix_blocks : 0)));
  const bool gather = (which & kStencilGather) && a.p.n_gather, splat = (which & kStencilSplat) != 0;
  if (need_grad) {
    if (gather) hipLaunchKernelGGL(loss_stencil_gather_kernel<true>, gg, dim3(kBlock), 0, stream, a, gb);
    if (splat) hipLaunchKernelGGL(loss_stencil_splat_kernel<true>, gs, dim3(kBlock), 0, stream, a);
  } else {
    if (gather) hipLaunchKernelGGL(loss_stencil_gather_kernel<false>, gg, dim3(kBlock), 0, stream, a, gb);
    if (splat) hipLaunchKernelGGL(loss_stencil_splat_kernel<false>, gs, dim3(kBlock), 0, stream, a);
  }
  return sfm_last_error();
}

}  // namespace sfm
