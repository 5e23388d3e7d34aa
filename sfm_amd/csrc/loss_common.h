// Shared between the translation units of the fused multi-scale loss (loss.hip: host entry
// points + the two warp kernels; loss_stencil.hip: the streaming L1/SSIM stencil kernel).
#pragma once
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {

constexpr int kMaxSrc = 4;
constexpr float kAlpha = 0.15f;  // SfMLearner_uni.py:84

struct LossDims {
  int B, H, W, S, N;
};

// Where the per-pixel inputs of the warps come from when the caller hands over the prediction PYRAMID
// (pred_depth[s] [B,h_s,w_s], pred_exp_logits[s] [B,h_s,w_s,3N] at their native sizes) instead of maps already resized
// to HxW: the kernels then evaluate tf.image.resize_bilinear (SfMLearner_uni.py:91, 113) per pixel while loading, with
// the arithmetic of the stand-alone resize kernels (sfm_device.h lerp_of / lerp4), and the [S,B,H,W,1+3N] intermediate
// never exists.  A scale at full size is read directly.
constexpr int kMaxScales = 8;
struct PyrMaps {
  const float* depth[kMaxScales];
  const float* trans[kMaxScales];
  int h[kMaxScales], w[kMaxScales];
  float sy[kMaxScales], sx[kMaxScales];  // h / (float)H, w / (float)W (CalculateResizeScale)
  // backward: where the gradients of the pyramid's maps are ADDED (the adjoint of the resize, evaluated in the kernel)
  float* d_depth[kMaxScales];
  float* d_trans[kMaxScales];
};
// LDS footprint of one backward tile (BTW x BTH full-size pixels) on a low-resolution level: every level is at most the
// image's size, so the tile touches at most (BTH + 2) x (BTW + 2) of its cells (lo .. hi of the tile's corner pixels)

// ---- streaming stencil geometry -------------------------------------------------------
// One wave = one strip of 64 image columns (the outer two on each side are halo: 60 owned)
// marching down `rows` image rows (+2 halo rows above and below).  No LDS, no barriers.
constexpr int kStripOwn = 60;
constexpr int kWaveTerms = 4;  // per-wave partial sums: splat (A, M, V, -), gather (E, -, -, -)

struct StencilPlan {
  int strips;     // per image row
  int rows;       // image rows owned by one wave of the splat-side kernel
  int chunks;     // per image column
  int per_img;    // strips * chunks
  int rows_g, chunks_g, per_img_g;  // the same for the gather-side kernel (its own register budget and occupancy)
  long n_splat, n_gather;  // work items (waves)
};

// Rows per wave.  More rows = fewer halo rows per owned row ((rows + 4) / rows row steps), fewer rows = more waves.  A wave
// issues at most one VALU instruction every ~6.4 cycles while a SIMD takes one every ~2.6 (tools/valu_bench.hip), so what
// counts on a small problem is how the waves fill the resident slots (1024 SIMDs x `waves_per_simd`).  Measured at c2
// (sweeps of SFM_STENCIL_ROWS / SFM_STENCIL_ROWS_G):
//   * large problems (>= 6 rounds of resident waves even with long strips): the longest strips that keep 6 rounds;
//   * small ones, `fill` set (gather side, 2 waves per SIMD): the strips that fill the resident slots exactly once - 2016
//     waves on 2048 slots at c2, rows = 15: 98 us for the pair of kernels where rows = 8 took 104 and rows = 19 106;
//   * small ones otherwise (splat side, 3 waves per SIMD): 8 rows - flat from 8 to 13, worse beyond.
static inline int stencil_rows(long wave_columns, int H, int waves_per_simd, bool fill) {
  const long slots = 1024L * waves_per_simd;
  auto waves = [&](int rows) { return wave_columns * ((H + rows - 1) / rows); };
  auto even = [&](int rows) { const int c = (H + rows - 1) / rows; return (H + c - 1) / c; };  // same chunks, equal rows
  if (waves(8) >= 6 * slots) {
    int rows = 64;
    while (rows > 8 && waves(rows) < 6 * slots) --rows;
    return even(rows);
  }
  if (fill) {
    const long chunks = slots / (wave_columns > 0 ? wave_columns : 1);
    if (chunks >= 1) {
      const int rows = (int)((H + chunks - 1) / chunks);
      if (rows >= 8 && rows <= 24) return even(rows);
    }
  }
  return H < 8 ? (H > 0 ? H : 1) : 8;
}

static inline StencilPlan stencil_plan(const LossDims& d, int rows_override, int rows_g_override) {
  StencilPlan p;
  p.strips = (d.W + kStripOwn - 1) / kStripOwn;
  const long cols_s = (long)d.S * d.N * d.B * p.strips, cols_g = (long)d.S * d.B * p.strips;
  p.rows = rows_override > 0 ? rows_override : stencil_rows(cols_s, d.H, 3, false);    // 146 VGPRs: 3 waves per SIMD
  p.rows_g = rows_g_override > 0 ? rows_g_override : stencil_rows(cols_g, d.H, 2, true);  // 194 VGPRs: 2
  p.chunks = (d.H + p.rows - 1) / p.rows;
  p.chunks_g = (d.H + p.rows_g - 1) / p.rows_g;
  p.per_img = p.strips * p.chunks;
  p.per_img_g = p.strips * p.chunks_g;
  p.n_splat = (long)d.S * d.N * d.B * p.per_img;
  p.n_gather = d.N >= 2 ? (long)d.S * d.B * p.per_img_g : 0;
  return p;
}

// ---- workspace map (floats) -------------------------------------------------------------
//   Acc [S,N,B,H,W,4]   splat accumulators (sum g*rgb, sum g), one 16-byte unit per pixel
//   Ovf [S,N,B,H,W,4]   overflow side accumulator (all zero between steps)
//   Hb  [S,N,B,H,W,4]   backward seed per destination pixel: (h[3], sum_c h*Sp), h = hA + kV0 * hB (loss_stencil.hip)
//   G   [S,2,B,H,W,3]   gathered sources 0,1
//   dG  [S,2,B,H,W,3]   backward: dL/dG_i (gloss = 1)
//   wavep [n_waves][4]  per-wave partial sums of the stencil kernel
//   dPp [S,N,B,nblk,12] dL/dP block partials of the backward warp
//   msumt [n_img][tiles] per splat tile: sum of nz(q) * wgt(q) = the tile's share of the mask sum M (see loss_kv0)
//   msum  [S*N] fp64     M per (scale, source) on this rank (the single-device step reads it back as the global one)
//   ovl [n_img][tiles][STW*STH] uint16: per splat tile, the pixels (index inside the tile) that received overflow
//                        contributions - each listed once, by the adder that found Ovf.w == 0; `flags` holds the count
#ifndef SFM_STH
#define SFM_STH 32
#endif
constexpr int STW = 64, STH = SFM_STH;  // splat: destination tile of one workgroup (LDS: 4 fp64 planes)
#ifndef SFM_SPLAT_PPT
#define SFM_SPLAT_PPT 4
#endif
constexpr int kSplatThreads = STW * STH / SFM_SPLAT_PPT;  // home pixels per thread
#ifndef SFM_BTH
#define SFM_BTH 4  // 64x4 = one pixel per thread: no loop-carried prefetch state, 84 VGPRs = five waves per SIMD with both
                   // sources' gathers in flight (SFM_BWD_PAIR 2): 54 -> 50 us at c2 against 64x8 tiles at 132 VGPRs
#endif
constexpr int BTW = 64, BTH = SFM_BTH;  // backward warp: pixel tile of one workgroup
constexpr int kLossPPT = BTW * BTH / kBlock;  // pixels per thread in the backward warp kernel
static_assert(BTW == STW && STH % BTH == 0, "a backward tile lies inside one splat tile (per-tile overflow flags)");

struct LossWs {
  size_t G, Acc, Ovf, Hb, dG, wavep, dPp, coef, msumt, msum, ovl, shift, flags, total;
  int nblk_bwd, btiles_x, btiles_y, stiles_x, stiles_y, n_img;
  long n_flags;  // one overflow counter per (image, splat tile): 0 = the tile's Acc is complete
};

static inline size_t up4(size_t v) { return (v + 3) & ~(size_t)3; }

static inline LossWs loss_ws(const LossDims& d, const StencilPlan& sp) {
  LossWs w;
  const size_t npx = (size_t)d.B * d.H * d.W;
  const int ng = d.N < 2 ? d.N : 2;
  w.stiles_x = (d.W + STW - 1) / STW;
  w.stiles_y = (d.H + STH - 1) / STH;
  w.btiles_x = (d.W + BTW - 1) / BTW;
  w.btiles_y = (d.H + BTH - 1) / BTH;
  w.nblk_bwd = w.btiles_x * w.btiles_y;
  w.n_img = d.S * d.N * d.B;
  size_t o = 0;
  w.Acc = o; o += (size_t)d.S * d.N * npx * 4;
  w.Ovf = o; o += (size_t)d.S * d.N * npx * 4;
  w.Hb = o;  o += (size_t)d.S * d.N * npx * 4;
  w.G = o;   o += up4((size_t)d.S * ng * npx * 3);
  w.dG = o;  o += up4((size_t)d.S * ng * npx * 3);
  w.wavep = o; o += up4((size_t)(sp.n_splat + sp.n_gather) * kWaveTerms);
  w.dPp = o; o += up4((size_t)d.S * d.N * d.B * w.nblk_bwd * 12);
  w.coef = o; o += 64;
  w.n_flags = (long)w.n_img * w.stiles_x * w.stiles_y;
  w.msumt = o; o += up4((size_t)w.n_flags);
  w.msum = o; o += up4((size_t)d.S * d.N * 2);            // doubles (every offset here is a multiple of 4 floats)
  w.ovl = o; o += (size_t)w.n_flags * (STW * STH / 2);   // uint16 entries
  w.shift = o; o += (size_t)w.n_img * w.stiles_x * w.stiles_y * 4;
  w.flags = o; o += up4((size_t)w.n_flags);
  w.total = o;
  return w;
}

struct StencilArgs {
  const float* tgt;
  const float* src;
  const float* G;
  const float* Acc;
  const float* Ovf;
  const int* flags;  // [n_img][stiles_y][stiles_x]: the splat tile holds overflow contributions in Ovf
  int stiles_x, stiles_y;
  float* Hb;
  float* dG;
  float* wavep;
  LossDims d;
  StencilPlan p;
  float kL1;  // alpha / n_elem_global          (coefficient of the masked L1 term, :149)
  float kE;   // 2 / n_elem_global              (coefficient of the min-reprojection term, :168)
  const double* msum;  // [S*N] GLOBAL mask sums M(s, i) = sum m (all ranks), known before the splat-side stencil runs;
                       // null (single device): every splat-side wave adds up the tile shares of its (scale, source) itself
  double inv_n;        // 1 / n_elem_global
  float* msumt;        // [n_img][tiles] the tile shares (loss_splat_tile_kernel; corrected by msum_fix_tile)
  const unsigned short* ovl;  // the listed overflow pixels per tile
  int fix_blocks;      // > 0 (single device): the gather-side launch carries that many extra workgroups, which run the
                       // mask-sum fix-up of the tiles (msum_fix_tile) beside the gather-side strips - one launch less
};

// Coefficient of the SSIM part of the splat-side seeds: d/dSp of (1 - alpha) mean(m) clip((1 - mean ssim) / 2, 0, 1)
// (SfMLearner_uni.py:144, 149; utils.py:573-575) = kV0 * d ssim / dSp with kV0 = (1 - alpha) (M / n) (-1/2) / n while the
// clip passes.  M = sum m = sum_q nz(q) wgt(q) (utils.py:478-479: m is the 3x3 SAME average of nz = 1 - [Sp == 0], so its
// sum needs no window: wgt(q) = sum over the windows containing q of 1 / their in-bounds tap count, position only) is
// taken where the splat tiles are written, BEFORE the stencil - so the stencil can write ONE combined seed plane
// h = hA + kV0 hB instead of the two parts.  Whether the clip passes is only known after the stencil (it depends on the
// global mean SSIM); it fails only for degenerate inputs (mean SSIM outside [-1, 1]); then the affected seeds are
// rewritten with kV0 = 0 before the backward reads them (loss_stencil_splat_redo_kernel).
__device__ __forceinline__ float loss_kv0(double M, double inv_n) {
  return (float)((1.0 - (double)kAlpha) * (M * inv_n) * (-0.5) * inv_n);
}
__device__ __forceinline__ bool loss_clip_passes(double V, double inv_n) {
  const float h = (1.0f - (float)(V * inv_n)) / 2.0f;
  return h >= 0.0f && h <= 1.0f;
}
// sum over the 3-tap SAME windows (along one axis of n positions) that contain position q of 1 / (their in-bounds taps)
__device__ __forceinline__ float mask_axis_weight(int q, int n) {
  if (q >= 2 && q <= n - 3) return 1.0f;  // three full windows: fl(1/3) * 3 rounds to 1
  float s = 0.0f;
  for (int p = max(q - 1, 0); p <= min(q + 1, n - 1); ++p) s += 1.0f / (float)(min(p + 1, n - 1) - max(p - 1, 0) + 1);
  return s;
}
// number of non-zero channels of Sp = acc.rgb / (acc.w + 1e-10) (utils.py:463, 470, 478), with the stencil's arithmetic
__device__ __forceinline__ float nz_count(float r, float g, float b, float w) {
  const float inv = __builtin_amdgcn_rcpf(w + 1e-10f);
  return ((r * inv == 0.0f) ? 0.0f : 1.0f) + ((g * inv == 0.0f) ? 0.0f : 1.0f) + ((b * inv == 0.0f) ? 0.0f : 1.0f);
}

// The mask-sum share of a tile that received overflow contributions: its Acc alone was not the final accumulator when
// the tile's owner counted the non-zero pixels.  One wave per (image, tile) w; returns at once unless the tile's counter
// is raised, else walks the listed pixels (each once) and adds [nz(Acc + Ovf) - nz(Acc)] * wgt to the tile's share.
__device__ __forceinline__ void msum_fix_tile(long w, int lane, const float* __restrict__ Acc, const float* __restrict__ Ovf,
                                              const int* __restrict__ flags, const unsigned short* __restrict__ ovl,
                                              float* __restrict__ msumt, int H, int W, int tiles_x, int ntile) {
  constexpr int TP = STW * STH;
  const int cnt = min(flags[w], TP);
  if (cnt == 0) return;
  const long img = w / ntile;
  const int tile = (int)(w - img * ntile);
  const int x0 = (tile % tiles_x) * STW, y0 = (tile / tiles_x) * STH;
  const float4* __restrict__ a = reinterpret_cast<const float4*>(Acc) + (size_t)img * H * W;
  const float4* __restrict__ o = reinterpret_cast<const float4*>(Ovf) + (size_t)img * H * W;
  float dsum = 0.0f;
  for (int e = lane; e < cnt; e += 64) {
    const int l = ovl[(size_t)w * TP + e];
    const int u = x0 + (l & (STW - 1)), v = y0 + l / STW;
    const size_t p = (size_t)v * W + u;
    const float4 av = a[p], ov = o[p];
    const float before = nz_count(av.x, av.y, av.z, av.w);
    const float after = nz_count(av.x + ov.x, av.y + ov.y, av.z + ov.z, av.w + ov.w);  // the stencil's Acc + Ovf
    dsum += (after - before) * (mask_axis_weight(u, W) * mask_axis_weight(v, H));
  }
  dsum = wave_sum_dpp_lane63(dsum);
  if (lane == 63) msumt[w] += dsum;
}

// loss_stencil.hip.  which: the gather side (min-reprojection term), the splat side (masked L1 + SSIM terms), or both
constexpr int kStencilGather = 1, kStencilSplat = 2;
int launch_loss_stencil(const StencilArgs& a, bool need_grad, int which, hipStream_t stream);
// rewrites the seeds of every (scale, source) whose clip does not pass (global sums in partials) with kV0 = 0; exits at
// once otherwise (the normal case)
int launch_loss_stencil_redo(const StencilArgs& a, const double* partials, hipStream_t stream);
// per-wave partial sums of the stencil kernels -> partials fp64 [S][3N+1], fixed summation order (deterministic).
// guard (single device: the sums are already global): the workgroup that has just formed V(s, i) applies the seed guard
// to that (scale, source) itself, so the backward needs no guard launch.
int launch_loss_reduce(const StencilArgs& a, double* partials, bool guard, hipStream_t stream);

}  // namespace sfm
