// Fused multi-scale photometric loss of SfMLearner_uni.py:85-168, forward and backward,
// for S scales x N sources in a fixed number of launches (independent of S and N):
//
//   sfm_loss_forward    tile_shift -> splat_tile (both warps) -> stencil (loss_stencil.hip) -> reduce
//   sfm_loss_backward   finalize -> warp_bwd -> pose_grad
//
// Between the two halves sits the only cross-GPU exchange of the data-parallel path:
// an all-reduce of the S*(3N+1) fp64 partial sums; tf_ssim's
// global mean (utils.py:573-575) is taken before its non-linear clip, so the sums
// must be global before the loss VALUE exists.  The backward coefficient of the SSIM part
// needs only the global mask sums M(s, i) (loss_kv0, loss_common.h), and those are known
// before the stencil runs: the splat tiles count their non-zero pixels where they write
// Acc (msumt), loss_msum_fix_kernel corrects the tiles that received overflow
// contributions, loss_msum_reduce_kernel adds them up.  Under data parallelism the S*N
// mask sums are all-reduced while the gather-side stencil runs (sfm_loss_forward_warps /
// _gather / _splat); the splat-side stencil then writes ONE combined seed plane.
//
// Per (scale s, source i), reference lines in SfMLearner_uni.py:
//   G_i      = projective_inverse_warp(src_i, depth_s, pose_i, K, trans_si)          :120
//   Sp_i, m  = projective_warp(tgt, depth_s, pose_i, K, trans_si)                    :128
//   e_i      = a |G_i - tgt| + (1-a) clip((1 - ssim(G_i, tgt)) / 2, 0, 1)            :137,142
//   pixel   += a mean(m |Sp_i - src_i|) + (1-a) mean(m) clip((1 - mean ssim(Sp_i, src_i)) / 2, 0, 1)   :139,144,149
//   pixel   += 2 mean(min(e_0, e_1))                                                 :168
// Both warps of a pair share one geometry (utils.py:185-202 == 220-237), so it is
// evaluated once per pixel and source; meshgrid / cam / coords never exist in memory.
//
// Data layout in HBM (all fp32, NHWC; workspace map in loss_common.h):
//   tgt [B,H,W,3]   src [B,H,W,3N]   depth [S,B,H,W]   trans [S,B,H,W,3N]
//   G   [S,2,B,H,W,3]     gathered sources 0,1 (the only ones line 168 reads)
//   Acc [S,N,B,H,W,4]     splat accumulators (sum g*rgb, sum g): one 16-B unit per pixel
//   Hb  [S,N,B,H,W,4]     backward seed per destination pixel: (h[3], sum_c h*Sp), h = hA + kV0 hB
//   dG  [S,2,B,H,W,3]     backward: dL/dG_i for gloss = 1
#include "loss_common.h"
#include <cstdlib>

namespace sfm {

constexpr int kCacheTX = 2, kCacheTY = 4, kCacheW = 2 * kCacheTX + 1, kCacheH = 2 * kCacheTY + 1;  // LDS-cached table window (tiles)
// splat: reach of a tile's scan window = ceil(half-range of the tile's displacement) + kReachAdd, clamped.
// Scanning is cheap (~8 us per 1.0 of redundancy at c2) and a flagged image costs its consumers a second
// accumulator read, so the reach is generous.
#ifndef SFM_REACH_ADD
#define SFM_REACH_ADD 4
#endif
#ifndef SFM_REACH_MIN
#define SFM_REACH_MIN 6
#endif
#ifndef SFM_REACH_MAX
#define SFM_REACH_MAX 32  // c4 (Cityscapes intrinsics, 20-90 px shifts): 16 left 6 % of the pixels to the overflow path; 32: splat -23 %
#endif
#ifndef SFM_REACH_MUL
#define SFM_REACH_MUL 1.0f
#endif
constexpr float kReachMul = SFM_REACH_MUL;  // margin proportional to the sampled spread (the variation between samples grows with it)
constexpr int kReachAdd = SFM_REACH_ADD, kReachMin = SFM_REACH_MIN, kReachMax = SFM_REACH_MAX;

// ------------------------------------------------------------------ fused warp forward
// The forward splat (utils.py:419-470) without global atomics in the common case.
//
// A workgroup OWNS a destination tile of STW x STH pixels of one (scale, source, batch item)
// image and keeps its (sum g*rgb, sum g) accumulators in LDS.  It scans the window of target
// pixels that can land in the tile - the tile moved back by the tile's estimated displacement
// and grown by the tile's reach (`shift` = (shift_x, shift_y, reach_x, reach_y), one int4 per tile,
// written by loss_tile_shift_kernel from the spread of the displacement over the tile) -
// evaluates the geometry of each scanned pixel and adds the corners that fall inside its own
// tile with LDS atomics (ds_add_f64).  The tile is then written to Acc with plain 16-byte
// stores: no memset of Acc, no global atomics, every Acc element written exactly once.
//
// Exactness does not depend on the table.  Each target pixel also has a HOME
// workgroup (the one whose tile contains the pixel itself).  For each of the pixel's four
// corners the home workgroup evaluates the same window predicate the owner of the corner's
// tile uses; a corner whose owner does not scan the pixel ("overflow", rare for smooth
// warps) is added by the home workgroup to the zero-initialised side accumulator Ovf with
// global atomics and the image's flag is raised; consumers read Acc + Ovf for flagged
// images.  Every (pixel, corner) contribution is therefore counted exactly once.
// The home workgroup also performs the 4-tap gather of the pixel (utils.py:297-335) -> G.
// scan window (x_lo, x_hi, y_lo, y_hi), half-open, of the owner of tile (tx, ty) with table entry sft
__device__ __forceinline__ int4 scan_rect(int tx, int ty, int4 sft) {
  const int xl = tx * STW - sft.x - sft.z, yl = ty * STH - sft.y - sft.w;
  return make_int4(xl, xl + STW + 2 * sft.z, yl, yl + STH + 2 * sft.w);
}

// One pixel's depth and per-source translation from the level-s maps of the pyramid: the four taps and weights of
// tf.image.resize_bilinear (same arithmetic as pyramid_upsample_fwd_kernel, bit for bit).
struct PyrLevel {
  const float* d;   // depth_s of this batch item
  const float* t;   // trans_s of this batch item, already offset to the source's three channels
  int h, w;
  float sy, sx;
  bool lowres;      // false: the level is at HxW and is read directly
};
__device__ __forceinline__ PyrLevel pyr_level(const PyrMaps& pm, int s, int b, int i, int C3, int H, int W) {
  PyrLevel l;
  l.h = pm.h[s]; l.w = pm.w[s]; l.sy = pm.sy[s]; l.sx = pm.sx[s];
  l.lowres = !(l.h == H && l.w == W);
  l.d = pm.depth[s] + (size_t)b * l.h * l.w;
  l.t = pm.trans[s] + (size_t)b * l.h * l.w * C3 + 3 * i;
  return l;
}
__device__ __forceinline__ void pyr_read(const PyrLevel& l, int C3, int v, int u, float& dep, float& t0, float& t1, float& t2) {
  const Lerp ly = lerp_of(v, l.sy, l.h), lx = lerp_of(u, l.sx, l.w);
  const int o00 = ly.lo * l.w + lx.lo, o01 = ly.lo * l.w + lx.hi, o10 = ly.hi * l.w + lx.lo, o11 = ly.hi * l.w + lx.hi;
  const float d00 = l.d[o00], d01 = l.d[o01], d10 = l.d[o10], d11 = l.d[o11];
  const f3 a = ld3(l.t + (size_t)o00 * C3), bq = ld3(l.t + (size_t)o01 * C3), c = ld3(l.t + (size_t)o10 * C3),
           e = ld3(l.t + (size_t)o11 * C3);
  dep = lerp4(d00, d01, d10, d11, lx.l, ly.l);
  t0 = lerp4(a.x, bq.x, c.x, e.x, lx.l, ly.l);
  t1 = lerp4(a.y, bq.y, c.y, e.y, lx.l, ly.l);
  t2 = lerp4(a.z, bq.z, c.z, e.z, lx.l, ly.l);
}

// Sixteen lanes per (image, tile), nine of them active: mid-range and half-range of the displacement (x - u, y - v)
// over a 3x3 grid of sample pixels of the tile -> shift and reach of the tile's scan window.  One sample per lane, so
// the kernel is one load round trip deep (it ran nine dependent ones when a thread walked the samples); the 16-lane
// min / max reduction is a shuffle butterfly.  A pixel at u with displacement in [lo, hi] has corners
// floor(u + dx) + {0, 1}: reach = half-range + 1.5 covers the sampled spread; the additive margin (kReachAdd) covers
// the variation between samples.  Any table is correct; a good one keeps the overflow rare and the scan small.
template <bool PYR>
__global__ __launch_bounds__(kBlock) void loss_tile_shift_kernel(const float* __restrict__ depth,
                                                                 const float* __restrict__ trans, PyrMaps pm,
                                                                 const float* __restrict__ params, LossDims d,
                                                                 float step_x, float step_y, int tiles_x, int tiles_y,
                                                                 int4* __restrict__ shift, int radd, int rmin, int rmax) {
  const int ntile = tiles_x * tiles_y;
  const long total = (long)d.S * d.N * d.B * ntile;
  const long k = ((long)blockIdx.x * kBlock + threadIdx.x) >> 4;
  const int j = threadIdx.x & 15;  // sample (jy, jx) = (j / 3, j % 3) for j < 9
  const bool tile_ok = k < total;
  const long kc = tile_ok ? k : total - 1;  // idle groups redo the last tile: every lane takes part in the shuffles
  const int tile = (int)(kc % ntile);
  const int img = (int)(kc / ntile);  // (s*N + i)*B + b
  const int b = img % d.B, si = img / d.B, i = si % d.N, s = si / d.N;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int HW = d.H * d.W, C3 = 3 * d.N;
  const float xmax = (float)(d.W - 1), ymax = (float)(d.H - 1);
  const int xa = tx * STW, ya = ty * STH;
  const int xb = min(xa + STW, d.W) - 1, yb = min(ya + STH, d.H) - 1;
  float lo_x = 3.0e38f, hi_x = -3.0e38f, lo_y = 3.0e38f, hi_y = -3.0e38f;
  if (j < 9) {
    const int jy = j / 3, jx = j - 3 * jy;
    const Pose ps = load_pose(params + ((size_t)i * d.B + b) * SFM_PARAM_STRIDE);
    const size_t sb = ((size_t)s * d.B + b) * HW;
    const int u = xa + ((xb - xa) * jx) / 2, v = ya + ((yb - ya) * jy) / 2;
    const int p = v * d.W + u;
    float dep, t0, t1, t2;
    bool direct = true;
    if constexpr (PYR) {
      const PyrLevel lv = pyr_level(pm, s, b, i, C3, d.H, d.W);
      direct = !lv.lowres;
      if (lv.lowres) pyr_read(lv, C3, v, u, dep, t0, t1, t2);
      else {
        const f3 t = ld3(lv.t + (size_t)p * C3);
        dep = lv.d[p]; t0 = t.x; t1 = t.y; t2 = t.z;
      }
    } else {
      dep = depth[sb + p];
      t0 = t1 = t2 = 0.0f;
      if (trans) {  // (null only from the stand-alone splat, add_trans = False)
        const f3 t = ld3(trans + (sb + p) * C3 + 3 * i);
        t0 = t.x; t1 = t.y; t2 = t.z;
      }
    }
    (void)direct;
    const Geom g = geom_fwd(ps, grid_coord(u, step_x, xmax), grid_coord(v, step_y, ymax), dep, PYR || trans != nullptr, t0, t1, t2);
    // displacement of the clipped landing point (what the corners are clipped to); NaNs drop out of fmin / fmax
    lo_x = hi_x = fminf(fmaxf(g.x, 0.0f), xmax) - (float)u;
    lo_y = hi_y = fminf(fmaxf(g.y, 0.0f), ymax) - (float)v;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    lo_x = fminf(lo_x, __shfl_xor(lo_x, o, 16)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o, 16));
    lo_y = fminf(lo_y, __shfl_xor(lo_y, o, 16)); hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o, 16));
  }
  if (j != 0 || !tile_ok) return;
  const float mx = 0.5f * (lo_x + hi_x), my = 0.5f * (lo_y + hi_y);
  const float hx = 0.5f * (hi_x - lo_x), hy = 0.5f * (hi_y - lo_y);
  int4 o;
  o.x = (mx == mx) ? (int)rintf(fminf(fmaxf(mx, -32768.0f), 32767.0f)) : 0;
  o.y = (my == my) ? (int)rintf(fminf(fmaxf(my, -32768.0f), 32767.0f)) : 0;
  o.z = (hx == hx) ? (int)fminf(fmaxf(ceilf(hx * kReachMul) + (float)radd, (float)rmin), (float)rmax) : rmin;
  o.w = (hy == hy) ? (int)fminf(fmaxf(ceilf(hy * kReachMul) + (float)radd, (float)rmin), (float)rmax) : rmin;
  shift[k] = o;
}

// grid: tiles_x * tiles_y * S*N*B blocks (1-D), 256 threads.
// SOLO = the stand-alone operator (sfm_splat_warp_fwd: one image set, S = N = 1): no gather, `trans` may be null, and Acc
// IS the zero-initialised side accumulator - the tile is flushed into it with float atomics (one add per element, no
// contention) because overflow corners of other workgroups land in the same buffer.
#ifdef SFM_SPLAT_WPE
#define SFM_SPLAT_WPE_ATTR __attribute__((amdgpu_waves_per_eu(SFM_SPLAT_WPE, SFM_SPLAT_WPE)))
#else
#define SFM_SPLAT_WPE_ATTR
#endif
template <bool PYR, bool SOLO = false>
__global__ __launch_bounds__(kSplatThreads) SFM_SPLAT_WPE_ATTR void loss_splat_tile_kernel(const float* __restrict__ tgt,
                                                                 const float* __restrict__ src,
                                                                 const float* __restrict__ depth,
                                                                 const float* __restrict__ trans, PyrMaps pm,
                                                                 const float* __restrict__ params, LossDims d,
                                                                 float step_x, float step_y, int tiles_x, int tiles_y,
                                                                 const int4* __restrict__ shift, float* __restrict__ G,
                                                                 float* __restrict__ Acc, float* __restrict__ Ovf,
                                                                 int* __restrict__ flags, float* __restrict__ msumt,
                                                                 unsigned short* __restrict__ ovl) {
  constexpr int TP = STW * STH;
  // planes: sum g*r, sum g*g, sum g*b, sum g.  fp64 on purpose: on gfx950 ds_add_f64 retires a wave
  // instruction in ~9 cycles while ds_add_f32 takes ~190 (profiles/r01_lds_atomic_microbench.txt);
  // the float -> double conversion is exact and the sums lose nothing before the final rounding.
  __shared__ double acc[4 * TP];
  __shared__ int4 tcache[kCacheW * kCacheH];
  __shared__ float msred[kSplatThreads / 64];
  // 1-D grid in XCD-contiguous order: neighbouring tiles (overlapping scan windows) share an L2
  // ... and the N sources of one tile are consecutive: they read the same depth / tgt lines and the same
  // lines of the channel-interleaved trans and src ([.., 3N]), which then leave HBM once instead of N times
  const int blk = xcd_contiguous(blockIdx.x, gridDim.x);
  const int i = blk % d.N, rest = blk / d.N;
  const int tx = rest % tiles_x, ty = (rest / tiles_x) % tiles_y;
  const int sbi = rest / (tiles_x * tiles_y), b = sbi % d.B, s = sbi / d.B;
  const int img = (s * d.N + i) * d.B + b;
  const int HW = d.H * d.W, C3 = 3 * d.N;
  const int ng = d.N < 2 ? d.N : 2;
  const int4* __restrict__ stab = shift + (size_t)img * (tiles_x * tiles_y);
  const int4 mine = stab[ty * tiles_x + tx];
  const int x0 = tx * STW, y0 = ty * STH;
  const int x1 = min(x0 + STW, d.W), y1 = min(y0 + STH, d.H);
  // scan window clipped to the image
  const int sxl = max(x0 - mine.x - mine.z, 0), sxh = min(x0 - mine.x + STW + mine.z, d.W);
  const int syl = max(y0 - mine.y - mine.w, 0), syh = min(y0 - mine.y + STH + mine.w, d.H);
  const int RW = max(sxh - sxl, 0), RH = max(syh - syl, 0);
  const int n_scan = RW * RH;
  const bool home_covered = (RW > 0 && RH > 0 && sxl <= x0 && sxh >= x1 && syl <= y0 && syh >= y1);
  const int n_all = n_scan + (home_covered ? 0 : TP);
  const float inv_rw = RW > 0 ? 1.0f / (float)RW : 0.0f;
  for (int k = threadIdx.x; k < 4 * TP; k += kSplatThreads) acc[k] = 0.0;
  // scan windows of the tiles this tile's pixels can land in (the neighbourhood of where its centre
  // lands), cached in LDS for the overflow test; anything further away is looked up in memory
  const int ctx0 = min(max(x0 + STW / 2 + mine.x, 0), d.W - 1) / STW - kCacheTX;
  const int cty0 = min(max(y0 + STH / 2 + mine.y, 0), d.H - 1) / STH - kCacheTY;
  if (threadIdx.x < kCacheW * kCacheH) {
    const int cx_ = ctx0 + (int)threadIdx.x % kCacheW, cy_ = cty0 + (int)threadIdx.x / kCacheW;
    if (cx_ >= 0 && cx_ < tiles_x && cy_ >= 0 && cy_ < tiles_y)
      tcache[threadIdx.x] = scan_rect(cx_, cy_, stab[cy_ * tiles_x + cx_]);
  }
  // (the barrier that publishes acc / tcache sits after the first pixel's loads have been issued: one memory round trip
  // of the workgroup's start-up less)

  // The 21 pose constants are wave-uniform, so the compiler keeps them in SGPRs - and with the kernel's ~100 other
  // uniform values spills them into VGPR lanes, re-read with v_readlane inside the scan loop (60 of them, plus hazard
  // s_nops).  The register budget that matters here is LDS-bound occupancy (4 waves per SIMD = 128 VGPRs, 62 used): pin
  // the constants into VGPRs instead.
  Pose ps = load_pose(params + ((size_t)i * d.B + b) * SFM_PARAM_STRIDE);
#ifndef SFM_SPLAT_POSE_SGPR
#pragma unroll
  for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(ps.ki[k]));
#pragma unroll
  for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(ps.p[k]));
#endif
  const size_t imgpix = (size_t)b * HW;              // pixel offset of batch item b
  const size_t sb = ((size_t)s * d.B + b) * HW;      // pixel offset of (s, b)
  const float xmax = (float)(d.W - 1), ymax = (float)(d.H - 1);
  float* __restrict__ ovf = Ovf + (size_t)img * HW * 4;

  // The scan loop is software pipelined by hand: the coalesced first stage (depth, translations, tgt) of
  // the NEXT scanned pixel is in flight while this one is processed, and a home pixel's four source taps
  // are requested before the LDS atomics and consumed after them.  (Half of a wave's life was spent parked
  // on these two round trips.)
  // pixel of scan step k: (u, v, scan); ok = false for the home-pass slots that do nothing
  auto locate = [&](int k, int& u, int& v, bool& scan) -> bool {
    if (k < n_scan) {
      const int ly = (int)(((float)k + 0.5f) * inv_rw);
      u = sxl + (k - ly * RW);
      v = syl + ly;
      scan = true;
      return true;
    }
    // home pixels the window does not cover (|shift| > reach)
    const int kk = k - n_scan;
    u = x0 + (kk & (STW - 1));
    v = y0 + (kk / STW);
    scan = false;
    return u < x1 && v < y1 && !(u >= sxl && u < sxh && v >= syl && v < syh);
  };
  int nu = 0, nv = 0;
  bool nscan = false, nok = false;
  float ndep = 0.f;
  f3 nt{0.f, 0.f, 0.f}, ntv{0.f, 0.f, 0.f};
  PyrLevel lv{};
  if constexpr (PYR) lv = pyr_level(pm, s, b, i, C3, d.H, d.W);
  const float* __restrict__ dmap = PYR ? lv.d : depth + sb;                       // a full-size level is read directly
  const bool has_trans = !SOLO || trans != nullptr;
  const float* __restrict__ tmap = PYR ? lv.t : (has_trans ? trans + sb * C3 + 3 * i : nullptr);
  const bool lowres = PYR && lv.lowres;  // workgroup-uniform
  // low-resolution level: the four taps of every map are requested here and interpolated when the pixel is processed
  // (one iteration later), like the plain loads - the resize must not put a memory round trip back into the loop
  float pd[4] = {0.f, 0.f, 0.f, 0.f}, pxl = 0.f, pyl = 0.f;
  f3 pt[4] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  auto fetch = [&](int k) {
    nok = k < n_all && locate(k, nu, nv, nscan);
    if (nok) {
      const int p = nv * d.W + nu;
      ntv = ld3(tgt + (imgpix + p) * 3);
      if (lowres) {
        const Lerp ly = lerp_of(nv, lv.sy, lv.h), lx = lerp_of(nu, lv.sx, lv.w);
        const int o00 = ly.lo * lv.w + lx.lo, o01 = ly.lo * lv.w + lx.hi, o10 = ly.hi * lv.w + lx.lo, o11 = ly.hi * lv.w + lx.hi;
        pd[0] = lv.d[o00]; pd[1] = lv.d[o01]; pd[2] = lv.d[o10]; pd[3] = lv.d[o11];
        pt[0] = ld3(lv.t + (size_t)o00 * C3); pt[1] = ld3(lv.t + (size_t)o01 * C3);
        pt[2] = ld3(lv.t + (size_t)o10 * C3); pt[3] = ld3(lv.t + (size_t)o11 * C3);
        pxl = lx.l; pyl = ly.l;
      } else {
        ndep = dmap[p];
        if (has_trans) nt = ld3(tmap + (size_t)p * C3);
      }
    }
  };
  fetch(threadIdx.x);
  __syncthreads();
  for (int k = threadIdx.x; k < n_all; k += kSplatThreads) {
    const int u = nu, v = nv;
    const bool scan = nscan, ok = nok;
    float dep = ndep;
    f3 t = nt;
    const f3 tv = ntv;
    if (lowres) {
      dep = lerp4(pd[0], pd[1], pd[2], pd[3], pxl, pyl);
      t.x = lerp4(pt[0].x, pt[1].x, pt[2].x, pt[3].x, pxl, pyl);
      t.y = lerp4(pt[0].y, pt[1].y, pt[2].y, pt[3].y, pxl, pyl);
      t.z = lerp4(pt[0].z, pt[1].z, pt[2].z, pt[3].z, pxl, pyl);
    }
    fetch(k + kSplatThreads);
    if (!ok) continue;
    const bool home = (u >= x0 && u < x1 && v >= y0 && v < y1);
    const int p = v * d.W + u;
    const Geom g = geom_fwd(ps, grid_coord(u, step_x, xmax), grid_coord(v, step_y, ymax), dep, has_trans, t.x, t.y, t.z);
    const Corners c = corners(g.x, g.y, xmax, ymax);
    const bool do_gather = !SOLO && home && i < ng;
    f3 ta{0.f, 0.f, 0.f}, tb = ta, tc = ta, td = ta;
    if (do_gather) {  // the taps of the gather (utils.py:322-325): requested now, used after the atomics
      const float* sbase = src + imgpix * C3 + 3 * i;
      ta = ld3(sbase + (size_t)(c.y0 * d.W + c.x0) * C3);
      tb = ld3(sbase + (size_t)(c.y1 * d.W + c.x0) * C3);
      tc = ld3(sbase + (size_t)(c.y0 * d.W + c.x1) * C3);
      td = ld3(sbase + (size_t)(c.y1 * d.W + c.x1) * C3);
    }
    const SplatW sw = splat_weights(g.x, g.y, c, true, g.z);
    const int cx[4] = {c.x0, c.x0, c.x1, c.x1}, cy[4] = {c.y0, c.y1, c.y0, c.y1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w = sw.g[q];
#ifdef SFM_ABL_HALF_ATOMICS  // timing ablation only (wrong numbers): the LDS atomics of the x1 corners as if a neighbouring lane had taken them
      if (q >= 2) continue;
#endif
      if (scan) {
        const int lx = cx[q] - x0, ly = cy[q] - y0;
        if ((unsigned)lx < (unsigned)STW && (unsigned)ly < (unsigned)STH) {
          const int l = ly * STW + lx;
          atomicAdd(&acc[l], (double)(tv.x * w));
          atomicAdd(&acc[TP + l], (double)(tv.y * w));
          atomicAdd(&acc[2 * TP + l], (double)(tv.z * w));
          atomicAdd(&acc[3 * TP + l], (double)w);
        }
      }
    }
    if (home) {
      // which of the four corners does the owner of its tile NOT scan?  (corners usually share a tile)
      auto covered = [&](unsigned dtx, unsigned dty) -> bool {
        const unsigned ox = dtx - (unsigned)ctx0, oy = dty - (unsigned)cty0;
        int4 r;
        if (ox < (unsigned)kCacheW && oy < (unsigned)kCacheH)
          r = tcache[oy * kCacheW + ox];
        else
          r = scan_rect((int)dtx, (int)dty, stab[dty * tiles_x + dtx]);
        return u >= r.x && u < r.y && v >= r.z && v < r.w;
      };
      const unsigned tx0 = (unsigned)c.x0 / STW, tx1 = (unsigned)c.x1 / STW;
      const unsigned ty0 = (unsigned)c.y0 / STH, ty1 = (unsigned)c.y1 / STH;
      const bool k00 = covered(tx0, ty0);
      const bool k01 = (ty1 == ty0) ? k00 : covered(tx0, ty1);
      const bool k10 = (tx1 == tx0) ? k00 : covered(tx1, ty0);
      const bool k11 = (tx1 == tx0) ? k01 : ((ty1 == ty0) ? k10 : covered(tx1, ty1));
      const unsigned miss = (k00 ? 0u : 1u) | (k01 ? 0u : 2u) | (k10 ? 0u : 4u) | (k11 ? 0u : 8u);
      if (miss) {  // rare: overflow path
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((miss >> q) & 1u) {
            float* o = ovf + ((size_t)cy[q] * d.W + cx[q]) * 4;
            atomicAdd(o + 0, tv.x * sw.g[q]);
            atomicAdd(o + 1, tv.y * sw.g[q]);
            atomicAdd(o + 2, tv.z * sw.g[q]);
            if constexpr (SOLO) {
              atomicAdd(o + 3, sw.g[q]);
            } else {
              // The DESTINATION tile's counter: only its consumers (stencil strips over it, the backward tile that clears
              // it, the mask-sum fix-up) pay for the side accumulator.  The weights are positive and Ovf starts at zero,
              // so exactly one adder per pixel finds Ovf.w == 0 in front of a non-zero weight of its own: that one lists
              // the pixel (an add of an underflowed weight 0 changes nothing and lists nothing).
              const float oldw = atomicAdd(o + 3, sw.g[q]);
              if (oldw == 0.0f && sw.g[q] != 0.0f) {
                const size_t dt = (size_t)img * (tiles_x * tiles_y) + ((unsigned)cy[q] / STH) * tiles_x + (unsigned)cx[q] / STW;
                const int slot = atomicAdd(&flags[dt], 1);  // < TP: one entry per pixel of the tile at most
                ovl[dt * TP + slot] = (unsigned short)(((unsigned)cy[q] % STH) * STW + (unsigned)cx[q] % STW);
              }
            }
          }
      }
    }
    if (do_gather) {  // gather (utils.py:297-335)
      const float wx0 = c.fx1 - g.x, wx1 = g.x - c.fx0, wy0 = c.fy1 - g.y, wy1 = g.y - c.fy0;
      const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
      st3(G + ((((size_t)s * ng + i) * d.B + b) * HW + p) * 3,
          ((w00 * ta.x + w01 * tb.x) + w10 * tc.x) + w11 * td.x,
          ((w00 * ta.y + w01 * tb.y) + w10 * tc.y) + w11 * td.y,
          ((w00 * ta.z + w01 * tb.z) + w10 * tc.z) + w11 * td.z);
    }
  }
  __syncthreads();
  float* __restrict__ accg = Acc + (size_t)img * HW * 4;
  float ms = 0.0f;  // this thread's share of the tile's mask sum (loss_kv0): its pixels share one column
  const float wxq = SOLO ? 0.0f : mask_axis_weight(x0 + ((int)threadIdx.x & (STW - 1)), d.W);
  static_assert(kSplatThreads % STW == 0, "a thread's pixels share their column");
  for (int k = threadIdx.x; k < TP; k += kSplatThreads) {
    const int u = x0 + (k & (STW - 1)), v = y0 + (k / STW);
    if (u < x1 && v < y1) {
      float* o = accg + ((size_t)v * d.W + u) * 4;
      if constexpr (SOLO) {
        if (acc[3 * TP + k] != 0.0) {  // (untouched cells stay zero)
          atomicAdd(o + 0, (float)acc[k]);
          atomicAdd(o + 1, (float)acc[TP + k]);
          atomicAdd(o + 2, (float)acc[2 * TP + k]);
          atomicAdd(o + 3, (float)acc[3 * TP + k]);
        }
      } else {
        const float4 a4 = make_float4((float)acc[k], (float)acc[TP + k], (float)acc[2 * TP + k], (float)acc[3 * TP + k]);
        *reinterpret_cast<float4*>(o) = a4;
        ms += nz_count(a4.x, a4.y, a4.z, a4.w) * (wxq * mask_axis_weight(v, d.H));
      }
    }
  }
  if constexpr (!SOLO) {
    ms = wave_sum_dpp_lane63(ms);
    if ((threadIdx.x & 63) == 63) msred[threadIdx.x >> 6] = ms;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
#pragma unroll
      for (int k = 0; k < kSplatThreads / 64; ++k) t += msred[k];
      msumt[(size_t)img * (tiles_x * tiles_y) + ty * tiles_x + tx] = t;
    }
  }
}

// the mask-sum fix-up as a launch of its own (data-parallel forward; the single-device forward runs it inside the
// gather-side stencil launch): one wave per (image, tile), msum_fix_tile in loss_common.h
__global__ __launch_bounds__(kBlock) void loss_msum_fix_kernel(const float* __restrict__ Acc, const float* __restrict__ Ovf,
                                                              const int* __restrict__ flags,
                                                              const unsigned short* __restrict__ ovl,
                                                              float* __restrict__ msumt, int H, int W, int tiles_x,
                                                              int ntile, long n_tiles_all) {
  const long w = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w >= n_tiles_all) return;
  msum_fix_tile(w, threadIdx.x & 63, Acc, Ovf, flags, ovl, msumt, H, W, tiles_x, ntile);
}

// msumt -> msum[S*N] (fp64; the addends are fp32 values of one magnitude, so the fp64 sum is exact whatever the order).
// grid: S*N blocks; the B * ntile tiles of (scale, source) are contiguous.
__global__ __launch_bounds__(kBlock) void loss_msum_reduce_kernel(const float* __restrict__ msumt, long per,
                                                                 double* __restrict__ msum) {
  __shared__ double red[kBlock];
  const float* __restrict__ m = msumt + (size_t)blockIdx.x * per;
  double acc = 0.0;
  for (long k = threadIdx.x; k < per; k += kBlock) acc += (double)m[k];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) msum[blockIdx.x] = red[0];
}

// Zeroes the side accumulator under the raised flags (forward-only steps).  grid (tiles per image, S*N*B).
__global__ __launch_bounds__(kBlock) void loss_ovf_clear_kernel(float* __restrict__ Ovf, const int* __restrict__ flags,
                                                                int H, int W, int tiles_x) {
  const int ntile = gridDim.x;
  if (!flags[(size_t)blockIdx.y * ntile + blockIdx.x]) return;
  const int x0 = (blockIdx.x % tiles_x) * STW, y0 = (blockIdx.x / tiles_x) * STH;
  float4* o = reinterpret_cast<float4*>(Ovf + (size_t)blockIdx.y * H * W * 4);
  for (int k = threadIdx.x; k < STW * STH; k += kBlock) {
    const int u = x0 + (k & (STW - 1)), v = y0 + k / STW;
    if (u < W && v < H) o[(size_t)v * W + u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ------------------------------------------------------------------ finalize
// partials (global sums) -> loss.  (The backward coefficient of the SSIM part, loss_kv0 in loss_common.h, is folded
// into the seeds by the splat-side stencil.)

__device__ __forceinline__ float loss_value(const double* __restrict__ partials, int S, int N, double n_elem,
                                            const double* __restrict__ e_sums = nullptr) {
  const int nterm = 3 * N + 1;
  const double a = (double)kAlpha;
  double loss = 0.0;
  for (int s = 0; s < S; ++s) {
    for (int i = 0; i < N; ++i) {
      const double A = partials[s * nterm + 3 * i], M = partials[s * nterm + 3 * i + 1],
                   V = partials[s * nterm + 3 * i + 2];
      const float h = (1.0f - (float)(V / n_elem)) / 2.0f;
      const float sc = fminf(fmaxf(h, 0.0f), 1.0f);
      loss += a * (A / n_elem) + (1.0 - a) * (double)sc * (M / n_elem);
    }
    loss += 2.0 * (e_sums ? e_sums[s] : partials[s * nterm + 3 * N]) / n_elem;
  }
  return (float)loss;
}

// forward-only steps: just the loss (with gradients, the backward warp kernel does this itself)
__global__ void loss_finalize_kernel(const double* __restrict__ partials, int S, int N, double n_elem,
                                     float* __restrict__ loss_out, const double* __restrict__ e_sums, int n_extra) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float l = loss_value(partials, S, N, n_elem, e_sums);
    for (int k = 0; k < n_extra; ++k) l += (float)partials[S * (3 * N + 1) + k];
    *loss_out = l;
  }
}

// ------------------------------------------------------------------ fused warp backward
// One workgroup = one BTW x BTH pixel tile of one (scale, batch item): 2-D tiles keep the footprint of
// the corner gathers (Hb, source taps) close to the tile itself, and the XCD-contiguous block order
// (sfm_device.h) puts neighbouring tiles behind the same L2, so their overlapping halos are fetched
// from HBM once.  A thread owns kLossPPT pixels; per pixel it walks the N sources (compile-time N: the
// dL/dP accumulators of every source stay in registers, d_depth is summed in a register and the
// 12-byte d_trans pieces of all sources are stored back to back, so each 128-byte line leaves L2 once).
// grid: S*B*tiles blocks (1-D).  d_depth [S,B,H,W], d_trans [S,B,H,W,3N], dP partials [S][N*B][tiles][12].
// One launch handles the NL sources i_base .. i_base + NL - 1 of the d.N in the tensors (NL <= 2: the register budget of
// three waves per SIMD; with all four sources of seq = 5 in one kernel it was 298 VGPRs = ONE wave per SIMD); launches
// after the first add their share to d_depth (accumulate != 0).
// GATHERED: these are sources 0, 1, the ones the min-reprojection term gathers (:168); NTC: d.N when known at compile time
// (the seq = 3 case keeps its compile-time strides: 143 VGPRs instead of 154), else 0
template <int NL, bool GATHERED, int NTC, int IB, bool PYR>
#ifndef SFM_WPE_BWD
#define SFM_WPE_BWD
#endif
#ifndef SFM_BWD_PAIR
#define SFM_BWD_PAIR 2  // sources whose gathers are in flight together (with one seed plane and one pixel per thread both fit: 84 VGPRs)
#endif
__global__ __launch_bounds__(kBlock) SFM_WPE_BWD
void loss_warp_bwd_kernel(const float* __restrict__ tgt,
                                                               const float* __restrict__ src,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ trans, PyrMaps pm,
                                                               const float* __restrict__ params,
                                                               const float* __restrict__ Hb,
                                                               const float* __restrict__ dG,
                                                               const double* __restrict__ partials, double n_elem,
                                                               float gl, int n_extra,
                                                               float* __restrict__ loss_out, LossDims d,
                                                               float step_x, float step_y, int tiles_x, int tiles_y,
                                                               float* __restrict__ d_depth,
                                                               float* __restrict__ d_trans,
                                                               float* __restrict__ dPp, float* __restrict__ Ovf,
                                                               const int* __restrict__ flags) {
  constexpr int i_base = IB;
  constexpr bool accumulate = IB != 0;
  __shared__ float red[4 * 12];
  constexpr int N = NL;  // sources of this launch
  // PYR, low-resolution level: the adjoint of the resize (ResizeBilinearGrad: each pixel's gradient goes to the four
  // cells it was interpolated from, with the forward's own weights) is accumulated per tile in LDS - fp64: ds_add_f64
  // retires in ~9 cycles where ds_add_f32 takes ~190 - and the tile's footprint is then added to
  // the level's gradient maps with one global float atomic per cell and value.
  constexpr int kFootW = BTW + 2, kFootH = BTH + 2, kFootV = 1 + 3 * NL;
  __shared__ double foot[PYR ? kFootW * kFootH * kFootV : 1];
  const int NT = NTC ? NTC : d.N, C3 = 3 * NT;
  constexpr int NG = 2;  // d.N >= 2 (checked on the host)
  const int ntile = tiles_x * tiles_y;
  const int blk = xcd_contiguous(blockIdx.x, gridDim.x);  // (s*B + b)*ntile + tile
  const int tile = blk % ntile, sbi = blk / ntile, b = sbi % d.B, s = sbi / d.B;
  const int x0 = (tile % tiles_x) * BTW, y0 = (tile / tiles_x) * BTH;
  const int HW = d.H * d.W;
  const size_t img = (size_t)b * HW;
  const size_t sb = ((size_t)s * d.B + b) * HW;
  const float xmax = (float)(d.W - 1), ymax = (float)(d.H - 1);
  // gl = dL/d loss.  The loss itself is one thread's job
  if (blockIdx.x == 0 && threadIdx.x == 0 && i_base == 0) {
    float l = loss_value(partials, d.S, NT, n_elem);
    for (int k = 0; k < n_extra; ++k) l += (float)partials[d.S * (3 * NT + 1) + k];  // other terms of total_loss (fp32 adds)
    *loss_out = l;
  }
  Pose ps[N];
  float dp[N][12];
  bool clr[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    ps[i] = load_pose(params + ((size_t)(i_base + i) * d.B + b) * SFM_PARAM_STRIDE);
    // BTW == STW and BTH divides STH: this tile lies inside splat tile (y0 / STH, x0 / STW)
    clr[i] = flags[((size_t)(s * NT + i_base + i) * d.B + b) * (tiles_x * ((d.H + STH - 1) / STH)) + (y0 / STH) * tiles_x + x0 / STW] != 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) dp[i][k] = 0.0f;
  }
  // Memory-level parallelism is what bounds this kernel (every pixel needs a coalesced first stage -
  // depth, tgt, translations - and then ~13 dependent gathers per source), so the loop is software
  // pipelined by hand: the first stage of pixel it+1 is in flight while pixel it is processed, and the
  // gathers of a PAIR of sources are all issued before the first of them is consumed.
  float n_dep;
  f3 n_tv, n_t[N];
  PyrLevel lv{};
  if constexpr (PYR) lv = pyr_level(pm, s, b, i_base, C3, d.H, d.W);
  const float* __restrict__ dmap = PYR ? lv.d : depth + sb;                       // a full-size level is read directly
  const float* __restrict__ tmap = PYR ? lv.t : trans + sb * C3 + 3 * i_base;
  const bool lowres = PYR && lv.lowres;  // workgroup-uniform
  int fx0 = 0, fy0 = 0, fw = 1, fh = 1;  // the tile's footprint on the level: cells [fy0, fy0 + fh) x [fx0, fx0 + fw)
  int fgroup = 1;  // W = w * fgroup with fgroup a power of two (<= 64): aligned groups of fgroup lanes share their cells
  if constexpr (PYR) {
    if (lowres) {
      const int q = d.W / lv.w;
      if (q * lv.w == d.W && q >= 2 && (q & (q - 1)) == 0) fgroup = min(q, 64);
      fx0 = lerp_of(x0, lv.sx, lv.w).lo;
      fy0 = lerp_of(y0, lv.sy, lv.h).lo;
      fw = lerp_of(min(x0 + BTW, d.W) - 1, lv.sx, lv.w).hi - fx0 + 1;
      fh = lerp_of(min(y0 + BTH, d.H) - 1, lv.sy, lv.h).hi - fy0 + 1;
      for (int k = threadIdx.x; k < fw * fh * kFootV; k += kBlock) foot[k] = 0.0;
      __syncthreads();
    }
  }
  auto fetch1 = [&](int it) {
    const int k = it * kBlock + threadIdx.x;
    const int u = min(x0 + (k & (BTW - 1)), d.W - 1), v = min(y0 + k / BTW, d.H - 1);  // clamped: always a valid address
    const int p = v * d.W + u;
    n_tv = ld3(tgt + (img + p) * 3);
    if (lowres) {
      const Lerp ly = lerp_of(v, lv.sy, lv.h), lx = lerp_of(u, lv.sx, lv.w);
      const int o00 = ly.lo * lv.w + lx.lo, o01 = ly.lo * lv.w + lx.hi, o10 = ly.hi * lv.w + lx.lo, o11 = ly.hi * lv.w + lx.hi;
      n_dep = lerp4(lv.d[o00], lv.d[o01], lv.d[o10], lv.d[o11], lx.l, ly.l);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float* tp = lv.t + 3 * i;
        const f3 a = ld3(tp + (size_t)o00 * C3), bq = ld3(tp + (size_t)o01 * C3), c = ld3(tp + (size_t)o10 * C3),
                 e = ld3(tp + (size_t)o11 * C3);
        n_t[i].x = lerp4(a.x, bq.x, c.x, e.x, lx.l, ly.l);
        n_t[i].y = lerp4(a.y, bq.y, c.y, e.y, lx.l, ly.l);
        n_t[i].z = lerp4(a.z, bq.z, c.z, e.z, lx.l, ly.l);
      }
    } else {
      n_dep = dmap[p];
#pragma unroll
      for (int i = 0; i < N; ++i) n_t[i] = ld3(tmap + (size_t)p * C3 + 3 * i);
    }
  };
  // (a low-resolution level interpolates 4 + 4N taps per pixel: fetched at the top of the pixel's own iteration, not one
  // ahead - with the taps of the next pixel in flight beside the gathers of this one the kernel needs 180 VGPRs and
  // loses its third wave per SIMD; the level's maps are small and stay in L1 / L2)
  if (!lowres) fetch1(0);
#pragma unroll 1
  for (int it = 0; it < kLossPPT; ++it) {
    const int k = it * kBlock + threadIdx.x;
    const int u = x0 + (k & (BTW - 1)), v = y0 + k / BTW;
    if (lowres) fetch1(it);
    const float dep = n_dep;
    const f3 tv = n_tv;
    f3 tr[N];
#pragma unroll
    for (int i = 0; i < N; ++i) tr[i] = n_t[i];
    if (!lowres && it + 1 < kLossPPT) fetch1(it + 1);
    const bool valid = u < d.W && v < d.H;
    if (!PYR && !valid) continue;
    const int p = v * d.W + u;
    const float gxc = grid_coord(u, step_x, xmax), gyc = grid_coord(v, step_y, ymax);
    float ddsum = 0.0f;
    float dts[3 * N];
#pragma unroll
    for (int j = 0; j < 3 * N; ++j) dts[j] = 0.0f;
    // (PYR: pixels beyond the image fall through with zero gradients - every lane takes part in the cross-lane sums below)
    if (valid) {
#pragma unroll
    for (int i0 = 0; i0 < N; i0 += SFM_BWD_PAIR) {
      constexpr int kPair = SFM_BWD_PAIR;
      Geom g[kPair];
      Corners c[kPair];
      int idx[kPair][4];
      f3 ta[kPair][4], go[kPair];
      float4 ha[kPair][4];
      // phase A: geometry of the pair, addresses of every gather
#pragma unroll
      for (int j = 0; j < kPair; ++j) {
        const int i = i0 + j;
        if (i >= N) continue;
        g[j] = geom_fwd(ps[i], gxc, gyc, dep, true, tr[i].x, tr[i].y, tr[i].z);
        c[j] = corners(g[j].x, g[j].y, xmax, ymax);
        idx[j][0] = c[j].y0 * d.W + c[j].x0; idx[j][1] = c[j].y1 * d.W + c[j].x0;
        idx[j][2] = c[j].y0 * d.W + c[j].x1; idx[j][3] = c[j].y1 * d.W + c[j].x1;
      }
      // phase B: all gathers of the pair in flight together
#pragma unroll
      for (int j = 0; j < kPair; ++j) {
        const int i = i0 + j;
        if (i >= N) continue;
        const int ig = i_base + i;  // source index in the tensors
        const size_t simg = ((size_t)s * NT + ig) * d.B + b;
        // this kernel runs after the last reader of Ovf: restore the all-zero invariant where it was used
        if (clr[i]) reinterpret_cast<float4*>(Ovf + simg * HW * 4)[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* hb = Hb + simg * HW * 4;
        if (GATHERED) {
          const float* sbase = src + img * C3 + 3 * ig;
#pragma unroll
          for (int q = 0; q < 4; ++q) ta[j][q] = ld3(sbase + (size_t)idx[j][q] * C3);
          go[j] = ld3(dG + ((((size_t)s * NG + ig) * d.B + b) * HW + p) * 3);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ha[j][q] = *reinterpret_cast<const float4*>(hb + (size_t)idx[j][q] * 4);
      }
      // phase C: chain rule
#pragma unroll
      for (int j = 0; j < kPair; ++j) {
        const int i = i0 + j;
        if (i >= N) continue;
        float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f;
        if (GATHERED) {
          const float wx0 = c[j].fx1 - g[j].x, wx1 = g[j].x - c[j].fx0, wy0 = c[j].fy1 - g[j].y, wy1 = g[j].y - c[j].fy0;
          const f3 a = ta[j][0], bb = ta[j][1], cc = ta[j][2], dd = ta[j][3], gq = go[j];
          gx_ = gl * (gq.x * (wy0 * (cc.x - a.x) + wy1 * (dd.x - bb.x)) + gq.y * (wy0 * (cc.y - a.y) + wy1 * (dd.y - bb.y)) +
                      gq.z * (wy0 * (cc.z - a.z) + wy1 * (dd.z - bb.z)));
          gy_ = gl * (gq.x * (wx0 * (bb.x - a.x) + wx1 * (dd.x - cc.x)) + gq.y * (wx0 * (bb.y - a.y) + wx1 * (dd.y - cc.y)) +
                      gq.z * (wx0 * (bb.z - a.z) + wx1 * (dd.z - cc.z)));
        }
        const SplatW sw = splat_weights(g[j].x, g[j].y, c[j], true, g[j].z);
        float dgk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float da = ((ha[j][q].x * tv.x + ha[j][q].y * tv.y) + ha[j][q].z * tv.z) - ha[j][q].w;
          dgk[q] = (gl * da) * sw.g[q];
        }
        gx_ += -(dgk[0] + dgk[1]) * sw.dx0 + (dgk[2] + dgk[3]) * sw.dx1;
        gy_ += -(dgk[0] + dgk[2]) * sw.dy0 + (dgk[1] + dgk[3]) * sw.dy1;
        gz_ = -0.25f * sw.zin * ((dgk[0] + dgk[1]) + (dgk[2] + dgk[3]));
        float dd_;
        geom_bwd(ps[i], g[j], dep, true, tr[i].x, tr[i].y, tr[i].z, gx_, gy_, gz_, dd_, dts[3 * i], dts[3 * i + 1],
                 dts[3 * i + 2], dp[i]);
        ddsum += dd_;
      }
    }
    }  // valid
    if constexpr (PYR) {
      if (lowres) {
        const Lerp ly = lerp_of(min(v, d.H - 1), lv.sy, lv.h), lx = lerp_of(min(u, d.W - 1), lv.sx, lv.w);
        const float xl = lx.l, yl = ly.l;  // plain locals: a select between struct members is a select between addresses
        const int cxl = lx.lo - fx0, cxh = lx.hi - fx0, cyl = (ly.lo - fy0) * fw, cyh = (ly.hi - fy0) * fw;
        if (fgroup > 1) {
          // The lanes of a wave are 64 consecutive columns of one row.  With W = w * 2^k the 2^k lanes of an aligned
          // group share their left cell (and their right one): summed across the group first (xor butterfly), one lane
          // adds for all - the same cell hit by 2^k lanes of one ds_add is serialised 2^k-fold otherwise.
          float a[kFootV], bq[kFootV];
          a[0] = (1.0f - xl) * ddsum; bq[0] = xl * ddsum;
#pragma unroll
          for (int j = 0; j < 3 * N; ++j) { a[1 + j] = (1.0f - xl) * dts[j]; bq[1 + j] = xl * dts[j]; }
          for (int m = 1; m < fgroup; m <<= 1) {
#pragma unroll
            for (int j = 0; j < kFootV; ++j) { a[j] += __shfl_xor(a[j], m, 64); bq[j] += __shfl_xor(bq[j], m, 64); }
          }
          if (((threadIdx.x & 63) & (fgroup - 1)) == 0) {
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {  // the two cell rows of this pixel row
              const float wy = q ? yl : 1.0f - yl;
              double* f0 = foot + (size_t)((q ? cyh : cyl) + cxl) * kFootV;
              double* f1 = foot + (size_t)((q ? cyh : cyl) + cxh) * kFootV;
#pragma unroll
              for (int j = 0; j < kFootV; ++j) { atomicAdd(f0 + j, (double)(wy * a[j])); atomicAdd(f1 + j, (double)(wy * bq[j])); }
            }
          }
        } else {
          // any other ratio: one corner at a time (not unrolled): 7 products live, not 28
#pragma unroll 1
          for (int q = 0; q < 4; ++q) {
            const bool qx = (q & 1) != 0, qy = (q & 2) != 0;
            const float wgt = (qx ? xl : 1.0f - xl) * (qy ? yl : 1.0f - yl);
            double* f = foot + (size_t)((qy ? cyh : cyl) + (qx ? cxh : cxl)) * kFootV;
            atomicAdd(f, (double)(wgt * ddsum));
#pragma unroll
            for (int j = 0; j < 3 * N; ++j) atomicAdd(f + 1 + j, (double)(wgt * dts[j]));
          }
        }
      } else if (valid) {  // a level at full size: one writer per element, added to what the caller put there
        float* dd = pm.d_depth[s] + (size_t)b * HW + p;
        *dd = *dd + ddsum;
        float* dt = pm.d_trans[s] + ((size_t)b * HW + p) * C3 + 3 * i_base;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const f3 o = ld3(dt + 3 * i);
          st3(dt + 3 * i, o.x + dts[3 * i], o.y + dts[3 * i + 1], o.z + dts[3 * i + 2]);
        }
      }
    } else {
      d_depth[sb + p] = accumulate ? d_depth[sb + p] + ddsum : ddsum;
#pragma unroll
      for (int i = 0; i < N; ++i)
        st3(d_trans + (sb + p) * C3 + 3 * (i_base + i), dts[3 * i], dts[3 * i + 1], dts[3 * i + 2]);
    }
  }
  if constexpr (PYR) {
    if (lowres) {  // the tile's footprint -> the level's gradient maps (cells shared with neighbouring tiles: atomics)
      __syncthreads();
      float* gd = pm.d_depth[s] + (size_t)b * lv.h * lv.w;
      float* gt = pm.d_trans[s] + (size_t)b * lv.h * lv.w * C3 + 3 * i_base;
      for (int k = threadIdx.x; k < fw * fh * kFootV; k += kBlock) {
        const int cell = k / kFootV, j = k - cell * kFootV;
        const int cy = cell / fw, cx = cell - cy * fw;
        const size_t o = (size_t)(fy0 + cy) * lv.w + (fx0 + cx);
        const float val = (float)foot[k];
        if (j == 0) atomicAdd(gd + o, val);
        else atomicAdd(gt + o * C3 + (j - 1), val);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i) __syncthreads();  // red is reused per source
    block_sum_store<12>(dp[i], red, dPp + ((((size_t)s * NT + i_base + i) * d.B + b) * ntile + tile) * 12);
  }
}


// ------------------------------------------------------------------ per-stage timing
// Optional measurement aid for bench.py: when enabled, the two loss entry points bracket
// each launch with HIP events on the caller's stream.  Off by default (no events, no
// overhead); process-wide, not thread-safe - a bench facility, not part of the data path.
enum { ST_SHIFT = 0, ST_WARP_FWD, ST_STENCIL, ST_REDUCE, ST_FINALIZE, ST_WARP_BWD, ST_POSE_GRAD, ST_MSUM, ST_COUNT };
static bool g_prof_on = false;
static hipEvent_t g_ev[ST_COUNT][2];
static bool g_ev_used[ST_COUNT];
static bool g_ev_made = false;

struct StageTimer {
  int id;
  hipStream_t st;
  StageTimer(int id_, hipStream_t st_) : id(id_), st(st_) {
    if (g_prof_on) { (void)hipEventRecord(g_ev[id][0], st); }
  }
  ~StageTimer() {
    if (g_prof_on) { (void)hipEventRecord(g_ev[id][1], st); g_ev_used[id] = true; }
  }
};

static inline bool dims_ok(const LossDims& d) {
  if (!(d.B > 0 && d.H > 0 && d.W > 0 && d.S > 0 && d.N >= 2 && d.N <= kMaxSrc && d.B <= 65535 && d.S <= 65535 &&
        (long)d.S * d.N * d.B <= 65535 && sfm_fits(d.H, d.W, 3 * d.N > 8 ? 3 * d.N : 8)))
    return false;
  const StencilPlan sp = stencil_plan(d, 6, 6);  // the smallest rows per wave = the most work items (32-bit in the kernels)
  return sp.n_splat + sp.n_gather < 2147483647L;
}

// Tuning knobs (experiments only): image rows per stencil wave, splat side / gather side; 0 = the model of stencil_rows.
static int env_rows(const char* name) {
  const char* e = getenv(name);
  const int r = e ? atoi(e) : 0;
  return (r >= 1 && r <= 4096) ? r : 0;
}
static int stencil_rows_override() {
  static const int v = env_rows("SFM_STENCIL_ROWS");
  return v;
}
static int stencil_rows_g_override() {
  static const int v = env_rows("SFM_STENCIL_ROWS_G");
  return v;
}

struct LossLayout {
  StencilPlan sp;
  LossWs w;
};
static inline LossLayout loss_layout(const LossDims& d) {
  LossLayout l;
  l.sp = stencil_plan(d, stencil_rows_override(), stencil_rows_g_override());
  l.w = loss_ws(d, l.sp);
  return l;
}

}  // namespace sfm

using namespace sfm;


// The forward splat of ONE image set on the tile-owned kernels (warp.hip: sfm_splat_warp_fwd).  acc [B,H,W,4] must be
// zero on the stream; table: scratch for B * tiles int4 (16-byte aligned).  False = shape not supported here.
namespace sfm {
long splat_solo_table_int4s(int B, int H, int W) { return (long)B * ((W + STW - 1) / STW) * ((H + STH - 1) / STH); }
bool launch_splat_solo(const float* img, const float* depth, const float* trans, const float* params, int B, int H, int W,
                       float* acc, void* table, hipStream_t stream) {
  if (B > 65535 || !sfm_fits(H, W, 8)) return false;
  LossDims d{B, H, W, 1, 1};
  const int tx = (W + STW - 1) / STW, ty = (H + STH - 1) / STH;
  const long n = (long)B * tx * ty;
  int4* shift = reinterpret_cast<int4*>(table);
  hipLaunchKernelGGL(loss_tile_shift_kernel<false>, dim3((unsigned)((n * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     depth, trans, PyrMaps{}, params, d, sfm_lin_step(W), sfm_lin_step(H), tx, ty, shift, kReachAdd, kReachMin,
                     kReachMax);
  hipLaunchKernelGGL((loss_splat_tile_kernel<false, true>), dim3((unsigned)n), dim3(kSplatThreads), 0, stream, img,
                     (const float*)nullptr, depth, trans, PyrMaps{}, params, d, sfm_lin_step(W), sfm_lin_step(H), tx, ty, shift,
                     (float*)nullptr, acc, acc, (int*)nullptr, (float*)nullptr, (unsigned short*)nullptr);
  return true;
}
}  // namespace sfm

extern "C" long long sfm_loss_workspace_floats(int B, int H, int W, int S, int N) {
  LossDims d{B, H, W, S, N};
  if (!dims_ok(d)) return SFM_ERR_ARG;
  return (long long)loss_layout(d).w.total;
}

extern "C" int sfm_loss_num_partials(int B, int H, int W, int S, int N) {
  LossDims d{B, H, W, S, N};
  if (!dims_ok(d)) return SFM_ERR_ARG;
  return S * (3 * N + 1);
}

extern "C" int sfm_loss_workspace_region(int B, int H, int W, int S, int N, int which, long long* offset, long long* count) {
  LossDims d{B, H, W, S, N};
  if (!offset || !count || !dims_ok(d)) return SFM_ERR_ARG;
  const LossWs w = loss_layout(d).w;
  const long long acc = (long long)S * N * B * H * W * 4;
  switch (which) {
    case 0: *offset = (long long)w.Acc; *count = acc; break;
    case 1: *offset = (long long)w.Ovf; *count = acc; break;
    case 2: *offset = (long long)w.Hb; *count = acc; break;
    case 3: *offset = (long long)w.flags; *count = w.n_flags; break;
    case 4: *offset = (long long)w.msum; *count = 2LL * S * N; break;
    default: return SFM_ERR_ARG;
  }
  return SFM_OK;
}

// Establishes the invariant both calls below rely on and restore: the side accumulator Ovf and
// the per-image overflow flags are all zero between steps.  Call once on a fresh workspace (and
// again after a failed call).
extern "C" int sfm_loss_workspace_init(float* ws, int B, int H, int W, int S, int N, void* stream_) {
  LossDims d{B, H, W, S, N};
  if (!ws || !dims_ok(d)) return SFM_ERR_ARG;
  const LossWs w = loss_layout(d).w;
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t e = hipMemsetAsync(ws + w.Ovf, 0, (size_t)S * N * B * H * W * 4 * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(ws + w.flags, 0, (size_t)w.n_flags * sizeof(int), stream);
  return e == hipSuccess ? SFM_OK : (int)e;
}

// phases of the forward (any subset, run in this order):
//   kPhaseWarps   tile_shift + both warps (Acc, G, the tiles' mask-sum shares) + the mask sums M(s, i) -> msum_out
//   kPhaseGather  gather-side stencil (min-reprojection term; dG)
//   kPhaseSplat   splat-side stencil (masked L1 + SSIM terms; Hb with kV0 from the GLOBAL mask sums msum_in) + the
//                 reduction of every per-wave sum -> partials (so kPhaseGather must have run before it)
// pm: the prediction pyramid the warps read (then depth / trans are unused and may be null), or null = depth / trans
// are the maps already resized to HxW, stacked over the scales.  msum_out / msum_in null: the workspace's own slot (the
// single-device step, where this rank's mask sums are the global ones).
constexpr int kPhaseWarps = 1, kPhaseGather = 2, kPhaseSplat = 4, kPhaseAll = 7;
static int loss_forward_phases(int phases, const float* tgt, const float* src, const float* depth, const float* trans,
                               const PyrMaps* pm, const float* params, float* ws, int B, int H, int W, int S, int N,
                               long long n_elem_global, int need_grad, double* msum_out, const double* msum_in,
                               double* partials, void* stream_) {
  LossDims d{B, H, W, S, N};
  if (!ws || !dims_ok(d)) return SFM_ERR_ARG;
  if ((phases & kPhaseWarps) && (!tgt || !src || (!pm && (!depth || !trans)) || !params)) return SFM_ERR_ARG;
  if ((phases & kPhaseGather) && (!tgt || n_elem_global <= 0)) return SFM_ERR_ARG;
  if ((phases & kPhaseSplat) && (!src || !partials || n_elem_global <= 0)) return SFM_ERR_ARG;
  const PyrMaps pmv = pm ? *pm : PyrMaps{};
  hipStream_t stream = (hipStream_t)stream_;
  const LossLayout l = loss_layout(d);
  const LossWs& w = l.w;
  int4* shift = reinterpret_cast<int4*>(ws + w.shift);
  int* flags = reinterpret_cast<int*>(ws + w.flags);
  float* msumt = ws + w.msumt;
  unsigned short* ovl = reinterpret_cast<unsigned short*>(ws + w.ovl);
  double* msum_ws = reinterpret_cast<double*>(ws + w.msum);
  const bool single = phases == kPhaseAll && !msum_out && !msum_in;
  int rc;
  if (phases & kPhaseWarps) {
    { StageTimer t(ST_SHIFT, stream);
    const long n = (long)w.n_img * w.stiles_x * w.stiles_y;
    const dim3 grid((unsigned)((n * 16 + kBlock - 1) / kBlock));
    if (pm)
      hipLaunchKernelGGL(loss_tile_shift_kernel<true>, grid, dim3(kBlock), 0, stream, depth, trans, pmv, params, d,
                         sfm_lin_step(W), sfm_lin_step(H), w.stiles_x, w.stiles_y, shift, kReachAdd, kReachMin, kReachMax);
    else
      hipLaunchKernelGGL(loss_tile_shift_kernel<false>, grid, dim3(kBlock), 0, stream, depth, trans, pmv, params, d,
                         sfm_lin_step(W), sfm_lin_step(H), w.stiles_x, w.stiles_y, shift, kReachAdd, kReachMin, kReachMax); }
    { StageTimer t(ST_WARP_FWD, stream);
    const dim3 grid((unsigned)((long)w.stiles_x * w.stiles_y * w.n_img));
    if (pm)
      hipLaunchKernelGGL(loss_splat_tile_kernel<true>, grid, dim3(kSplatThreads), 0, stream, tgt, src, depth, trans, pmv, params, d,
                         sfm_lin_step(W), sfm_lin_step(H), w.stiles_x, w.stiles_y, shift, ws + w.G, ws + w.Acc, ws + w.Ovf, flags,
                         msumt, ovl);
    else
      hipLaunchKernelGGL(loss_splat_tile_kernel<false>, grid, dim3(kSplatThreads), 0, stream, tgt, src, depth, trans, pmv, params, d,
                         sfm_lin_step(W), sfm_lin_step(H), w.stiles_x, w.stiles_y, shift, ws + w.G, ws + w.Acc, ws + w.Ovf, flags,
                         msumt, ovl); }
    // single device (every phase in this call, the mask sums never leave the workspace): no launch here - the fix-up rides
    // in the gather-side stencil launch and the splat-side waves add up the tile shares themselves
    if (!single) { StageTimer t(ST_MSUM, stream);
    const int ntile = w.stiles_x * w.stiles_y;
    hipLaunchKernelGGL(loss_msum_fix_kernel, dim3((unsigned)((w.n_flags + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                       stream, ws + w.Acc, ws + w.Ovf, flags, ovl, msumt, H, W, w.stiles_x, ntile, w.n_flags);
    hipLaunchKernelGGL(loss_msum_reduce_kernel, dim3(S * N), dim3(kBlock), 0, stream, msumt, (long)B * ntile,
                       msum_out ? msum_out : msum_ws); }
    rc = sfm_last_error();
    if (rc) return rc;
  }
  if (!(phases & (kPhaseGather | kPhaseSplat))) return SFM_OK;
  StencilArgs a{tgt, src, ws + w.G, ws + w.Acc, ws + w.Ovf, flags, w.stiles_x, w.stiles_y, ws + w.Hb, ws + w.dG, ws + w.wavep, d, l.sp,
                (float)((double)kAlpha / (double)n_elem_global), (float)(2.0 / (double)n_elem_global),
                single ? (const double*)nullptr : (msum_in ? msum_in : msum_ws), 1.0 / (double)n_elem_global, msumt, ovl,
                single ? (int)((w.n_flags + kBlock / 64 - 1) / (kBlock / 64)) : 0};
  { StageTimer t(ST_STENCIL, stream);
  rc = launch_loss_stencil(a, need_grad != 0, ((phases & kPhaseGather) ? kStencilGather : 0) | ((phases & kPhaseSplat) ? kStencilSplat : 0),
                           stream);
  if (rc) return rc; }
  if (phases & kPhaseSplat) {
    StageTimer t(ST_REDUCE, stream);
    rc = launch_loss_reduce(a, partials, single && need_grad != 0, stream);
    if (rc) return rc;
  }
  return sfm_last_error();
}

extern "C" int sfm_loss_forward(const float* tgt, const float* src, const float* depth, const float* trans,
                                const float* params, float* ws, int B, int H, int W, int S, int N,
                                long long n_elem_global, int need_grad, double* partials, void* stream_) {
  return loss_forward_phases(kPhaseAll, tgt, src, depth, trans, nullptr, params, ws, B, H, W, S, N, n_elem_global, need_grad,
                             nullptr, nullptr, partials, stream_);
}

// The forward in three calls, for the data-parallel step (same kernels as sfm_loss_forward):
//   _warps   both warps and this rank's mask sums msum[S*N] (fp64)
//   [all-reduce (sum) of msum over the ranks, on the communication stream ...]
//   _gather  [... while] the gather-side stencil runs
//   _splat   the splat-side stencil with the GLOBAL mask sums, then every partial sum -> partials[S*(3N+1)]
//   [all-reduce (sum) of partials]  ->  sfm_loss_backward
extern "C" int sfm_loss_forward_warps(const float* tgt, const float* src, const float* depth, const float* trans,
                                      const float* params, float* ws, int B, int H, int W, int S, int N, double* msum,
                                      void* stream_) {
  if (!msum) return SFM_ERR_ARG;
  return loss_forward_phases(kPhaseWarps, tgt, src, depth, trans, nullptr, params, ws, B, H, W, S, N, 0, 1, msum, nullptr,
                             nullptr, stream_);
}

extern "C" int sfm_loss_forward_gather(const float* tgt, float* ws, int B, int H, int W, int S, int N,
                                       long long n_elem_global, int need_grad, void* stream_) {
  return loss_forward_phases(kPhaseGather, tgt, nullptr, nullptr, nullptr, nullptr, nullptr, ws, B, H, W, S, N, n_elem_global,
                             need_grad, nullptr, nullptr, nullptr, stream_);
}

extern "C" int sfm_loss_forward_splat(const float* src, float* ws, int B, int H, int W, int S, int N,
                                      long long n_elem_global, int need_grad, const double* msum_global, double* partials,
                                      void* stream_) {
  if (!msum_global) return SFM_ERR_ARG;
  return loss_forward_phases(kPhaseSplat, nullptr, src, nullptr, nullptr, nullptr, nullptr, ws, B, H, W, S, N, n_elem_global,
                             need_grad, nullptr, msum_global, partials, stream_);
}

// partials: the (all-reduced) sums of sfm_loss_forward; n_elem_global = B_global*H*W*3 (the value
// given to sfm_loss_forward).  loss_out: one float.  need_grad = 0 stops after the loss.
static int loss_backward_common(const float* tgt, const float* src, const float* depth, const float* trans, const PyrMaps* pm,
                                const float* params, float* ws, int B, int H, int W, int S, int N, long long n_elem_global,
                                float gloss, int need_grad, const double* partials, int n_extra, const float* pose,
                                const float* K, float* loss_out, float* d_depth, float* d_trans, float* d_pose,
                                void* stream_) {
  LossDims d{B, H, W, S, N};
  if (n_extra < 0 || n_extra > 16) return SFM_ERR_ARG;
  if (!tgt || !src || (!pm && (!depth || !trans)) || !params || !ws || !partials || !loss_out || n_elem_global <= 0 ||
      !dims_ok(d))
    return SFM_ERR_ARG;
  if (need_grad && (!pose || !K || !d_depth || !d_trans || !d_pose)) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const LossLayout ll = loss_layout(d);
  const LossWs& w = ll.w;
  const PyrMaps pmv = pm ? *pm : PyrMaps{};
  int* flags = reinterpret_cast<int*>(ws + w.flags);
  if (!need_grad) {  // forward-only step: the loss, and restore the Ovf / flags invariant here
    { StageTimer t(ST_FINALIZE, stream);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, partials, S, N, (double)n_elem_global, loss_out,
                       (const double*)nullptr, n_extra); }
    hipLaunchKernelGGL(loss_ovf_clear_kernel, dim3(w.stiles_x * w.stiles_y, w.n_img), dim3(kBlock), 0, stream, ws + w.Ovf, flags,
                       H, W, w.stiles_x);
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)w.n_flags * sizeof(int), stream);
    return e != hipSuccess ? (int)e : sfm_last_error();
  }
  if (!(need_grad & 2)) {  // (need_grad = 3: sfm_loss_forward's own reduction has applied the guard - single device)
    // seeds written under a clip that turned out not to pass (degenerate inputs only): rewritten before they are read
    StageTimer t(ST_FINALIZE, stream);
    StencilArgs a{tgt, src, ws + w.G, ws + w.Acc, ws + w.Ovf, flags, w.stiles_x, w.stiles_y, ws + w.Hb, ws + w.dG, ws + w.wavep, d, ll.sp,
                  (float)((double)kAlpha / (double)n_elem_global), (float)(2.0 / (double)n_elem_global),
                  reinterpret_cast<const double*>(ws + w.msum), 1.0 / (double)n_elem_global, nullptr, nullptr, 0};
    const int rc0 = launch_loss_stencil_redo(a, partials, stream);
    if (rc0) return rc0;
  }
  { StageTimer t(ST_WARP_BWD, stream);
  const dim3 grid((unsigned)((long)S * B * w.nblk_bwd));
#define SFM_LAUNCH_BWD_P(NSRC, IBASE, NTC, PYR)                                                                          \
  hipLaunchKernelGGL((loss_warp_bwd_kernel<NSRC, (IBASE) == 0, NTC, IBASE, PYR>), grid, dim3(kBlock), 0, stream, tgt, src, depth,  \
                     trans, pmv, params, ws + w.Hb, ws + w.dG, partials, (double)n_elem_global,                                     \
                     gloss, n_extra, loss_out, d, sfm_lin_step(W), sfm_lin_step(H), w.btiles_x, w.btiles_y, d_depth, d_trans,       \
                     ws + w.dPp, ws + w.Ovf, flags)
#define SFM_LAUNCH_BWD(NSRC, IBASE, NTC)                                   \
  do {                                                                     \
    if (pm) SFM_LAUNCH_BWD_P(NSRC, IBASE, NTC, true);                      \
    else SFM_LAUNCH_BWD_P(NSRC, IBASE, NTC, false);                        \
  } while (0)
  // sources 0, 1 (the gathered ones); then the rest, two at a time, adding into d_depth
  if (N == 2) SFM_LAUNCH_BWD(2, 0, 2); else SFM_LAUNCH_BWD(2, 0, 0);
  if (N == 3) SFM_LAUNCH_BWD(1, 2, 0);
  if (N == 4) SFM_LAUNCH_BWD(2, 2, 0);
#undef SFM_LAUNCH_BWD
#undef SFM_LAUNCH_BWD_P
  }
  int rc = sfm_last_error();
  if (rc) return rc;
  // pose [B,N,6]: element (b,i,k) at b*6N + i*6 + k; partial sets = scales
  StageTimer t(ST_POSE_GRAD, stream);
  // the last kernel of the step also lowers the overflow flags (Ovf itself was re-zeroed by warp_bwd)
  return sfm_launch_pose_grad(pose, 6L * N, 6L, K, ws + w.dPp, S, w.nblk_bwd, B, N, 0, 1.0f, d_pose, 6L * N, 6L, flags,
                              (int)w.n_flags, stream);
}

extern "C" int sfm_loss_backward(const float* tgt, const float* src, const float* depth, const float* trans,
                                 const float* params, float* ws, int B, int H, int W, int S, int N,
                                 long long n_elem_global, float gloss, int need_grad, const double* partials,
                                 const float* pose, const float* K, float* loss_out, float* d_depth, float* d_trans,
                                 float* d_pose, void* stream_) {
  return loss_backward_common(tgt, src, depth, trans, nullptr, params, ws, B, H, W, S, N, n_elem_global, gloss, need_grad,
                              partials, 0, pose, K, loss_out, d_depth, d_trans, d_pose, stream_);
}

// ---- the same step fed with the prediction pyramid -----------------------------------------
static int fill_pyr_maps(PyrMaps& pm, const float* const* depth_s, const float* const* trans_s, const int* h, const int* w,
                         int S, int H, int W, float* const* d_depth_s = nullptr, float* const* d_trans_s = nullptr) {
  if (!depth_s || !trans_s || !h || !w || S < 1 || S > kMaxScales) return SFM_ERR_ARG;
  pm = PyrMaps{};
  for (int s = 0; s < S; ++s) {
    if (!depth_s[s] || !trans_s[s] || h[s] < 1 || w[s] < 1 || h[s] > H || w[s] > W) return SFM_ERR_ARG;
    pm.depth[s] = depth_s[s];
    pm.trans[s] = trans_s[s];
    pm.h[s] = h[s];
    pm.w[s] = w[s];
    pm.sy[s] = (float)h[s] / (float)H;  // CalculateResizeScale: in / static_cast<float>(out)
    pm.sx[s] = (float)w[s] / (float)W;
    if (d_depth_s) {
      if (!d_trans_s || !d_depth_s[s] || !d_trans_s[s]) return SFM_ERR_ARG;
      pm.d_depth[s] = d_depth_s[s];
      pm.d_trans[s] = d_trans_s[s];
    }
  }
  return SFM_OK;
}

extern "C" int sfm_loss_forward_pyr(const float* tgt, const float* src, const float* const* depth_s,
                                    const float* const* trans_s, const int* h, const int* w, const float* params, float* ws,
                                    int B, int H, int W, int S, int N, long long n_elem_global, int need_grad,
                                    double* partials, void* stream_) {
  PyrMaps pm;
  const int rc = fill_pyr_maps(pm, depth_s, trans_s, h, w, S, H, W);
  if (rc) return rc;
  return loss_forward_phases(kPhaseAll, tgt, src, nullptr, nullptr, &pm, params, ws, B, H, W, S, N, n_elem_global, need_grad,
                             nullptr, nullptr, partials, stream_);
}

// sfm_loss_forward_warps fed with the pyramid (the other two calls of the data-parallel forward read no prediction map)
extern "C" int sfm_loss_forward_warps_pyr(const float* tgt, const float* src, const float* const* depth_s,
                                          const float* const* trans_s, const int* h, const int* w, const float* params,
                                          float* ws, int B, int H, int W, int S, int N, double* msum, void* stream_) {
  if (!msum) return SFM_ERR_ARG;
  PyrMaps pm;
  const int rc = fill_pyr_maps(pm, depth_s, trans_s, h, w, S, H, W);
  if (rc) return rc;
  return loss_forward_phases(kPhaseWarps, tgt, src, nullptr, nullptr, &pm, params, ws, B, H, W, S, N, 0, 1, msum, nullptr,
                             nullptr, stream_);
}

extern "C" int sfm_loss_backward_pyr(const float* tgt, const float* src, const float* const* depth_s,
                                     const float* const* trans_s, const int* h, const int* w, const float* params, float* ws,
                                     int B, int H, int W, int S, int N, long long n_elem_global, float gloss, int need_grad,
                                     const double* partials, int n_extra, const float* pose, const float* K,
                                     float* loss_out, float* const* d_depth_s, float* const* d_trans_s, float* d_pose,
                                     void* stream_) {
  PyrMaps pm;
  const int rc = fill_pyr_maps(pm, depth_s, trans_s, h, w, S, H, W, need_grad ? d_depth_s : nullptr,
                               need_grad ? d_trans_s : nullptr);
  if (rc) return rc;
  if (need_grad && (!d_depth_s || !d_trans_s)) return SFM_ERR_ARG;
  // the kernels reach the gradient maps through pm; the stacked pointers are unused (non-null for the shared checks)
  float* const any = need_grad ? d_depth_s[0] : nullptr;
  return loss_backward_common(tgt, src, nullptr, nullptr, &pm, params, ws, B, H, W, S, N, n_elem_global, gloss, need_grad,
                              partials, n_extra, pose, K, loss_out, any, any, d_pose, stream_);
}

// ---- measurement aid (bench.py): per-launch HIP-event timing of the two calls above ----
extern "C" int sfm_prof_enable(int on) {
  if (on && !g_ev_made) {
    for (int i = 0; i < ST_COUNT; ++i)
      for (int k = 0; k < 2; ++k) {
        hipError_t e = hipEventCreate(&g_ev[i][k]);
        if (e != hipSuccess) return (int)e;
      }
    g_ev_made = true;
  }
  for (int i = 0; i < ST_COUNT; ++i) g_ev_used[i] = false;
  g_prof_on = on != 0;
  return SFM_OK;
}

extern "C" int sfm_prof_stages(void) { return ST_COUNT; }

// Adds the elapsed milliseconds of the last step's launches into ms[ST_COUNT]; the
// caller must have synchronised the stream.  Stage order: tile_shift, warp_fwd, stencil,
// reduce, finalize (forward-only steps) / seed redo check, warp_bwd, pose_grad, mask sums.
extern "C" int sfm_prof_read(float* ms) {
  if (!g_ev_made || !ms) return SFM_ERR_ARG;
  for (int i = 0; i < ST_COUNT; ++i) {
    if (!g_ev_used[i]) continue;
    float t = 0.f;
    hipError_t e = hipEventElapsedTime(&t, g_ev[i][0], g_ev[i][1]);
    if (e != hipSuccess) return (int)e;
    ms[i] += t;
    g_ev_used[i] = false;
  }
  return SFM_OK;
}
