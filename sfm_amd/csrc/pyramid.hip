// The steps of build_train_graph either side of the warp loop that work on the prediction PYRAMID
// (SURVEY.md section 8f rows 1-2), each for all scales in one launch:
//
//   sfm_disp_normalize_*    pred_disp = d / reduce_mean(d, [1,2,3]); pred_depth = 1 / pred_disp   SfMLearner_uni.py:40,47
//   sfm_pyramid_upsample_*  tf.image.resize_bilinear(pred_depth[s] / pred_exp_logits[s], [H, W])   SfMLearner_uni.py:91,113
//   sfm_smooth_loss         smooth_weight/2^s compute_smooth_loss(pred_disp[s]) + explain_reg_weight/2^s
//                           (compute_smooth_loss(pred_exp_logits[s]) + mean|pred_exp_logits[s]|)   SfMLearner_uni.py:94-101, 205-219
//
// and their adjoints (implicit in the reference: tf.gradients, SfMLearner_uni.py:306).  None of them has an
// exchange step: under data parallelism every mean is over the global batch (B_global in the denominators)
// and the caller sums the scalar over ranks.  All reductions run in a fixed order: results are reproducible.
//
// The per-scale pointers and sizes of a call are copied from the caller's host arrays into the kernel
// arguments (<= kMaxMaps maps), so no device-side table has to be built.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {
namespace {

constexpr int kMaxMaps = 16;
constexpr int kMaxC = 12;  // 3 * (at most 4 sources)

// ------------------------------------------------------------------ disp normalisation
// Two launches per direction, both over chunks of kNormChunk elements of one (scale, batch item): per-chunk
// partial sums (fixed order within a chunk), then every chunk's workgroup adds up its item's partials in the
// same fixed order and applies the result - no atomics, no single-workgroup-per-item bottleneck.
struct NormArgs {
  const float* disp[kMaxMaps];
  float* dispn[kMaxMaps];
  float* depth[kMaxMaps];
  const float* g_dispn[kMaxMaps];
  const float* g_depth[kMaxMaps];
  float* g_disp[kMaxMaps];
  int n[kMaxMaps];      // h*w*1 per batch item
  int chunk0[kMaxMaps + 1];  // first chunk of each scale; a scale has B * chunks(s) chunks, item-major
  int S, B;
};
constexpr int kNormChunk = 1024;  // elements per workgroup (4 per thread): the maps are a few MB - more, smaller workgroups
                                  // (4096: 190 workgroups for the whole c2 pyramid, fewer than CUs)

struct NormItem {
  int s, b, chunk, nchunk, n;
};
__device__ __forceinline__ NormItem norm_item(const NormArgs& a, int blk) {
  NormItem it;
  it.s = 0;
  while (it.s + 1 < a.S && blk >= a.chunk0[it.s + 1]) ++it.s;
  it.n = a.n[it.s];
  it.nchunk = (it.n + kNormChunk - 1) / kNormChunk;
  const int r = blk - a.chunk0[it.s];
  it.b = r / it.nchunk;
  it.chunk = r - it.b * it.nchunk;
  return it;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();  // red may still be read from a previous use
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// sum of the item's chunk partials, the same order in every thread of every workgroup of the item
__device__ __forceinline__ float item_total(const float* __restrict__ part, int first, int nchunk) {
  float t = 0.f;
  for (int k = 0; k < nchunk; ++k) t += part[first + k];
  return t;
}

// MODE 0: part[chunk] = sum d.   MODE 1: part[chunk] = sum t d  with t = g_dispn - g_depth / u^2, u = d / mean
template <int MODE>
__global__ __launch_bounds__(kBlock) void disp_partial_kernel(NormArgs a, const float* __restrict__ mean,
                                                              float* __restrict__ part) {
  __shared__ float red[4];
  const NormItem it = norm_item(a, blockIdx.x);
  const size_t base = (size_t)it.b * it.n;
  const float* __restrict__ d = a.disp[it.s] + base;
  const int k0 = it.chunk * kNormChunk, k1 = min(k0 + kNormChunk, it.n);
  float acc = 0.f;
  if (MODE == 0) {
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) acc += d[k];
  } else {
    const float m = mean[it.s * a.B + it.b];
    const float* __restrict__ gu = a.g_dispn[it.s] ? a.g_dispn[it.s] + base : nullptr;
    const float* __restrict__ gd = a.g_depth[it.s] ? a.g_depth[it.s] + base : nullptr;
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
      float t = gu ? gu[k] : 0.f;
      if (gd) {
        const float u = d[k] / m;
        t -= gd[k] / (u * u);
      }
      acc += t * d[k];
    }
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void disp_normalize_fwd_kernel(NormArgs a, const float* __restrict__ part,
                                                                    float* __restrict__ mean) {
  const NormItem it = norm_item(a, blockIdx.x);
  const float m = item_total(part, blockIdx.x - it.chunk, it.nchunk) / (float)it.n;
  if (it.chunk == 0 && threadIdx.x == 0) mean[it.s * a.B + it.b] = m;
  const size_t base = (size_t)it.b * it.n;
  const float* __restrict__ d = a.disp[it.s] + base;
  float* __restrict__ dn = a.dispn[it.s] + base;
  float* __restrict__ dp = a.depth[it.s] + base;
  const int k0 = it.chunk * kNormChunk, k1 = min(k0 + kNormChunk, it.n);
  for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
    const float q = d[k] / m;
    dn[k] = q;
    dp[k] = 1.0f / q;
  }
}

// u = d / m, depth = 1 / u.  With t_k = g_dispn_k - g_depth_k / u_k^2 (dL/du_k):
//   dL/dd_k = t_k / m - (sum_j t_j d_j) / (m^2 n)
__global__ __launch_bounds__(kBlock) void disp_normalize_bwd_kernel(NormArgs a, const float* __restrict__ part,
                                                                    const float* __restrict__ mean) {
  const NormItem it = norm_item(a, blockIdx.x);
  const float m = mean[it.s * a.B + it.b];
  const float corr = item_total(part, blockIdx.x - it.chunk, it.nchunk) / ((m * m) * (float)it.n);
  const size_t base = (size_t)it.b * it.n;
  const float* __restrict__ d = a.disp[it.s] + base;
  const float* __restrict__ gu = a.g_dispn[it.s] ? a.g_dispn[it.s] + base : nullptr;
  const float* __restrict__ gd = a.g_depth[it.s] ? a.g_depth[it.s] + base : nullptr;
  float* __restrict__ g = a.g_disp[it.s] + base;
  const int k0 = it.chunk * kNormChunk, k1 = min(k0 + kNormChunk, it.n);
  for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
    float t = gu ? gu[k] : 0.f;
    if (gd) {
      const float u = d[k] / m;
      t -= gd[k] / (u * u);
    }
    g[k] = t / m - corr;
  }
}

// ---- one launch per direction for pyramids whose items fit a workgroup's registers ---------------------------------
// (128 x 416 training images: 53 k elements per item at scale 0).  kNormSlices 1024-thread workgroups per (scale, batch
// item), largest items first.  EVERY workgroup of an item sums the whole item (16 float4 per thread, all loads in flight,
// the same order in each: identical means without any exchange) and applies the result to its own quarter, which still
// sits in its registers - the divisions, which are the cost, are spread over 4 x as many CUs as there are items, and no
// partial sums go through memory: 2 launches of ~6 us -> one.  Items beyond kNormOneMax elements or not a multiple of 4
// take the chunked pair above.
constexpr int kNormOneThreads = 1024;
constexpr int kNormSlots = 16;
constexpr int kNormSlices = 4;
constexpr int kNormOwn = kNormSlots / kNormSlices;
constexpr int kNormOneMax = kNormOneThreads * kNormSlots * 4;

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < kNormOneThreads / 64; ++k) t += red[k];
  return t;
}

template <int Q>
__device__ __forceinline__ void normalize_fwd_slice(const NormArgs& a, int s, int b, float* __restrict__ mean, float* red) {
  const int n4 = a.n[s] >> 2;
  const size_t base = (size_t)b * a.n[s];
  const float4* __restrict__ d = reinterpret_cast<const float4*>(a.disp[s] + base);
  float4* __restrict__ dn = reinterpret_cast<float4*>(a.dispn[s] + base);
  float4* __restrict__ dp = reinterpret_cast<float4*>(a.depth[s] + base);
  float4 v[kNormSlots];
#pragma unroll
  for (int j = 0; j < kNormSlots; ++j) {
    const int k = j * kNormOneThreads + threadIdx.x;
    v[j] = d[min(k, n4 - 1)];  // unconditional, selected below (see the backward)
  }
#pragma unroll
  for (int j = 0; j < kNormSlots; ++j)
    if (j * kNormOneThreads + (int)threadIdx.x >= n4) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kNormSlots; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float m = block_sum_1024(acc, red) / (float)a.n[s];
  if (Q == 0 && threadIdx.x == 0) mean[s * a.B + b] = m;
#pragma unroll
  for (int j = Q * kNormOwn; j < (Q + 1) * kNormOwn; ++j) {
    const int k = j * kNormOneThreads + threadIdx.x;
    if (k < n4) {
      const float4 q = make_float4(v[j].x / m, v[j].y / m, v[j].z / m, v[j].w / m);
      dn[k] = q;
      dp[k] = make_float4(1.0f / q.x, 1.0f / q.y, 1.0f / q.z, 1.0f / q.w);
    }
  }
}
__global__ __launch_bounds__(kNormOneThreads) void disp_normalize_fwd_item_kernel(NormArgs a, float* __restrict__ mean) {
  __shared__ float red[kNormOneThreads / 64];
  const int item = blockIdx.x / kNormSlices, q = blockIdx.x - item * kNormSlices;
  const int s = item / a.B, b = item - s * a.B;
  switch (q) {  // the slice as a compile-time constant: its slots of v[] are then registers, not an indexed array
    case 0: normalize_fwd_slice<0>(a, s, b, mean, red); break;
    case 1: normalize_fwd_slice<1>(a, s, b, mean, red); break;
    case 2: normalize_fwd_slice<2>(a, s, b, mean, red); break;
    default: normalize_fwd_slice<3>(a, s, b, mean, red); break;
  }
}

// dL/du = g_dispn - g_depth * depth^2 (the form autodiff of 1 / u takes: -g y^2 with the forward's y), depth = m / d by
// v_rcp_f32: an IEEE division costs ~11 instructions and two mode-register writes on this part, and a workgroup here
// evaluates 64 elements per thread (144 divisions per thread took 33 us; this form 6)
__device__ __forceinline__ float norm_t(float d, float gu, float gd, float m, bool has_gd) {
  float t = gu;
  if (has_gd) {
    const float y = m * __builtin_amdgcn_rcpf(d);
    t -= gd * (y * y);
  }
  return t;
}
template <int Q>
__device__ __forceinline__ void normalize_bwd_slice(const NormArgs& a, int s, int b, const float* __restrict__ mean, float* red) {
  const int n4 = a.n[s] >> 2;
  const size_t base = (size_t)b * a.n[s];
  const float m = mean[s * a.B + b];
  const float4* __restrict__ d = reinterpret_cast<const float4*>(a.disp[s] + base);
  const float4* __restrict__ gu = a.g_dispn[s] ? reinterpret_cast<const float4*>(a.g_dispn[s] + base) : nullptr;
  const float4* __restrict__ gd = a.g_depth[s] ? reinterpret_cast<const float4*>(a.g_depth[s] + base) : nullptr;
  float4* __restrict__ g = reinterpret_cast<float4*>(a.g_disp[s] + base);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* __restrict__ gup = gu ? gu : d;  // (a missing gradient: any readable address, the value is dropped)
  const float4* __restrict__ gdp = gd ? gd : d;
  float4 own[kNormOwn];
  float acc = 0.f;
  constexpr int kGroup = kNormOwn;  // chunks whose three loads are in flight together (48 registers); NOT unrolled over
                                    // the groups: the compiler then hoists all 48 loads and spills at 1024 threads
#pragma unroll 1
  for (int j0 = 0; j0 < kNormSlots; j0 += kGroup) {
    float4 dv[kGroup], uv[kGroup], dd[kGroup];
#pragma unroll
    for (int jj = 0; jj < kGroup; ++jj) {
      // unconditional loads at a clamped index, selected afterwards: a guarded load is a branch with its own wait, and
      // the 12 loads of a group then go one round trip at a time (29 us instead of 6)
      const int k = (j0 + jj) * kNormOneThreads + threadIdx.x;
      const int kc = min(k, n4 - 1);
      dv[jj] = d[kc];
      uv[jj] = gup[kc];
      dd[jj] = gdp[kc];
    }
#pragma unroll
    for (int jj = 0; jj < kGroup; ++jj) {
      const bool in = (j0 + jj) * kNormOneThreads + (int)threadIdx.x < n4;
      if (!(in && gu)) uv[jj] = z4;
      if (!(in && gd)) dd[jj] = z4;
    }
#pragma unroll
    for (int jj = 0; jj < kGroup; ++jj) {
      const float4 t = make_float4(norm_t(dv[jj].x, uv[jj].x, dd[jj].x, m, gd != nullptr), norm_t(dv[jj].y, uv[jj].y, dd[jj].y, m, gd != nullptr),
                                   norm_t(dv[jj].z, uv[jj].z, dd[jj].z, m, gd != nullptr), norm_t(dv[jj].w, uv[jj].w, dd[jj].w, m, gd != nullptr));
      acc += (t.x * dv[jj].x + t.y * dv[jj].y) + (t.z * dv[jj].z + t.w * dv[jj].w);  // (elements beyond the item: t = 0)
      if (j0 == Q * kNormOwn) own[jj] = t;
    }
  }
  const float corr = block_sum_1024(acc, red) / ((m * m) * (float)a.n[s]);
  const float rm = 1.0f / m;
#pragma unroll
  for (int j = 0; j < kNormOwn; ++j) {
    const int k = (Q * kNormOwn + j) * kNormOneThreads + threadIdx.x;
    if (k < n4) g[k] = make_float4(own[j].x * rm - corr, own[j].y * rm - corr, own[j].z * rm - corr, own[j].w * rm - corr);
  }
}
__global__ __launch_bounds__(kNormOneThreads) void disp_normalize_bwd_item_kernel(NormArgs a, const float* __restrict__ mean) {
  __shared__ float red[kNormOneThreads / 64];
  const int item = blockIdx.x / kNormSlices, q = blockIdx.x - item * kNormSlices;
  const int s = item / a.B, b = item - s * a.B;
  switch (q) {
    case 0: normalize_bwd_slice<0>(a, s, b, mean, red); break;
    case 1: normalize_bwd_slice<1>(a, s, b, mean, red); break;
    case 2: normalize_bwd_slice<2>(a, s, b, mean, red); break;
    default: normalize_bwd_slice<3>(a, s, b, mean, red); break;
  }
}

// ------------------------------------------------------------------ bilinear upsample of the pyramid
// tf.image.resize_bilinear, TF-1 defaults (align_corners=False, no half-pixel centres), resize_bilinear_op.cc:
//   scale = in / (float)out;  src = dst * scale;  lo = floor(src);  hi = min(lo + 1, in - 1);  lerp = src - lo
//   top = tl + (tr - tl) * xl;  bot = bl + (br - bl) * xl;  out = top + (bot - top) * yl
// (Lerp / lerp_of live in sfm_device.h: the fused loss kernels read the pyramid with the same arithmetic)

// C consecutive floats as the widest chunks whose alignment the NHWC layout guarantees
template <int C>
__device__ __forceinline__ void ld_vec(const float* __restrict__ p, float (&v)[C]) {
  if constexpr (C % 3 == 0) {
#pragma unroll
    for (int j = 0; j < C / 3; ++j) { const f3 t = ld3(p + 3 * j); v[3 * j] = t.x; v[3 * j + 1] = t.y; v[3 * j + 2] = t.z; }
  } else if constexpr (C % 4 == 0) {
#pragma unroll
    for (int j = 0; j < C / 4; ++j) { const float4 t = *reinterpret_cast<const float4*>(p + 4 * j); v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
  } else if constexpr (C % 2 == 0) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) { const float2 t = *reinterpret_cast<const float2*>(p + 2 * j); v[2 * j] = t.x; v[2 * j + 1] = t.y; }
  } else {
#pragma unroll
    for (int j = 0; j < C; ++j) v[j] = p[j];
  }
}
template <int C>
__device__ __forceinline__ void st_vec(float* __restrict__ p, const float (&v)[C]) {
  if constexpr (C % 3 == 0) {
#pragma unroll
    for (int j = 0; j < C / 3; ++j) st3(p + 3 * j, v[3 * j], v[3 * j + 1], v[3 * j + 2]);
  } else if constexpr (C % 4 == 0) {
#pragma unroll
    for (int j = 0; j < C / 4; ++j) *reinterpret_cast<float4*>(p + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  } else if constexpr (C % 2 == 0) {
#pragma unroll
    for (int j = 0; j < C / 2; ++j) *reinterpret_cast<float2*>(p + 2 * j) = make_float2(v[2 * j], v[2 * j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < C; ++j) p[j] = v[j];
  }
}

struct PyrArgs {
  const float* in[kMaxMaps];
  float* gin[kMaxMaps];
  int h[kMaxMaps], w[kMaxMaps];
  int blk0[kMaxMaps + 1];  // first block of each scale (1-D grid)
  long tmp0[kMaxMaps + 1]; // backward: first pixel of each scale's [B,H,w] intermediate
  float sy[kMaxMaps], sx[kMaxMaps];
  int S, B, C, H, W, accumulate;
};

__device__ __forceinline__ int scale_of_block(const PyrArgs& a, int blk) {
  int s = 0;
  while (s + 1 < a.S && blk >= a.blk0[s + 1]) ++s;
  return s;
}

// one thread per output pixel, all C channels.  out [S,B,H,W,C].
template <int C>
__global__ __launch_bounds__(kBlock) void pyramid_upsample_fwd_kernel(PyrArgs a, float* __restrict__ out) {
  const int s = scale_of_block(a, blockIdx.x);
  // 32-bit index arithmetic (B*H*W < 2^31 is checked on the host): 64-bit divisions cost more than the lerp
  const unsigned k = (blockIdx.x - a.blk0[s]) * kBlock + threadIdx.x;
  const unsigned npx = (unsigned)a.B * a.H * a.W;
  if (k >= npx) return;
  const unsigned row_ = k / (unsigned)a.W;
  const int X = (int)(k - row_ * a.W), b = (int)(row_ / (unsigned)a.H), Y = (int)(row_ - (unsigned)b * a.H);
  const int h = a.h[s], w = a.w[s];
  const float* __restrict__ in = a.in[s] + (size_t)b * h * w * C;
  float* __restrict__ o = out + ((size_t)s * npx + k) * C;
  float vo[C];
  if (h == a.H && w == a.W) {  // tf.image.resize_bilinear at the same size is the identity
    ld_vec<C>(in + ((size_t)Y * w + X) * C, vo);
    st_vec<C>(o, vo);
    return;
  }
  const Lerp ly = lerp_of(Y, a.sy[s], h), lx = lerp_of(X, a.sx[s], w);
  const float* tl = in + ((size_t)ly.lo * w + lx.lo) * C;
  const float* tr = in + ((size_t)ly.lo * w + lx.hi) * C;
  const float* bl = in + ((size_t)ly.hi * w + lx.lo) * C;
  const float* br = in + ((size_t)ly.hi * w + lx.hi) * C;
  float vtl[C], vtr[C], vbl[C], vbr[C];
  ld_vec<C>(tl, vtl); ld_vec<C>(tr, vtr); ld_vec<C>(bl, vbl); ld_vec<C>(br, vbr);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float top = vtl[c] + (vtr[c] - vtl[c]) * lx.l;
    const float bot = vbl[c] + (vbr[c] - vbl[c]) * lx.l;
    vo[c] = top + (bot - top) * ly.l;
  }
  st_vec<C>(o, vo);
}

// Adjoint (ResizeBilinearGrad) as two separable GATHER passes - the upsample is R_y (x) R_x, so its adjoint
// is R_x^T then R_y^T - with no atomics and a fixed summation order:
//   pass 1   tmp[b, Y, x, :] = sum_X  wx(X -> x) gout[b, Y, X, :]     one thread per (b, Y, x)
//   pass 2   gin[b, y, x, :] (+)= sum_Y  wy(Y -> y) tmp[b, Y, x, :]   one thread per (b, y, x)
// where wx(X -> x) = (1 - xl) if lo(X) == x, plus xl if hi(X) == x.  The candidate output columns / rows of an
// input one are bracketed generously and each is tested with the forward's own arithmetic (lerp_of), so the
// two directions can never disagree about which taps a pixel has.  A scale at full size is copied by pass 1.
template <int C>
__global__ __launch_bounds__(kBlock) void pyramid_upsample_bwd_x_kernel(PyrArgs a, const float* __restrict__ gout,
                                                                        float* __restrict__ tmp) {
  const int s = scale_of_block(a, blockIdx.x);
  const int h = a.h[s], w = a.w[s];
  const unsigned k = (blockIdx.x - a.blk0[s]) * kBlock + threadIdx.x;
  if (k >= (unsigned)a.B * a.H * w) return;
  const unsigned row_ = k / (unsigned)w;
  const int x = (int)(k - row_ * w), b = (int)(row_ / (unsigned)a.H), Y = (int)(row_ - (unsigned)b * a.H);
  const size_t npx = (size_t)a.B * a.H * a.W;
  const float* __restrict__ g = gout + ((size_t)s * npx + ((size_t)b * a.H + Y) * a.W) * C;
  float acc[C];
  if (h == a.H && w == a.W) {  // identity scale: straight to the output (pass 2 skips it)
    float* __restrict__ o = a.gin[s] + (size_t)k * C;
    ld_vec<C>(g + (size_t)x * C, acc);
    if (a.accumulate) {
      float prev[C];
      ld_vec<C>(o, prev);
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += prev[c];
    }
    st_vec<C>(o, acc);
    return;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  const float rx = (float)a.W / (float)w;
  const int X0 = max((int)floorf((float)(x - 1) * rx) - 1, 0), X1 = min((int)ceilf((float)(x + 1) * rx) + 1, a.W - 1);
  for (int X = X0; X <= X1; ++X) {
    const Lerp lx = lerp_of(X, a.sx[s], w);
    if (lx.lo != x && lx.hi != x) continue;
    const float wx = (lx.lo == x ? 1.0f - lx.l : 0.0f) + (lx.hi == x ? lx.l : 0.0f);
    float gv[C];
    ld_vec<C>(g + (size_t)X * C, gv);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += gv[c] * wx;
  }
  st_vec<C>(tmp + (a.tmp0[s] + (size_t)k) * C, acc);
}

template <int C>
__global__ __launch_bounds__(kBlock) void pyramid_upsample_bwd_y_kernel(PyrArgs a, const float* __restrict__ tmp) {
  const int s = scale_of_block(a, blockIdx.x);
  const int h = a.h[s], w = a.w[s];
  if (h == a.H && w == a.W) return;
  const unsigned k = (blockIdx.x - a.blk0[s]) * kBlock + threadIdx.x;
  if (k >= (unsigned)a.B * h * w) return;
  const unsigned row_ = k / (unsigned)w;
  const int x = (int)(k - row_ * w), b = (int)(row_ / (unsigned)h), y = (int)(row_ - (unsigned)b * h);
  const float* __restrict__ t = tmp + (a.tmp0[s] + (size_t)b * a.H * w + x) * C;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  const float ry = (float)a.H / (float)h;
  const int Y0 = max((int)floorf((float)(y - 1) * ry) - 1, 0), Y1 = min((int)ceilf((float)(y + 1) * ry) + 1, a.H - 1);
  for (int Y = Y0; Y <= Y1; ++Y) {
    const Lerp ly = lerp_of(Y, a.sy[s], h);
    if (ly.lo != y && ly.hi != y) continue;
    const float wy = (ly.lo == y ? 1.0f - ly.l : 0.0f) + (ly.hi == y ? ly.l : 0.0f);
    float tv[C];
    ld_vec<C>(t + (size_t)Y * w * C, tv);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += tv[c] * wy;
  }
  float* __restrict__ o = a.gin[s] + (size_t)k * C;
  if (a.accumulate) {
    float prev[C];
    ld_vec<C>(o, prev);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += prev[c];
  }
  st_vec<C>(o, acc);
}

// ------------------------------------------------------------------ the same resize for feature maps (C a multiple of 4)
// disp_bottleneck_dense upsamples its input x2 (nets.py:454, 465) with up to 1024 channels: one thread per
// (pixel, group of 4 channels), float4 accesses; the adjoint as the same two separable gather passes.
struct WideArgs {
  int B, h, w, G, H, W;  // G = C / 4
  float sy, sx;
};
__device__ __forceinline__ float4 f4_lerp(float4 a, float4 b, float l) {
  return make_float4(a.x + (b.x - a.x) * l, a.y + (b.y - a.y) * l, a.z + (b.z - a.z) * l, a.w + (b.w - a.w) * l);
}
__global__ __launch_bounds__(kBlock) void resize_wide_fwd_kernel(WideArgs a, const float4* __restrict__ in,
                                                                 float4* __restrict__ out) {
  const unsigned e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= (unsigned)a.B * a.H * a.W * a.G) return;
  const unsigned pix = e / a.G, g = e - pix * a.G, row = pix / a.W;
  const int X = (int)(pix - row * a.W), b = (int)(row / a.H), Y = (int)(row - (unsigned)b * a.H);
  const float4* __restrict__ src = in + (size_t)b * a.h * a.w * a.G + g;
  if (a.h == a.H && a.w == a.W) { out[e] = src[((size_t)Y * a.w + X) * a.G]; return; }
  const Lerp ly = lerp_of(Y, a.sy, a.h), lx = lerp_of(X, a.sx, a.w);
  const float4 tl = src[((size_t)ly.lo * a.w + lx.lo) * a.G], tr = src[((size_t)ly.lo * a.w + lx.hi) * a.G];
  const float4 bl = src[((size_t)ly.hi * a.w + lx.lo) * a.G], br = src[((size_t)ly.hi * a.w + lx.hi) * a.G];
  out[e] = f4_lerp(f4_lerp(tl, tr, lx.l), f4_lerp(bl, br, lx.l), ly.l);
}
// pass 1: tmp [B,H,w,G]
__global__ __launch_bounds__(kBlock) void resize_wide_bwd_x_kernel(WideArgs a, const float4* __restrict__ gout,
                                                                   float4* __restrict__ tmp) {
  const unsigned e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= (unsigned)a.B * a.H * a.w * a.G) return;
  const unsigned pix = e / a.G, g = e - pix * a.G, row = pix / a.w;  // row = b * H + Y
  const int x = (int)(pix - row * a.w);
  const float4* __restrict__ gr = gout + (size_t)row * a.W * a.G + g;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float rx = (float)a.W / (float)a.w;
  const int X0 = max((int)floorf((float)(x - 1) * rx) - 1, 0), X1 = min((int)ceilf((float)(x + 1) * rx) + 1, a.W - 1);
  for (int X = X0; X <= X1; ++X) {
    const Lerp lx = lerp_of(X, a.sx, a.w);
    if (lx.lo != x && lx.hi != x) continue;
    const float wx = (lx.lo == x ? 1.0f - lx.l : 0.0f) + (lx.hi == x ? lx.l : 0.0f);
    const float4 v = gr[(size_t)X * a.G];
    acc.x += v.x * wx; acc.y += v.y * wx; acc.z += v.z * wx; acc.w += v.w * wx;
  }
  tmp[e] = acc;
}
__global__ __launch_bounds__(kBlock) void resize_wide_bwd_y_kernel(WideArgs a, const float4* __restrict__ tmp,
                                                                   float4* __restrict__ gin) {
  const unsigned e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= (unsigned)a.B * a.h * a.w * a.G) return;
  const unsigned pix = e / a.G, g = e - pix * a.G, row = pix / a.w;
  const int x = (int)(pix - row * a.w), b = (int)(row / a.h), y = (int)(row - (unsigned)b * a.h);
  const float4* __restrict__ t = tmp + ((size_t)b * a.H * a.w + x) * a.G + g;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float ry = (float)a.H / (float)a.h;
  const int Y0 = max((int)floorf((float)(y - 1) * ry) - 1, 0), Y1 = min((int)ceilf((float)(y + 1) * ry) + 1, a.H - 1);
  for (int Y = Y0; Y <= Y1; ++Y) {
    const Lerp ly = lerp_of(Y, a.sy, a.h);
    if (ly.lo != y && ly.hi != y) continue;
    const float wy = (ly.lo == y ? 1.0f - ly.l : 0.0f) + (ly.hi == y ? ly.l : 0.0f);
    const float4 v = t[(size_t)Y * a.w * a.G];
    acc.x += v.x * wy; acc.y += v.y * wy; acc.z += v.z * wy; acc.w += v.w * wy;
  }
  gin[e] = acc;
}

static inline bool small_c(int C) { return C == 1 || C == 2 || C == 3 || C == 4 || C == 6 || C == 9 || C == 12; }
static int fill_wide(WideArgs& a, int B, int h, int w, int C, int H, int W) {
  if (B < 1 || h < 1 || w < 1 || H < h || W < w || C < 4 || C % 4) return SFM_ERR_ARG;
  if ((long)B * H * W * (C / 4) >= 2147483647L) return SFM_ERR_TOO_LARGE;
  a.B = B; a.h = h; a.w = w; a.G = C / 4; a.H = H; a.W = W;
  a.sy = (float)h / (float)H;
  a.sx = (float)w / (float)W;
  return SFM_OK;
}

// ------------------------------------------------------------------ smoothness + |t| regulariser
// compute_smooth_loss (SfMLearner_uni.py:205-219), forward differences:
//   dx = p[:, :, 1:] - p[:, :, :-1]      dy = p[:, 1:] - p[:, :-1]
//   dx2 = dx[:, :, 1:] - dx[:, :, :-1]   dxdy = dx[:, 1:] - dx[:, :-1]
//   dydx = dy[:, :, 1:] - dy[:, :, :-1]  dy2 = dy[:, 1:] - dy[:, :-1]
//   0.25 (mean|dx| + mean|dy|) + 0.75 (mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|)
// Every coefficient of the loss is known up front, so the value (7 sums per map) and the gradient come out of
// one pass: a thread owns one element, evaluates the 18 differences it takes part in (each with the
// forward's own fp32 arithmetic, so the signs agree with autodiff of the rounded values) and the 6 terms
// anchored at it.
struct SmoothK {  // gradient coefficients of one map: weight / (global element count of the term), formed on the host in fp64
  float k1x, k1y, k2xx, k2yy, k2xy, kab;
};
struct SmoothArgs {
  SmoothK kk[kMaxMaps];
  const float* map[kMaxMaps];
  float* grad[kMaxMaps];  // may be null
  int h[kMaxMaps], w[kMaxMaps], c[kMaxMaps];
  int blk0[kMaxMaps + 1];
  float w_smooth[kMaxMaps], w_abs[kMaxMaps];
  int n_maps, B;
  double b_global;
  float gscale;
};
constexpr int kSmoothTerms = 7;  // |dx|, |dy|, |dx2|, |dxdy|, |dydx|, |dy2|, |p|

__device__ __forceinline__ float sgnf(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }

constexpr int kSmoothRun = 4;                    // vertically adjacent pixels per thread
constexpr int kSmoothPerBlock = kBlock;          // runs per workgroup

// the 18 differences / 6 anchored terms of one element from its 13 samples (see the header of this section)
__device__ __forceinline__ float smooth_element(const SmoothK& c, float p00, float pm2, float pm1, float pp1, float pp2,
                                                float qm2, float qm1, float qp1, float qp2, float dmm, float dmp, float dpm,
                                                float dpp, int xi, int yi, int wi, int hi, float* __restrict__ t) {
  // first differences, named by their anchor relative to (y, x): dx_ab = DX(y + a, x + b) = P(.., +1) - P(..)
  const float dx_0m2 = pm1 - pm2, dx_0m1 = p00 - pm1, dx_00 = pp1 - p00, dx_0p1 = pp2 - pp1;
  const float dx_p1m1 = qp1 - dpm, dx_p10 = dpp - qp1, dx_m1m1 = qm1 - dmm, dx_m10 = dmp - qm1;
  const float dy_m20 = qm1 - qm2, dy_m10 = p00 - qm1, dy_00 = qp1 - p00, dy_p10 = qp2 - qp1;
  const float dy_0m1 = dpm - pm1, dy_0p1 = dpp - pp1, dy_m1m1 = pm1 - dmm, dy_m1p1 = pp1 - dmp;
  const bool xa = xi <= wi - 2, xb = xi >= 1, ya = yi <= hi - 2, yb = yi >= 1;
  // The 19 signs as INTEGERS (one v_med3_i32 each on the float's bits: the samples are canonicalised on load, so no
  // difference is -0.0), summed per coefficient and converted once: 2 compares + 2 selects + an FMA per sign term were
  // two thirds of the kernel's instructions.
  auto sg = [](float v) __attribute__((always_inline)) {
    int r;  // (written as asm: the compiler turns max(min(bits, 1), -1) back into two compares and two selects)
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v));
    return r;
  };
  auto on = [](bool cnd, int v) __attribute__((always_inline)) { return cnd ? v : 0; };
  t[6] += fabsf(p00);
  if (xa) t[0] += fabsf(dx_00);
  if (ya) t[1] += fabsf(dy_00);
  const int n1x = on(xb, sg(dx_0m1)) - on(xa, sg(dx_00));
  const int n1y = on(yb, sg(dy_m10)) - on(ya, sg(dy_00));
  const float vxx = dx_0p1 - dx_00, vyy = dy_p10 - dy_00;   // dx2 / dy2 anchored at this element
  if (xi <= wi - 3) t[2] += fabsf(vxx);
  if (yi <= hi - 3) t[5] += fabsf(vyy);
  const int nxx = on(xi <= wi - 3, sg(vxx)) - 2 * on(xb && xa, sg(dx_00 - dx_0m1)) + on(xi >= 2, sg(dx_0m1 - dx_0m2));
  const int nyy = on(yi <= hi - 3, sg(vyy)) - 2 * on(yb && ya, sg(dy_00 - dy_m10)) + on(yi >= 2, sg(dy_m10 - dy_m20));
  // dxdy = DX(y+1, x) - DX(y, x), dydx = DY(y, x+1) - DY(y, x): equal in exact arithmetic only
  const float v1 = dx_p10 - dx_00, v2 = dy_0p1 - dy_00;
  if (ya && xa) { t[3] += fabsf(v1); t[4] += fabsf(v2); }
  const int nxy = on(ya && xa, sg(v1) + sg(v2))
                - on(ya && xb, sg(dx_p1m1 - dx_0m1) + sg(dy_00 - dy_0m1))     // anchored at (y, x-1)
                - on(yb && xa, sg(dx_00 - dx_m10) + sg(dy_m1p1 - dy_m10))     // anchored at (y-1, x)
                + on(yb && xb, sg(dx_0m1 - dx_m1m1) + sg(dy_m10 - dy_m1m1));  // anchored at (y-1, x-1)
  float g = c.kab * (float)sg(p00);
  g += c.k1x * (float)n1x;
  g += c.k1y * (float)n1y;
  g += c.k2xx * (float)nxx;
  g += c.k2yy * (float)nyy;
  g += c.k2xy * (float)nxy;
  return g;
}

// One thread per RUN of kSmoothRun vertically adjacent pixels (x 3 channels when C is a multiple of 3); the lanes of a wave
// are adjacent columns / channel groups (coalesced rows).  A run shares its samples: 8 + 6 + 6 + 4 + 4 = 28 loads for four
// pixels instead of 4 x 13.  The kernel follows its VALU count.  V = channels handled together.
template <int V>
__device__ __forceinline__ void smooth_run(const SmoothK& c, const float* __restrict__ img, float* __restrict__ gimg, int C,
                                           int ch, int xi, int y0, int wi, int hi, float gs, float* __restrict__ t) {
  const int sy = wi * C;
  int ox[5], oy[8];
#pragma unroll
  for (int k = 0; k < 5; ++k) ox[k] = min(max(xi + k - 2, 0), wi - 1) * C + ch;   // column x + k - 2 (clamped: never used then)
#pragma unroll
  for (int k = 0; k < 8; ++k) oy[k] = min(max(y0 + k - 2, 0), hi - 1) * sy;        // row y0 + k - 2
  float c0[8][V], cm1[6][V], cp1[6][V], cm2[4][V], cp2[4][V];
  auto ld = [&](float (&d)[V], int o) __attribute__((always_inline)) {
    // (+ 0.0f: -0.0 -> +0.0, everything else unchanged - the integer sign of smooth_element reads the bits)
    if constexpr (V == 3) { const f3 v = ld3(img + o); d[0] = v.x + 0.0f; d[1] = v.y + 0.0f; d[2] = v.z + 0.0f; }
    else d[0] = img[o] + 0.0f;
  };
#pragma unroll
  for (int k = 0; k < 8; ++k) ld(c0[k], oy[k] + ox[2]);
#pragma unroll
  for (int k = 0; k < 6; ++k) { ld(cm1[k], oy[k + 1] + ox[1]); ld(cp1[k], oy[k + 1] + ox[3]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { ld(cm2[k], oy[k + 2] + ox[0]); ld(cp2[k], oy[k + 2] + ox[4]); }
#pragma unroll
  for (int j = 0; j < kSmoothRun; ++j) {
    const int yi = y0 + j;
    if (yi >= hi) break;
    float g[V];
#pragma unroll
    for (int v = 0; v < V; ++v)
      g[v] = gs * smooth_element(c, c0[j + 2][v], cm2[j][v], cm1[j + 1][v], cp1[j + 1][v], cp2[j][v], c0[j][v], c0[j + 1][v],
                                 c0[j + 3][v], c0[j + 4][v], cm1[j][v], cp1[j][v], cm1[j + 2][v], cp1[j + 2][v], xi, yi, wi, hi, t);
    if (gimg) {
      float* o = gimg + yi * sy + xi * C + ch;
      if constexpr (V == 3) st3(o, g[0], g[1], g[2]);
      else o[0] = g[0];
    }
  }
}

struct SmoothCoef {  // weight / (global element count) of every (map, term), fp64, formed on the host
  double k[kMaxMaps * kSmoothTerms];
};
// blockp[block] = the block's share of the regulariser: its 7 term sums (fp32, fixed order) weighted in fp64
__global__ __launch_bounds__(kBlock) void smooth_loss_kernel(SmoothArgs a, SmoothCoef kc, double* __restrict__ blockp) {
  __shared__ float red[4 * kSmoothTerms];
  __shared__ float tot[kSmoothTerms];
  int m = 0;
  while (m + 1 < a.n_maps && (int)blockIdx.x >= a.blk0[m + 1]) ++m;
  const unsigned h = a.h[m], w = a.w[m], C = a.c[m];
  const unsigned hg = (h + kSmoothRun - 1) / kSmoothRun;
  const unsigned nrun = (unsigned)a.B * hg * w;  // < 2^31 (checked on the host)
  const SmoothK c = a.kk[m];
  float t[kSmoothTerms];
#pragma unroll
  for (int j = 0; j < kSmoothTerms; ++j) t[j] = 0.f;
  // a map whose channel count is a multiple of 3 gives each group of 3 channels a thread of its own (adjacent lanes: the
  // same pixel's next 12 bytes - fully contiguous wave loads - and C / 3 times the waves of a small grid)
  const unsigned ngrp = (C % 3 == 0) ? C / 3 : 1;
  const unsigned tg = (blockIdx.x - a.blk0[m]) * kSmoothPerBlock + threadIdx.x;
  const unsigned run = tg / ngrp, grp = tg - run * ngrp;
  if (run < nrun) {
    const unsigned rowg = run / w, x = run - rowg * w, yg = rowg % hg, b = rowg / hg;
    const size_t ibase = (size_t)b * h * w * C;  // the image of this batch item
    const float* __restrict__ img = a.map[m] + ibase;
    float* __restrict__ gimg = a.grad[m] ? a.grad[m] + ibase : nullptr;
    if (C % 3 == 0) {
      smooth_run<3>(c, img, gimg, (int)C, (int)(3 * grp), (int)x, (int)(yg * kSmoothRun), (int)w, (int)h, a.gscale, t);
    } else {
      for (unsigned ch = 0; ch < C; ++ch)
        smooth_run<1>(c, img, gimg, (int)C, (int)ch, (int)x, (int)(yg * kSmoothRun), (int)w, (int)h, a.gscale, t);
    }
  }
  block_sum_store<kSmoothTerms>(t, red, tot);
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < kSmoothTerms; ++j) v += (double)tot[j] * kc.k[m * kSmoothTerms + j];
    blockp[blockIdx.x] = v;
  }
}

// one workgroup adds the block shares: thread-strided fp64 sums, then a fixed-order tree
constexpr int kSmoothReduceThreads = 256;
__global__ __launch_bounds__(kSmoothReduceThreads) void smooth_reduce_kernel(int nblk, const double* __restrict__ blockp,
                                                                             float* __restrict__ loss_out,
                                                                             double* __restrict__ loss_out_f64) {
  __shared__ double red[kSmoothReduceThreads];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four loads in flight per thread
  int k = threadIdx.x;
  for (; k + 3 * kSmoothReduceThreads < nblk; k += 4 * kSmoothReduceThreads) {
    a0 += blockp[k];
    a1 += blockp[k + kSmoothReduceThreads];
    a2 += blockp[k + 2 * kSmoothReduceThreads];
    a3 += blockp[k + 3 * kSmoothReduceThreads];
  }
  for (; k < nblk; k += kSmoothReduceThreads) a0 += blockp[k];
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int o = kSmoothReduceThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double total = red[0];
    if (loss_out) *loss_out = (float)total;
    if (loss_out_f64) *loss_out_f64 = (double)(float)total;  // the fp32 value, in the all-reduce buffer's type
  }
}

// ------------------------------------------------------------------ upstream-gradient scaling
// The fused loss computes its gradients with the value (dL/dloss = 1); autograd hands the upstream scalar g over later,
// as a device tensor.  sfm_scale_grads multiplies every listed buffer by *g in place - and does nothing at all when
// *g == 1, which is every plain loss.backward(): one launch that exits at once instead of a pass over all gradients.
struct ScaleArgs {
  float* p[2 * kMaxMaps + 2];
  long n[2 * kMaxMaps + 2];
  int blk0[2 * kMaxMaps + 3];
  int count;
};
__global__ __launch_bounds__(kBlock) void scale_grads_kernel(ScaleArgs a, const float* __restrict__ g) {
  const float f = *g;
  if (f == 1.0f) return;
  int m = 0;
  while (m + 1 < a.count && (int)blockIdx.x >= a.blk0[m + 1]) ++m;
  const long k = ((long)blockIdx.x - a.blk0[m]) * kBlock * 4 + threadIdx.x * 4;
  float* p = a.p[m];
  if (k + 3 < a.n[m] && (((uintptr_t)p) & 15) == 0) {
    float4 v = *reinterpret_cast<float4*>(p + k);
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    *reinterpret_cast<float4*>(p + k) = v;
  } else {
    for (long j = k; j < a.n[m] && j < k + 4; ++j) p[j] *= f;
  }
}

static inline int blocks_for(long n) { return (int)((n + kBlock - 1) / kBlock); }

// mode: grid over 0 = output pixels [B,H,W], 1 = the backward intermediate [B,H,w_s], 2 = input pixels [B,h_s,w_s]
static int fill_pyr(PyrArgs& a, const int* h, const int* w, int S, int B, int C, int H, int W, int mode) {
  if (!h || !w || S < 1 || S > kMaxMaps || B < 1 || H < 1 || W < 1 || C < 1 || C > kMaxC) return SFM_ERR_ARG;
  if (!sfm_fits(H, W, C) || (long)B * H * W >= 2147483647L) return SFM_ERR_TOO_LARGE;
  a.S = S; a.B = B; a.C = C; a.H = H; a.W = W;
  long blk = 0;
  for (int s = 0; s < S; ++s) {
    if (h[s] < 1 || w[s] < 1 || h[s] > H || w[s] > W) return SFM_ERR_ARG;  // the pyramid is only ever upsampled
    a.h[s] = h[s]; a.w[s] = w[s];
    a.sy[s] = (float)h[s] / (float)H;  // CalculateResizeScale: in / static_cast<float>(out)
    a.sx[s] = (float)w[s] / (float)W;
    a.blk0[s] = (int)blk;
    blk += blocks_for(mode == 0 ? (long)B * H * W : (mode == 1 ? (long)B * H * w[s] : (long)B * h[s] * w[s]));
    if (blk > 2147483647L) return SFM_ERR_TOO_LARGE;
  }
  a.blk0[S] = (int)blk;
  long t = 0;
  for (int s = 0; s < S; ++s) {
    a.tmp0[s] = t;
    if (h[s] != H || w[s] != W) t += (long)B * H * w[s];  // a scale at full size needs no intermediate
  }
  a.tmp0[S] = t;
  a.accumulate = 0;
  return SFM_OK;
}

}  // namespace
}  // namespace sfm

using namespace sfm;

#define SFM_DISPATCH_C(C, KERNEL, ...)                                                  \
  switch (C) {                                                                          \
    case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                          \
    case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                          \
    case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                          \
    case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                          \
    case 6: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                          \
    case 9: hipLaunchKernelGGL(KERNEL<9>, __VA_ARGS__); break;                          \
    case 12: hipLaunchKernelGGL(KERNEL<12>, __VA_ARGS__); break;                        \
    default: return SFM_ERR_ARG;                                                        \
  }

extern "C" int sfm_pyramid_upsample_fwd(const float* const* in, const int* h, const int* w, int S, int B, int C, int H,
                                        int W, float* out, void* stream) {
  PyrArgs a;
  const int rc = fill_pyr(a, h, w, S, B, C, H, W, 0);
  if (rc) return rc;
  if (!in || !out) return SFM_ERR_ARG;
  for (int s = 0; s < S; ++s) {
    if (!in[s]) return SFM_ERR_ARG;
    a.in[s] = in[s];
    a.gin[s] = nullptr;
  }
  SFM_DISPATCH_C(C, pyramid_upsample_fwd_kernel, dim3((unsigned)a.blk0[S]), dim3(kBlock), 0, (hipStream_t)stream, a, out)
  return sfm_last_error();
}

extern "C" long long sfm_pyramid_upsample_bwd_workspace_floats(const int* h, const int* w, int S, int B, int C, int H,
                                                               int W) {
  PyrArgs a;
  const int rc = fill_pyr(a, h, w, S, B, C, H, W, 1);
  if (rc) return rc;
  return (long long)a.tmp0[S] * C + 4;
}

extern "C" int sfm_pyramid_upsample_bwd(const float* gout, const int* h, const int* w, int S, int B, int C, int H, int W,
                                        int accumulate, float* const* gin, float* ws, void* stream) {
  PyrArgs a;
  int rc = fill_pyr(a, h, w, S, B, C, H, W, 1);
  if (rc) return rc;
  if (!gin || !gout || !ws) return SFM_ERR_ARG;
  for (int s = 0; s < S; ++s) {
    if (!gin[s]) return SFM_ERR_ARG;
    a.in[s] = nullptr;
    a.gin[s] = gin[s];
  }
  a.accumulate = accumulate != 0;
  SFM_DISPATCH_C(C, pyramid_upsample_bwd_x_kernel, dim3((unsigned)a.blk0[S]), dim3(kBlock), 0, (hipStream_t)stream, a, gout, ws)
  PyrArgs a2 = a;
  rc = fill_pyr(a2, h, w, S, B, C, H, W, 2);
  if (rc) return rc;
  for (int s = 0; s < S; ++s) a2.gin[s] = gin[s];
  a2.accumulate = accumulate != 0;
  SFM_DISPATCH_C(C, pyramid_upsample_bwd_y_kernel, dim3((unsigned)a2.blk0[S]), dim3(kBlock), 0, (hipStream_t)stream, a2, ws)
  return sfm_last_error();
}

extern "C" int sfm_scale_grads(float* const* bufs, const long long* n, int count, const float* g, void* stream) {
  if (!bufs || !n || !g || count < 1 || count > 2 * kMaxMaps + 2) return SFM_ERR_ARG;
  ScaleArgs a;
  long blk = 0;
  for (int m = 0; m < count; ++m) {
    if (!bufs[m] || n[m] < 1) return SFM_ERR_ARG;
    a.p[m] = bufs[m];
    a.n[m] = (long)n[m];
    a.blk0[m] = (int)blk;
    blk += (n[m] + kBlock * 4 - 1) / (kBlock * 4);
    if (blk > 2147483647L) return SFM_ERR_TOO_LARGE;
  }
  a.blk0[count] = (int)blk;
  a.count = count;
  hipLaunchKernelGGL(scale_grads_kernel, dim3((unsigned)blk), dim3(kBlock), 0, (hipStream_t)stream, a, g);
  return sfm_last_error();
}

extern "C" int sfm_resize_bilinear_fwd(const float* in, int B, int h, int w, int C, int H, int W, float* out, void* stream) {
  if (small_c(C)) return sfm_pyramid_upsample_fwd(&in, &h, &w, 1, B, C, H, W, out, stream);
  WideArgs a;
  const int rc = fill_wide(a, B, h, w, C, H, W);
  if (rc) return rc;
  if (!in || !out) return SFM_ERR_ARG;
  const long n = (long)B * H * W * a.G;
  hipLaunchKernelGGL(resize_wide_fwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a,
                     reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out));
  return sfm_last_error();
}

extern "C" long long sfm_resize_bilinear_bwd_workspace_floats(int B, int h, int w, int C, int H, int W) {
  if (small_c(C)) return sfm_pyramid_upsample_bwd_workspace_floats(&h, &w, 1, B, C, H, W);
  WideArgs a;
  const int rc = fill_wide(a, B, h, w, C, H, W);
  if (rc) return rc;
  return (long long)B * H * w * C + 4;
}

extern "C" int sfm_resize_bilinear_bwd(const float* gout, int B, int h, int w, int C, int H, int W, float* gin, float* ws,
                                       void* stream) {
  if (small_c(C)) return sfm_pyramid_upsample_bwd(gout, &h, &w, 1, B, C, H, W, 0, &gin, ws, stream);
  WideArgs a;
  const int rc = fill_wide(a, B, h, w, C, H, W);
  if (rc) return rc;
  if (!gout || !gin || !ws || ((uintptr_t)ws & 15)) return SFM_ERR_ARG;
  if (h == H && w == W) {
    const hipError_t e = hipMemcpyAsync(gin, gout, (size_t)B * H * W * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    return e == hipSuccess ? SFM_OK : (int)e;
  }
  const long n1 = (long)B * H * w * a.G, n2 = (long)B * h * w * a.G;
  hipLaunchKernelGGL(resize_wide_bwd_x_kernel, dim3((unsigned)((n1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     a, reinterpret_cast<const float4*>(gout), reinterpret_cast<float4*>(ws));
  hipLaunchKernelGGL(resize_wide_bwd_y_kernel, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     a, reinterpret_cast<const float4*>(ws), reinterpret_cast<float4*>(gin));
  return sfm_last_error();
}

static int fill_norm(NormArgs& a, const float* const* disp, const int* n, int S, int B) {
  if (!disp || !n || S < 1 || S > kMaxMaps || B < 1) return SFM_ERR_ARG;
  long chunk = 0;
  for (int s = 0; s < S; ++s) {
    if (!disp[s] || n[s] < 1) return SFM_ERR_ARG;
    a.disp[s] = disp[s];
    a.n[s] = n[s];
    a.dispn[s] = a.depth[s] = a.g_disp[s] = nullptr;
    a.g_dispn[s] = a.g_depth[s] = nullptr;
    a.chunk0[s] = (int)chunk;
    chunk += (long)B * ((n[s] + kNormChunk - 1) / kNormChunk);
    if (chunk > 2147483647L) return SFM_ERR_TOO_LARGE;
  }
  a.chunk0[S] = (int)chunk;
  a.S = S;
  a.B = B;
  return SFM_OK;
}

// every item a multiple of 4 elements, at most kNormOneMax, every listed pointer 16-byte aligned: the one-launch form
static bool norm_one_launch(const NormArgs& a) {
  for (int s = 0; s < a.S; ++s) {
    if (a.n[s] % 4 || a.n[s] > kNormOneMax) return false;
    const void* ps[6] = {a.disp[s], a.dispn[s], a.depth[s], a.g_dispn[s], a.g_depth[s], a.g_disp[s]};
    for (const void* q : ps)
      if ((uintptr_t)q & 15) return false;
  }
  return (long)a.S * a.B * kNormSlices <= 2147483647L;
}

extern "C" long long sfm_disp_normalize_workspace_floats(const int* n, int S, int B) {
  if (!n || S < 1 || S > kMaxMaps || B < 1) return SFM_ERR_ARG;
  long chunk = 0;
  for (int s = 0; s < S; ++s) {
    if (n[s] < 1) return SFM_ERR_ARG;
    chunk += (long)B * ((n[s] + kNormChunk - 1) / kNormChunk);
  }
  return chunk;
}

extern "C" int sfm_disp_normalize_fwd(const float* const* disp, const int* n, int S, int B, float* const* disp_norm,
                                      float* const* depth, float* mean, float* ws, void* stream) {
  NormArgs a;
  const int rc = fill_norm(a, disp, n, S, B);
  if (rc) return rc;
  if (!disp_norm || !depth || !mean || !ws) return SFM_ERR_ARG;
  for (int s = 0; s < S; ++s) {
    if (!disp_norm[s] || !depth[s]) return SFM_ERR_ARG;
    a.dispn[s] = disp_norm[s];
    a.depth[s] = depth[s];
  }
  if (norm_one_launch(a)) {
    hipLaunchKernelGGL(disp_normalize_fwd_item_kernel, dim3((unsigned)(S * B * kNormSlices)), dim3(kNormOneThreads), 0, (hipStream_t)stream, a,
                       mean);
    return sfm_last_error();
  }
  const dim3 grid((unsigned)a.chunk0[S]);
  hipLaunchKernelGGL(disp_partial_kernel<0>, grid, dim3(kBlock), 0, (hipStream_t)stream, a, (const float*)nullptr, ws);
  hipLaunchKernelGGL(disp_normalize_fwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, a, ws, mean);
  return sfm_last_error();
}

extern "C" int sfm_disp_normalize_bwd(const float* const* disp, const float* mean, const float* const* g_disp_norm,
                                      const float* const* g_depth, const int* n, int S, int B, float* const* g_disp,
                                      float* ws, void* stream) {
  NormArgs a;
  const int rc = fill_norm(a, disp, n, S, B);
  if (rc) return rc;
  if (!mean || !g_disp || !ws) return SFM_ERR_ARG;
  for (int s = 0; s < S; ++s) {
    if (!g_disp[s]) return SFM_ERR_ARG;
    a.g_disp[s] = g_disp[s];
    a.g_dispn[s] = g_disp_norm ? g_disp_norm[s] : nullptr;
    a.g_depth[s] = g_depth ? g_depth[s] : nullptr;
  }
  if (norm_one_launch(a)) {
    hipLaunchKernelGGL(disp_normalize_bwd_item_kernel, dim3((unsigned)(S * B * kNormSlices)), dim3(kNormOneThreads), 0, (hipStream_t)stream, a,
                       mean);
    return sfm_last_error();
  }
  const dim3 grid((unsigned)a.chunk0[S]);
  hipLaunchKernelGGL(disp_partial_kernel<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, a, mean, ws);
  hipLaunchKernelGGL(disp_normalize_bwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, a, ws, mean);
  return sfm_last_error();
}

static int fill_smooth(SmoothArgs& a, const int* h, const int* w, const int* c, int n_maps, int B) {
  if (!h || !w || !c || n_maps < 1 || n_maps > kMaxMaps || B < 1) return SFM_ERR_ARG;
  long blk = 0;
  for (int m = 0; m < n_maps; ++m) {
    if (h[m] < 3 || w[m] < 3 || c[m] < 1) return SFM_ERR_ARG;  // the second differences need three samples per axis
    a.h[m] = h[m]; a.w[m] = w[m]; a.c[m] = c[m];
    a.blk0[m] = (int)blk;
    const long nel = (long)B * h[m] * w[m] * c[m];
    if (nel > 2147483647L) return SFM_ERR_TOO_LARGE;
    const long nrun = (long)B * ((h[m] + kSmoothRun - 1) / kSmoothRun) * w[m] * (c[m] % 3 == 0 ? c[m] / 3 : 1);  // threads
    if (nrun > 2147483647L) return SFM_ERR_TOO_LARGE;
    blk += (nrun + kSmoothPerBlock - 1) / kSmoothPerBlock;
  }
  a.blk0[n_maps] = (int)blk;
  a.n_maps = n_maps;
  a.B = B;
  return SFM_OK;
}

extern "C" long long sfm_smooth_workspace_floats(const int* h, const int* w, const int* c, int n_maps, int B) {
  SmoothArgs a;
  const int rc = fill_smooth(a, h, w, c, n_maps, B);
  if (rc) return rc;
  return (long long)a.blk0[n_maps] * kSmoothTerms;
}

extern "C" int sfm_smooth_loss(const float* const* maps, float* const* d_maps, const int* h, const int* w, const int* c,
                               const float* w_smooth, const float* w_abs, int n_maps, int B, int B_global, float gscale,
                               float* ws, float* loss_out, double* loss_out_f64, void* stream) {
  SmoothArgs a;
  const int rc = fill_smooth(a, h, w, c, n_maps, B);
  if (rc) return rc;
  if (!maps || !w_smooth || !w_abs || !ws || ((uintptr_t)ws & 7) || (!loss_out && !loss_out_f64) || B_global < B)
    return SFM_ERR_ARG;
  for (int m = 0; m < n_maps; ++m) {
    if (!maps[m]) return SFM_ERR_ARG;
    a.map[m] = maps[m];
    a.grad[m] = d_maps ? d_maps[m] : nullptr;
    a.w_smooth[m] = w_smooth[m];
    a.w_abs[m] = w_abs[m];
  }
  a.b_global = (double)B_global;
  a.gscale = gscale;
  SmoothCoef kc;
  for (int m = 0; m < n_maps; ++m) {
    const double hh = h[m], ww = w[m], C = c[m], bg = a.b_global, wsm = w_smooth[m];
    const double den[kSmoothTerms] = {bg * hh * (ww - 1) * C, bg * (hh - 1) * ww * C, bg * hh * (ww - 2) * C,
                                      bg * (hh - 1) * (ww - 1) * C, bg * (hh - 1) * (ww - 1) * C, bg * (hh - 2) * ww * C,
                                      bg * hh * ww * C};
    const double coef[kSmoothTerms] = {0.25 * wsm, 0.25 * wsm, 0.75 * wsm, 0.75 * wsm, 0.75 * wsm, 0.75 * wsm, (double)w_abs[m]};
    for (int j = 0; j < kSmoothTerms; ++j) kc.k[m * kSmoothTerms + j] = coef[j] / den[j];
    const double* k = kc.k + m * kSmoothTerms;  // the same quotients, rounded once, weight the gradient
    a.kk[m] = SmoothK{(float)k[0], (float)k[1], (float)k[2], (float)k[5], (float)k[3], (float)k[6]};
  }
  double* blockp = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(smooth_loss_kernel, dim3((unsigned)a.blk0[n_maps]), dim3(kBlock), 0, (hipStream_t)stream, a, kc, blockp);
  hipLaunchKernelGGL(smooth_reduce_kernel, dim3(1), dim3(kSmoothReduceThreads), 0, (hipStream_t)stream, a.blk0[n_maps], blockp,
                     loss_out, loss_out_f64);
  return sfm_last_error();
}
