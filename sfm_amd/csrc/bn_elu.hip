// The normaliser + activation slim appends to every slim.conv2d of disp_bottleneck_dense (nets.py:455-470) through the
// arg-scope of disp_aspp_u_dense (nets.py:653-656 -> resnet/resnet_utils.py:232-251): slim.batch_norm(decay, epsilon,
// scale=True, center=True, fused=True, is_training=True, updates_collections=None) followed by tf.nn.elu
// (SURVEY.md section 8f row 3).  x is the convolution's output, [M, C] with M = B*H*W rows (NHWC).  Per channel:
//
//   mean = E[x]            var = E[(x - mean)^2]   (population variance: what fused_batch_norm normalises with)
//   xhat = (x - mean) rsqrt(var + eps)             z = gamma xhat + beta            y = z > 0 ? z : exp(z) - 1
//   moving_mean -= (moving_mean - mean) * fp32(1 - decay) ;  moving_var likewise with var M / (M - 1)   (in place;
//   moving_averages.assign_moving_average: the factor 1 - decay is formed in double on the host, then cast)
//
// and the adjoint (TF autodiff: EluGrad on the activations, FusedBatchNormGrad):
//
//   dz = gy (y < 0 ? y + 1 : 1)      gbeta = sum dz      ggamma = sum dz xhat
//   gx = gamma rsqrt(var + eps) (dz - gbeta / M - xhat ggamma / M)
//
// HBM-bound elementwise work around two per-channel reductions.  Reductions: a workgroup sums a chunk of rows in fp64
// (lanes own 4 adjacent channels, so every load is a coalesced float4), chunk partials are added in chunk order by the
// finalising kernel: no atomics, reproducible.  Forward = stats + finalise + apply (x is read twice); backward = sums +
// finalise + apply.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {
namespace {

struct BnPlan {
  int groups;      // C / 4 channel groups of a row
  int gslab;       // channel groups one workgroup covers (<= kBlock); wider rows are cut into slabs along grid.y
  int slabs;
  int row_lanes;   // kBlock / gslab: rows a workgroup reads per iteration
  long rows_per_chunk;
  int chunks;
};

static inline bool bn_plan(long M, int C, BnPlan& p) {
  if (M < 1 || C < 4 || C % 4 || C > 65536) return false;
  p.groups = C / 4;
  p.gslab = p.groups < kBlock ? p.groups : kBlock;
  p.slabs = (p.groups + p.gslab - 1) / p.gslab;
  p.row_lanes = kBlock / p.gslab;  // threads beyond gslab * row_lanes idle (C / 4 need not divide 256)
  long chunks = (M + p.row_lanes - 1) / p.row_lanes;  // at least one iteration each
  if (chunks > 512) chunks = 512;
  p.rows_per_chunk = ((M + chunks - 1) / chunks + p.row_lanes - 1) / p.row_lanes * p.row_lanes;
  p.chunks = (int)((M + p.rows_per_chunk - 1) / p.rows_per_chunk);
  return true;
}

__device__ __forceinline__ float elu_grad_from_y(float y) { return y < 0.0f ? y + 1.0f : 1.0f; }

// dL/dy as the SUM of up to three tensors, each a channel slice of its own buffer (row stride ld floats): what the
// concat-free decoder level hands over - a slice of the gradient of the concat buffer plus the input gradients of the
// convolutions that read the same activation (nets.py:458, 463, 467) - without a pass that adds them up first.
struct GySrc {
  const float* p[3];
  int ld[3];
  int n;
};
__device__ __forceinline__ float4 gy_load(const GySrc& s, long row, int g) {
  float4 v = *reinterpret_cast<const float4*>(s.p[0] + row * s.ld[0] + 4 * g);
  for (int k = 1; k < s.n; ++k) {  // uniform
    const float4 u = *reinterpret_cast<const float4*>(s.p[k] + row * s.ld[k] + 4 * g);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  return v;
}

// part [chunks][2][C] (fp64): MODE 0: (sum x, sum x^2 about `shift`);  MODE 1: (sum dz, sum dz xhat)
// The forward sums are taken about a per-channel shift (the channel's first row) so that E[d^2] - E[d]^2 does not cancel.
template <int MODE>
__global__ __launch_bounds__(kBlock) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, int ldy,
                                                            GySrc gy, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, long M, int C, BnPlan p,
                                                            int act, double* __restrict__ part) {
  __shared__ double red[kBlock][8];
  const int gl = threadIdx.x % p.gslab, rl = threadIdx.x / p.gslab;
  const int g = min((int)blockIdx.y * p.gslab + gl, p.groups - 1);  // (the tail of the last slab repeats its last group)
  const bool mine = (int)blockIdx.y * p.gslab + gl < p.groups;
  const long r0 = (long)blockIdx.x * p.rows_per_chunk, r1 = min(M, r0 + p.rows_per_chunk);
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = s4, i4 = s4;
  if (MODE == 0) s4 = *reinterpret_cast<const float4*>(x + 4 * g);  // row 0 of the channel group: the shift
  else {
    m4 = *reinterpret_cast<const float4*>(mean + 4 * g);
    i4 = *reinterpret_cast<const float4*>(invstd + 4 * g);
  }
  for (long r = (rl < p.row_lanes && mine) ? r0 + rl : r1; r < r1; r += p.row_lanes) {
    const size_t o = (size_t)r * C + 4 * g;
    const float4 xv = *reinterpret_cast<const float4*>(x + o);
    if (MODE == 0) {
      const float d0 = xv.x - s4.x, d1 = xv.y - s4.y, d2 = xv.z - s4.z, d3 = xv.w - s4.w;
      a[0] += d0; a[1] += d1; a[2] += d2; a[3] += d3;
      b[0] += (double)d0 * d0; b[1] += (double)d1 * d1; b[2] += (double)d2 * d2; b[3] += (double)d3 * d3;
    } else {
      const float4 yv = *reinterpret_cast<const float4*>(y + (size_t)r * ldy + 4 * g), gv = gy_load(gy, r, g);
      const float z0 = act ? gv.x * elu_grad_from_y(yv.x) : gv.x, z1 = act ? gv.y * elu_grad_from_y(yv.y) : gv.y;
      const float z2 = act ? gv.z * elu_grad_from_y(yv.z) : gv.z, z3 = act ? gv.w * elu_grad_from_y(yv.w) : gv.w;
      a[0] += z0; a[1] += z1; a[2] += z2; a[3] += z3;
      b[0] += (double)(z0 * ((xv.x - m4.x) * i4.x)); b[1] += (double)(z1 * ((xv.y - m4.y) * i4.y));
      b[2] += (double)(z2 * ((xv.z - m4.z) * i4.z)); b[3] += (double)(z3 * ((xv.w - m4.w) * i4.w));
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[threadIdx.x][k] = a[k]; red[threadIdx.x][4 + k] = b[k]; }
  __syncthreads();
  if (rl == 0 && mine) {
    double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    for (int j = 0; j < p.row_lanes; ++j)  // fixed order
#pragma unroll
      for (int k = 0; k < 4; ++k) { sa[k] += red[j * p.gslab + gl][k]; sb[k] += red[j * p.gslab + gl][4 + k]; }
    double* o = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[4 * g + k] = sa[k]; o[C + 4 * g + k] = sb[k]; }
  }
}

// one WAVE per channel: lanes stride over the chunk partials, then a shuffle tree (a fixed order) -> statistics (and the
// moving averages) / parameter gradients.  grid: ceil(C / 4) workgroups of 4 waves.
__device__ __forceinline__ void bn_chunk_sums(const double* __restrict__ part, int C, int chunks, int c, double& s, double& q) {
  const int lane = threadIdx.x & 63;
  s = 0.0; q = 0.0;
  for (int k = lane; k < chunks; k += 64) { s += part[(size_t)k * 2 * C + c]; q += part[(size_t)k * 2 * C + C + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); q += __shfl_down(q, o, 64); }
}

__global__ __launch_bounds__(kBlock) void bn_finalize_fwd_kernel(const double* __restrict__ part, const float* __restrict__ x,
                                                                 long M, int C, int chunks, float eps, float one_minus_decay,
                                                                 float* __restrict__ moving_mean,
                                                                 float* __restrict__ moving_var,
                                                                 float* __restrict__ save_mean,
                                                                 float* __restrict__ save_invstd) {
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  bn_chunk_sums(part, C, chunks, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  const double md = s / (double)M;                       // mean of (x - shift)
  const double var = fmax(q / (double)M - md * md, 0.0);  // population variance
  const double mean = (double)x[c] + md;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (moving_mean) moving_mean[c] = moving_mean[c] - (moving_mean[c] - (float)mean) * one_minus_decay;
  if (moving_var) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    moving_var[c] = moving_var[c] - (moving_var[c] - (float)unbiased) * one_minus_decay;
  }
}

// ---- synchronised statistics (data parallel): a rank publishes (n, mean, M2) per channel in fp64, the
// ranks' triples are gathered and every rank merges them in RANK ORDER with Chan's pairwise update - the same numbers on
// every rank, exact up to fp64 rounding whatever the ranks' own means are (no common shift needed).
__global__ __launch_bounds__(kBlock) void bn_local_stats_kernel(const double* __restrict__ part, const float* __restrict__ x,
                                                                long M, int C, int chunks, double* __restrict__ stats) {
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  bn_chunk_sums(part, C, chunks, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  stats[c] = (double)M;
  stats[C + c] = (double)x[c] + s / (double)M;
  stats[2 * C + c] = fmax(q - s * s / (double)M, 0.0);
}

// stats [world][3][C] -> batch statistics over all ranks (and the moving averages, as bn_finalize_fwd_kernel)
__global__ __launch_bounds__(kBlock) void bn_merge_stats_kernel(const double* __restrict__ stats, int world, int C, float eps,
                                                                float one_minus_decay, float* __restrict__ moving_mean,
                                                                float* __restrict__ moving_var,
                                                                float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* sr = stats + (size_t)r * 3 * C;
    const double nb = sr[c], mb = sr[C + c], m2b = sr[2 * C + c];
    if (!(nb > 0.0)) continue;
    const double nn = n + nb, delta = mb - mean;
    mean += delta * (nb / nn);
    m2 += m2b + delta * delta * (n * nb / nn);
    n = nn;
  }
  const double var = n > 0.0 ? m2 / n : 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (moving_mean) moving_mean[c] = moving_mean[c] - (moving_mean[c] - (float)mean) * one_minus_decay;
  if (moving_var) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    moving_var[c] = moving_var[c] - (moving_var[c] - (float)unbiased) * one_minus_decay;
  }
}

__global__ __launch_bounds__(kBlock) void bn_finalize_bwd_kernel(const double* __restrict__ part, int C, int chunks,
                                                                 float* __restrict__ ggamma, float* __restrict__ gbeta) {
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  bn_chunk_sums(part, C, chunks, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  gbeta[c] = (float)s;
  ggamma[c] = (float)q;
}

__device__ __forceinline__ float elu(float z) { return z > 0.0f ? z : expf(z) - 1.0f; }

// one thread per float4 of x
__global__ __launch_bounds__(kBlock) void bn_elu_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, long n4, int groups,
                                                              int act, float* __restrict__ y, int ldy) {
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n4) return;
  const long row = e / groups;
  const int g = (int)(e - row * groups);
  const float4 xv = reinterpret_cast<const float4*>(x)[e];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[g], be = reinterpret_cast<const float4*>(beta)[g];
  const float4 m = reinterpret_cast<const float4*>(mean)[g], is = reinterpret_cast<const float4*>(invstd)[g];
  float4 o;
  o.x = ga.x * ((xv.x - m.x) * is.x) + be.x;
  o.y = ga.y * ((xv.y - m.y) * is.y) + be.y;
  o.z = ga.z * ((xv.z - m.z) * is.z) + be.z;
  o.w = ga.w * ((xv.w - m.w) * is.w) + be.w;
  if (act) { o.x = elu(o.x); o.y = elu(o.y); o.z = elu(o.z); o.w = elu(o.w); }
  *reinterpret_cast<float4*>(y + row * ldy + 4 * g) = o;  // ldy = C: y is a tensor of its own; more: a slice of the concat buffer
}

// inference (is_training=False): the moving statistics instead of the batch's, y = gamma (x - mm) rsqrt(mv + eps) + beta
__global__ __launch_bounds__(kBlock) void bn_elu_infer_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mm,
                                                              const float* __restrict__ mv, float eps, long n4, int groups,
                                                              int act, float* __restrict__ y, int ldy) {
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n4) return;
  const long row = e / groups;
  const int g = (int)(e - row * groups);
  const float4 xv = reinterpret_cast<const float4*>(x)[e];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[g], be = reinterpret_cast<const float4*>(beta)[g];
  const float4 m = reinterpret_cast<const float4*>(mm)[g], v = reinterpret_cast<const float4*>(mv)[g];
  float4 o;
  o.x = ga.x * ((xv.x - m.x) * (1.0f / sqrtf(v.x + eps))) + be.x;
  o.y = ga.y * ((xv.y - m.y) * (1.0f / sqrtf(v.y + eps))) + be.y;
  o.z = ga.z * ((xv.z - m.z) * (1.0f / sqrtf(v.z + eps))) + be.z;
  o.w = ga.w * ((xv.w - m.w) * (1.0f / sqrtf(v.w + eps))) + be.w;
  if (act) { o.x = elu(o.x); o.y = elu(o.y); o.z = elu(o.z); o.w = elu(o.w); }
  *reinterpret_cast<float4*>(y + row * ldy + 4 * g) = o;
}

__global__ __launch_bounds__(kBlock) void bn_elu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  int ldy, GySrc gy, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ ggamma,
                                                                  const float* __restrict__ gbeta, long n4, int groups,
                                                                  float inv_m, int act, float* __restrict__ gx) {
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n4) return;
  const long row = e / groups;
  const int g = (int)(e - row * groups);
  const float4 xv = reinterpret_cast<const float4*>(x)[e], yv = *reinterpret_cast<const float4*>(y + row * ldy + 4 * g);
  const float4 gv = gy_load(gy, row, g);
  const float4 ga = reinterpret_cast<const float4*>(gamma)[g];
  const float4 m = reinterpret_cast<const float4*>(mean)[g], is = reinterpret_cast<const float4*>(invstd)[g];
  const float4 gg = reinterpret_cast<const float4*>(ggamma)[g], gb = reinterpret_cast<const float4*>(gbeta)[g];
  auto one = [&](float xs, float ys, float gs, float gam, float mu, float iv, float dgam, float dbet) {
    const float dz = act ? gs * elu_grad_from_y(ys) : gs;
    const float xh = (xs - mu) * iv;
    return gam * iv * (dz - dbet * inv_m - xh * (dgam * inv_m));
  };
  float4 o;
  o.x = one(xv.x, yv.x, gv.x, ga.x, m.x, is.x, gg.x, gb.x);
  o.y = one(xv.y, yv.y, gv.y, ga.y, m.y, is.y, gg.y, gb.y);
  o.z = one(xv.z, yv.z, gv.z, ga.z, m.z, is.z, gg.z, gb.z);
  o.w = one(xv.w, yv.w, gv.w, ga.w, m.w, is.w, gg.w, gb.w);
  reinterpret_cast<float4*>(gx)[e] = o;
}

}  // namespace
}  // namespace sfm

using namespace sfm;

// scratch of both calls: the chunk partials [chunks][2][C] in fp64, counted in floats
extern "C" long long sfm_bn_elu_workspace_floats(long long M, int C) {
  BnPlan p;
  if (!bn_plan((long)M, C, p)) return SFM_ERR_ARG;
  return (long long)p.chunks * 2 * C * 2 + 4;
}

static bool slice_ok(const float* q, int ld, int C) { return q && ld >= C && ld % 4 == 0 && ((uintptr_t)q & 15) == 0; }

// y may be a channel slice of a wider [M, ld_y] buffer (the level's concat buffer): ld_y floats between its rows
extern "C" int sfm_bn_elu_fwd_ld(const float* x, const float* gamma, const float* beta, long long M, int C, float eps,
                                 double decay, int act, float* moving_mean, float* moving_var, float* y, int ld_y,
                                 float* save_mean, float* save_invstd, float* ws, void* stream_) {
  BnPlan p;
  if (!x || !gamma || !beta || !slice_ok(y, ld_y, C) || !save_mean || !save_invstd || !ws || !bn_plan((long)M, C, p) ||
      !(eps > 0.0f))
    return SFM_ERR_ARG;
  if (((uintptr_t)ws & 7) != 0) return SFM_ERR_ARG;  // fp64 partials
  hipStream_t stream = (hipStream_t)stream_;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(p.chunks, p.slabs), dim3(kBlock), 0, stream, x, (const float*)nullptr, 0, GySrc{},
                     (const float*)nullptr, (const float*)nullptr, (long)M, C, p, 0, part);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 3) / 4), dim3(kBlock), 0, stream, part, x, (long)M, C,
                     p.chunks, eps, (float)(1.0 - decay), moving_mean, moving_var, save_mean, save_invstd);
  const long n4 = (long)M * p.groups;
  hipLaunchKernelGGL(bn_elu_apply_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, x, gamma, beta,
                     save_mean, save_invstd, n4, p.groups, act, y, ld_y);
  return sfm_last_error();
}

// slim.batch_norm(x, is_training=False) (+ tf.nn.elu): the batch norms of the inference graph that do not follow a
// convolution (nets.py:672, 460 under test_kitti_depth_con.py's is_training=False); y may be a channel slice (ld_y)
extern "C" int sfm_bn_elu_infer(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                                const float* moving_var, long long M, int C, float eps, int act, float* y, int ld_y,
                                void* stream_) {
  BnPlan p;
  if (!x || !gamma || !beta || !moving_mean || !moving_var || !slice_ok(y, ld_y, C) || !bn_plan((long)M, C, p) || !(eps > 0.0f))
    return SFM_ERR_ARG;
  const long n4 = (long)M * p.groups;
  hipLaunchKernelGGL(bn_elu_infer_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, x,
                     gamma, beta, moving_mean, moving_var, eps, n4, p.groups, act, y, ld_y);
  return sfm_last_error();
}

extern "C" int sfm_bn_elu_fwd(const float* x, const float* gamma, const float* beta, long long M, int C, float eps, double decay,
                              int act, float* moving_mean, float* moving_var, float* y, float* save_mean,
                              float* save_invstd, float* ws, void* stream_) {
  return sfm_bn_elu_fwd_ld(x, gamma, beta, M, C, eps, decay, act, moving_mean, moving_var, y, C, save_mean, save_invstd, ws, stream_);
}

// y a channel slice ([M, ld_y] buffer); dL/dy = gy[0] + ... + gy[n_gy - 1], each a channel slice with its own row stride
extern "C" int sfm_bn_elu_bwd_ld(const float* x, const float* y, int ld_y, const float* const* gy, const int* ld_gy, int n_gy,
                                 const float* gamma, const float* save_mean, const float* save_invstd, long long M, int C,
                                 int act, float* gx, float* ggamma, float* gbeta, float* ws, void* stream_) {
  BnPlan p;
  if (!x || !slice_ok(y, ld_y, C) || !gy || !ld_gy || n_gy < 1 || n_gy > 3 || !gamma || !save_mean || !save_invstd || !gx ||
      !ggamma || !gbeta || !ws || !bn_plan((long)M, C, p))
    return SFM_ERR_ARG;
  if (((uintptr_t)ws & 7) != 0) return SFM_ERR_ARG;
  GySrc src{};
  src.n = n_gy;
  for (int k = 0; k < n_gy; ++k) {
    if (!slice_ok(gy[k], ld_gy[k], C)) return SFM_ERR_ARG;
    src.p[k] = gy[k];
    src.ld[k] = ld_gy[k];
  }
  hipStream_t stream = (hipStream_t)stream_;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(p.chunks, p.slabs), dim3(kBlock), 0, stream, x, y, ld_y, src, save_mean, save_invstd,
                     (long)M, C, p, act, part);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 3) / 4), dim3(kBlock), 0, stream, part, C, p.chunks, ggamma,
                     gbeta);
  const long n4 = (long)M * p.groups;
  hipLaunchKernelGGL(bn_elu_bwd_apply_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, x, y, ld_y, src,
                     gamma, save_mean, save_invstd, ggamma, gbeta, n4, p.groups, (float)(1.0 / (double)M), act, gx);
  return sfm_last_error();
}

extern "C" int sfm_bn_elu_bwd(const float* x, const float* y, const float* gy, const float* gamma, const float* save_mean,
                              const float* save_invstd, long long M, int C, int act, float* gx, float* ggamma,
                              float* gbeta, float* ws, void* stream_) {
  return sfm_bn_elu_bwd_ld(x, y, C, &gy, &C, 1, gamma, save_mean, save_invstd, M, C, act, gx, ggamma, gbeta, ws, stream_);
}

// ---- the same batch norm with statistics over ALL ranks of a data-parallel job (no counterpart in the reference, which is
// single-device: this IS its graph when the global batch is sharded).  The collectives stay with the caller:
//   sfm_bn_local_stats  -> all-gather of stats [3][C] fp64  -> sfm_bn_elu_apply_stats          (forward)
//   sfm_bn_elu_bwd_sums -> all-reduce (sum) of the two [C] vectors -> sfm_bn_elu_bwd_apply     (backward)
// stats = (row count, mean, sum of squared deviations) per channel of this rank's M rows.
extern "C" int sfm_bn_local_stats(const float* x, long long M, int C, double* stats, float* ws, void* stream_) {
  BnPlan p;
  if (!x || !stats || !ws || !bn_plan((long)M, C, p) || ((uintptr_t)ws & 7) != 0) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(p.chunks, p.slabs), dim3(kBlock), 0, stream, x, (const float*)nullptr, 0, GySrc{},
                     (const float*)nullptr, (const float*)nullptr, (long)M, C, p, 0, part);
  hipLaunchKernelGGL(bn_local_stats_kernel, dim3((C + 3) / 4), dim3(kBlock), 0, stream, part, x, (long)M, C, p.chunks, stats);
  return sfm_last_error();
}

// stats_all [world][3][C]: the gathered triples, rank-major.  Writes y and the merged save_mean / save_invstd.
extern "C" int sfm_bn_elu_apply_stats(const float* x, const float* gamma, const float* beta, const double* stats_all, int world,
                                      long long M, int C, float eps, double decay, int act, float* moving_mean,
                                      float* moving_var, float* y, float* save_mean, float* save_invstd, void* stream_) {
  BnPlan p;
  if (!x || !gamma || !beta || !stats_all || world < 1 || !y || !save_mean || !save_invstd || !bn_plan((long)M, C, p) ||
      !(eps > 0.0f))
    return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(bn_merge_stats_kernel, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, stats_all, world, C, eps,
                     (float)(1.0 - decay), moving_mean, moving_var, save_mean, save_invstd);
  const long n4 = (long)M * p.groups;
  hipLaunchKernelGGL(bn_elu_apply_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, x, gamma, beta,
                     save_mean, save_invstd, n4, p.groups, act, y, C);
  return sfm_last_error();
}

// this rank's (sum dz xhat, sum dz) = its share of dL/dgamma, dL/dbeta
extern "C" int sfm_bn_elu_bwd_sums(const float* x, const float* y, const float* gy, const float* save_mean,
                                   const float* save_invstd, long long M, int C, int act, float* ggamma, float* gbeta,
                                   float* ws, void* stream_) {
  BnPlan p;
  if (!x || !y || !gy || !save_mean || !save_invstd || !ggamma || !gbeta || !ws || !bn_plan((long)M, C, p) ||
      ((uintptr_t)ws & 7) != 0)
    return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  double* part = reinterpret_cast<double*>(ws);
  GySrc src{};
  src.n = 1; src.p[0] = gy; src.ld[0] = C;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(p.chunks, p.slabs), dim3(kBlock), 0, stream, x, y, C, src, save_mean, save_invstd,
                     (long)M, C, p, act, part);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 3) / 4), dim3(kBlock), 0, stream, part, C, p.chunks, ggamma, gbeta);
  return sfm_last_error();
}

// gx from the sums over ALL ranks (ggamma_all, gbeta_all: the all-reduced vectors) and the global row count
extern "C" int sfm_bn_elu_bwd_apply(const float* x, const float* y, const float* gy, const float* gamma, const float* save_mean,
                                    const float* save_invstd, const float* ggamma_all, const float* gbeta_all, long long M,
                                    long long M_global, int C, int act, float* gx, void* stream_) {
  BnPlan p;
  if (!x || !y || !gy || !gamma || !save_mean || !save_invstd || !ggamma_all || !gbeta_all || !gx || M_global < M ||
      !bn_plan((long)M, C, p))
    return SFM_ERR_ARG;
  const long n4 = (long)M * p.groups;
  GySrc src{};
  src.n = 1; src.p[0] = gy; src.ld[0] = C;
  hipLaunchKernelGGL(bn_elu_bwd_apply_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream_,
                     x, y, C, src, gamma, save_mean, save_invstd, ggamma_all, gbeta_all, n4, p.groups,
                     (float)(1.0 / (double)M_global), act, gx);
  return sfm_last_error();
}
