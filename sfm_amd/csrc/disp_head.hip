// The disparity heads of disp_aspp_u_dense (nets.py:678, 683, 689):
//
//   disp = DISP_SCALING * slim.conv2d(x, 1, [3, 3], activation_fn=tf.sigmoid, normalizer_fn=None) + MIN_DISP
//
// (DISP_SCALING = 10, MIN_DISP = 0.01, nets.py:11-12; no normaliser, so the convolution has a bias), forward and
// backward.  One output channel: per pixel a dot product over the 9 * Cin values of its 3x3 neighbourhood - 18 flop per
// loaded float, nothing for the matrix cores; the bound is the read of x (each element is wanted by 9 outputs, served
// by L1 / L2 after its first fetch from HBM).
//
// Work decomposition (all three kernels): a pixel is handled by L = Cin / 4 adjacent lanes, lane g owning channels
// 4g .. 4g+3 (one float4 of the NHWC pixel: a pixel's taps are contiguous, fully coalesced loads); a wave holds 64 / L
// pixels side by side.  Cin / 4 must be a power of two <= 64.
//   forward          9 float4 loads x 9 float4 weights (registers) -> xor-butterfly over the L lanes -> sigmoid
//   backward, data   gx[q, c] = sum_tap dpre[q - tap] w[tap, c]: 9 scalar loads, one float4 store
//   backward, weight gw[tap, c] = sum_q x[q, c] dpre[q - tap]: every thread keeps 9 float4 sums over its pixels, the
//                    lanes of equal channel group are added by butterfly, waves through LDS, workgroups through a
//                    partials buffer summed by a second kernel in a fixed order (no atomics: reproducible)
// with dpre = gdisp * DISP_SCALING * s (1 - s), s = the sigmoid saved by the forward.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {
namespace {

constexpr float kDispScaling = 10.0f;  // nets.py:11
constexpr float kMinDisp = 0.01f;      // nets.py:12
constexpr int kHeadBlocks = 512;       // workgroups of the weight-gradient kernel (= rows of its partials buffer)

struct HeadDims {
  int B, H, W, Cin, L, shift;  // L = Cin / 4 lanes per pixel = 1 << shift
  long npix;
};

__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
  acc.x = __builtin_fmaf(s, a.x, acc.x); acc.y = __builtin_fmaf(s, a.y, acc.y);
  acc.z = __builtin_fmaf(s, a.z, acc.z); acc.w = __builtin_fmaf(s, a.w, acc.w);
  return acc;
}

__global__ __launch_bounds__(kBlock) void disp_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, HeadDims d,
                                                               float* __restrict__ disp, float* __restrict__ sig) {
  const int g = threadIdx.x & (d.L - 1);
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const float4*>(w + (size_t)t * d.Cin + 4 * g);
  const float b0 = bias[0];
  const long ppb = kBlock >> d.shift;  // pixels per workgroup and pass
  const long per = d.npix;
  // every lane of a wave runs the same number of passes (the butterfly needs all lanes)
  for (long base = (long)blockIdx.x * ppb; base < per; base += (long)gridDim.x * ppb) {
    const long p = base + (threadIdx.x >> d.shift);
    const bool ok = p < per;
    const long pc = ok ? p : per - 1;
    const int u = (int)(pc % d.W), v = (int)((pc / d.W) % d.H);
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int uu = u + dx, vv = v + dy;
      if (uu >= 0 && uu < d.W && vv >= 0 && vv < d.H) {  // SAME: zero padding
        const float4 a = *reinterpret_cast<const float4*>(x + (size_t)(pc + (long)dy * d.W + dx) * d.Cin + 4 * g);
        acc += (a.x * wt[t].x + a.y * wt[t].y) + (a.z * wt[t].z + a.w * wt[t].w);
      }
    }
    for (int m = 1; m < d.L; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if (ok && g == 0) {
      const float s = 1.0f / (1.0f + expf(-(acc + b0)));
      sig[p] = s;
      disp[p] = kDispScaling * s + kMinDisp;
    }
  }
}

__device__ __forceinline__ float dpre_at(const float* __restrict__ sig, const float* __restrict__ gdisp, long p) {
  const float s = sig[p];
  return gdisp[p] * kDispScaling * (s * (1.0f - s));
}

__global__ __launch_bounds__(kBlock) void disp_head_bwd_data_kernel(const float* __restrict__ w, const float* __restrict__ sig,
                                                                    const float* __restrict__ gdisp, HeadDims d,
                                                                    float* __restrict__ gx) {
  const int g = threadIdx.x & (d.L - 1);
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const float4*>(w + (size_t)t * d.Cin + 4 * g);
  const long ppb = kBlock >> d.shift;
  for (long base = (long)blockIdx.x * ppb; base < d.npix; base += (long)gridDim.x * ppb) {
    const long q = base + (threadIdx.x >> d.shift);
    if (q >= d.npix) continue;
    const int u = (int)(q % d.W), v = (int)((q / d.W) % d.H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // x[q] is tap t of the output pixel p = q - offset(t)
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int uu = u - dx, vv = v - dy;
      if (uu >= 0 && uu < d.W && vv >= 0 && vv < d.H) acc = fma4(dpre_at(sig, gdisp, q - (long)dy * d.W - dx), wt[t], acc);
    }
    *reinterpret_cast<float4*>(gx + (size_t)q * d.Cin + 4 * g) = acc;
  }
}

// partials [kHeadBlocks][9 * Cin + 4]: the weight gradient of the workgroup's pixels, then its bias gradient
__global__ __launch_bounds__(kBlock) void disp_head_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ sig,
                                                                      const float* __restrict__ gdisp, HeadDims d,
                                                                      float* __restrict__ partials) {
  __shared__ float red[(kBlock / 64) * 64 * 37];
  const int g = threadIdx.x & (d.L - 1);
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  float gb = 0.0f;
  const long ppb = kBlock >> d.shift;
  for (long base = (long)blockIdx.x * ppb; base < d.npix; base += (long)gridDim.x * ppb) {
    const long q = base + (threadIdx.x >> d.shift);
    if (q >= d.npix) continue;
    const int u = (int)(q % d.W), v = (int)((q / d.W) % d.H);
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)q * d.Cin + 4 * g);
    if (g == 0) gb += dpre_at(sig, gdisp, q);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int uu = u - dx, vv = v - dy;
      if (uu >= 0 && uu < d.W && vv >= 0 && vv < d.H) acc[t] = fma4(dpre_at(sig, gdisp, q - (long)dy * d.W - dx), a, acc[t]);
    }
  }
  // lanes of equal channel group within the wave (they differ in the bits above `shift`)
  for (int m = d.L; m < 64; m <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t].x += __shfl_xor(acc[t].x, m, 64); acc[t].y += __shfl_xor(acc[t].y, m, 64);
      acc[t].z += __shfl_xor(acc[t].z, m, 64); acc[t].w += __shfl_xor(acc[t].w, m, 64);
    }
    gb += __shfl_xor(gb, m, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < d.L) {  // lane == g here
    float* r = red + ((size_t)wave * 64 + lane) * 37;
#pragma unroll
    for (int t = 0; t < 9; ++t) { r[4 * t] = acc[t].x; r[4 * t + 1] = acc[t].y; r[4 * t + 2] = acc[t].z; r[4 * t + 3] = acc[t].w; }
    r[36] = gb;
  }
  __syncthreads();
  float* out = partials + (size_t)blockIdx.x * (9 * d.Cin + 4);
  for (int k = threadIdx.x; k < 9 * d.Cin; k += kBlock) {
    const int t = k / d.Cin, c = k - t * d.Cin, gg = c >> 2, j = c & 3;
    float s = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) s += red[((size_t)wv * 64 + gg) * 37 + 4 * t + j];
    out[k] = s;
  }
  if (threadIdx.x == 0) {
    float s = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) s += red[(size_t)wv * 64 * 37 + 36];
    out[9 * d.Cin] = s;
  }
}

// gw[k] = sum of the workgroups' partials (fp64; fixed order: four interleaved row classes, then those in order);
// k == 9 Cin: the bias gradient.  A workgroup owns 64 adjacent outputs (coalesced rows), its four waves a row class each.
__global__ __launch_bounds__(kBlock) void disp_head_bwd_reduce_kernel(const float* __restrict__ partials, int n_rows, int Cin,
                                                                      float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ double red[kBlock];
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
  const int stride = 9 * Cin + 4;
  const bool ok = k <= 9 * Cin;
  double s = 0.0;
  if (ok) {
#pragma unroll 8
    for (int r = rq; r < n_rows; r += kBlock / 64) s += (double)partials[(size_t)r * stride + k];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rq == 0 && ok) {
    const double t = ((red[threadIdx.x] + red[64 + threadIdx.x]) + red[128 + threadIdx.x]) + red[192 + threadIdx.x];
    if (k < 9 * Cin) gw[k] = (float)t;
    else gb[0] = (float)t;
  }
}

bool head_dims(int B, int H, int W, int Cin, HeadDims& d) {
  if (B < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3)) return false;
  const int L = Cin / 4;
  if (L > 64 || (L & (L - 1))) return false;
  int sh = 0;
  while ((1 << sh) < L) ++sh;
  d = HeadDims{B, H, W, Cin, L, sh, (long)B * H * W};
  return sfm_fits(H, W, Cin) && d.npix * Cin < (1L << 40);
}

unsigned head_grid(const HeadDims& d, long cap) {
  const long ppb = kBlock >> d.shift;
  const long n = (d.npix + ppb - 1) / ppb;
  return (unsigned)(n < cap ? n : cap);
}

}  // namespace
}  // namespace sfm

using namespace sfm;

extern "C" long long sfm_disp_head_workspace_floats(int B, int H, int W, int Cin) {
  HeadDims d;
  if (!head_dims(B, H, W, Cin, d)) return SFM_ERR_ARG;
  return (long long)kHeadBlocks * (9 * Cin + 4);
}

extern "C" int sfm_disp_head_fwd(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, float* disp,
                                 float* sig, void* stream_) {
  HeadDims d;
  if (!x || !w || !bias || !disp || !sig || !head_dims(B, H, W, Cin, d)) return SFM_ERR_ARG;
  hipLaunchKernelGGL(disp_head_fwd_kernel, dim3(head_grid(d, 8192)), dim3(kBlock), 0, (hipStream_t)stream_, x, w, bias, d, disp, sig);
  return sfm_last_error();
}

extern "C" int sfm_disp_head_bwd(const float* x, const float* w, const float* sig, const float* gdisp, int B, int H, int W,
                                 int Cin, float* gx, float* gw, float* gb, float* ws, void* stream_) {
  HeadDims d;
  if (!x || !w || !sig || !gdisp || !head_dims(B, H, W, Cin, d)) return SFM_ERR_ARG;
  if ((gw || gb) && (!gw || !gb || !ws)) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (gx)
    hipLaunchKernelGGL(disp_head_bwd_data_kernel, dim3(head_grid(d, 8192)), dim3(kBlock), 0, stream, w, sig, gdisp, d, gx);
  if (gw) {
    const unsigned nb = head_grid(d, kHeadBlocks);
    hipLaunchKernelGGL(disp_head_bwd_weight_kernel, dim3(nb), dim3(kBlock), 0, stream, x, sig, gdisp, d, ws);
    hipLaunchKernelGGL(disp_head_bwd_reduce_kernel, dim3((9 * Cin + 1 + 63) / 64), dim3(kBlock), 0, stream, ws, (int)nb,
                       Cin, gw, gb);
  }
  return sfm_last_error();
}
