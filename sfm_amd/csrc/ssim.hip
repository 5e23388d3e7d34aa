// tf_ssim as a stand-alone operator (utils.py:548-575, window 504-519): the raw SSIM
// map `value` (utils.py:572) for 3-channel NHWC images with the 3x3, sigma 1.5
// Gaussian window and SAME zero padding, plus its backward with respect to both
// images.  The clip((1-v)/2, 0, 1) and the optional global mean of lines 573-575
// are elementwise/reduction glue left to the caller (Python/torch on the device).
//
// Backward: v at pixel q depends on the five windowed raw moments
// (m1, m2, m11, m22, m12).  With u[q] = dL/dv[q]:
//   dL/dx[p] = sum_q w(p-q) u[q] (a_x[q] + 2 b_x[q] x[p] + c[q] y[p])
//   dL/dy[p] = sum_q w(p-q) u[q] (a_y[q] + 2 b_y[q] y[p] + c[q] x[p])
// Stage A writes the five u-scaled coefficient maps, stage B convolves them.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {

__device__ __forceinline__ void moments_at(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                           int v, int u, const SsimWin& win, float* m1, float* m2, float* m11,
                                           float* m22, float* m12) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) m1[ch] = m2[ch] = m11[ch] = m22[ch] = m12[ch] = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = v + dy, xx = u + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // SAME: zero padding
      const float w = win.w[(dy + 1) * 3 + (dx + 1)];
      f3 a = ld3(x + (yy * W + xx) * 3), b = ld3(y + (yy * W + xx) * 3);
      const float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        m1[ch] += w * av[ch];
        m2[ch] += w * bv[ch];
        m11[ch] += w * (av[ch] * av[ch]);
        m22[ch] += w * (bv[ch] * bv[ch]);
        m12[ch] += w * (av[ch] * bv[ch]);
      }
    }
}

__global__ __launch_bounds__(kBlock) void ssim_fwd_kernel(const float* __restrict__ img1,
                                                          const float* __restrict__ img2, int H, int W,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int v = p / W, u = p - v * W;
  const long base = (long)b * H * W;
  const SsimWin win = ssim_window();
  float m1[3], m2[3], m11[3], m22[3], m12[3];
  moments_at(img1 + base * 3, img2 + base * 3, H, W, v, u, win, m1, m2, m11, m22, m12);
  st3(out + (base + p) * 3, ssim_from_moments(m1[0], m2[0], m11[0], m22[0], m12[0]),
      ssim_from_moments(m1[1], m2[1], m11[1], m22[1], m12[1]),
      ssim_from_moments(m1[2], m2[2], m11[2], m22[2], m12[2]));
}

// coef [B,H,W,3,5] = u * (a_x, b_x, c, a_y, b_y)
__global__ __launch_bounds__(kBlock) void ssim_coef_kernel(const float* __restrict__ img1,
                                                           const float* __restrict__ img2,
                                                           const float* __restrict__ gv, int H, int W,
                                                           float* __restrict__ coef) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int v = p / W, u = p - v * W;
  const long base = (long)b * H * W;
  const SsimWin win = ssim_window();
  float m1[3], m2[3], m11[3], m22[3], m12[3];
  moments_at(img1 + base * 3, img2 + base * 3, H, W, v, u, win, m1, m2, m11, m22, m12);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float ax, bx, cx, ay, by, cy;
    ssim_with_grad(m1[ch], m2[ch], m11[ch], m22[ch], m12[ch], ax, bx, cx);
    ssim_with_grad(m2[ch], m1[ch], m22[ch], m11[ch], m12[ch], ay, by, cy);
    const float g = gv[(base + p) * 3 + ch];
    float* o = coef + ((base + p) * 3 + ch) * 5;
    o[0] = g * ax; o[1] = g * bx; o[2] = g * cx; o[3] = g * ay; o[4] = g * by;
  }
}

__global__ __launch_bounds__(kBlock) void ssim_bwd_kernel(const float* __restrict__ img1,
                                                          const float* __restrict__ img2,
                                                          const float* __restrict__ coef, int H, int W,
                                                          float* __restrict__ d1, float* __restrict__ d2) {
  const int b = blockIdx.y, p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int v = p / W, u = p - v * W;
  const long base = (long)b * H * W;
  const SsimWin win = ssim_window();
  const float* cb = coef + base * 15;
  float A[3][5];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < 5; ++k) A[ch][k] = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = v + dy, xx = u + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const float w = win.w[(dy + 1) * 3 + (dx + 1)];
      const float* c = cb + (long)(yy * W + xx) * 15;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int k = 0; k < 5; ++k) A[ch][k] += w * c[ch * 5 + k];
    }
  f3 x = ld3(img1 + (base + p) * 3), y = ld3(img2 + (base + p) * 3);
  const float xv[3] = {x.x, x.y, x.z}, yv[3] = {y.x, y.y, y.z};
  float o1[3], o2[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    o1[ch] = A[ch][0] + 2.0f * A[ch][1] * xv[ch] + A[ch][2] * yv[ch];
    o2[ch] = A[ch][3] + 2.0f * A[ch][4] * yv[ch] + A[ch][2] * xv[ch];
  }
  if (d1) st3(d1 + (base + p) * 3, o1[0], o1[1], o1[2]);
  if (d2) st3(d2 + (base + p) * 3, o2[0], o2[1], o2[2]);
}

}  // namespace sfm

using namespace sfm;

extern "C" int sfm_ssim_fwd(const float* img1, const float* img2, int B, int H, int W, float* out, void* stream) {
  if (!img1 || !img2 || !out || B <= 0 || H <= 0 || W <= 0 || B > 65535 || !sfm_fits(H, W, 15)) return SFM_ERR_ARG;
  dim3 grid((H * W + kBlock - 1) / kBlock, B);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, img1, img2, H, W, out);
  return sfm_last_error();
}

// coef: caller-provided workspace of B*H*W*15 floats.
extern "C" int sfm_ssim_bwd(const float* img1, const float* img2, const float* gv, int B, int H, int W, float* coef,
                            float* d1, float* d2, void* stream) {
  if (!img1 || !img2 || !gv || !coef || (!d1 && !d2) || B <= 0 || H <= 0 || W <= 0 || B > 65535 ||
      !sfm_fits(H, W, 15))
    return SFM_ERR_ARG;
  dim3 grid((H * W + kBlock - 1) / kBlock, B);
  hipLaunchKernelGGL(ssim_coef_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, img1, img2, gv, H, W, coef);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, img1, img2, coef, H, W, d1, d2);
  return sfm_last_error();
}
