// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 (operands in registers, 4 independent
// accumulators per wave, 1 / 2 waves per SIMD) and the clock the chip holds while doing it
// (s_memtime / s_memrealtime).  Design input for the ASPP roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
    a += 1e-7f; b -= 1e-7f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps, iters = 20000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz (memtime %llu / memrealtime %llu x 100 MHz)\n",
           wps, ms, flops / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 0.1, h[0], h[1]);
    hipFree(out); hipFree(clk);
  }
  return 0;
}
