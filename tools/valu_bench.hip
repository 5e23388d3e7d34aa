// Microbench: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_mov_b32_dpp on gfx950 (full chip, 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
  float2v x[8], av = {a, a}, bv = {b, b};
  for (int i = 0; i < 8; ++i) x[i] = float2v{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(av), "v"(bv));
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dpp(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(a));
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_rcp(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i + 1;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
#define DPP_KERNEL(NAME, ASM)                                                                 \
  __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                 \
    float x[8], y[8];                                                                          \
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = i; }                          \
    for (int it = 0; it < ITERS; ++it)                                                         \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(y[i]) : "v"(x[i]), "v"(a)); \
    float s = 0; for (int i = 0; i < 8; ++i) s += y[i];                                        \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                   \
  }
// the DPP source x[i] is never written in the loop: no VALU-write -> DPP-read hazard, pure issue rate
DPP_KERNEL(k_fmac_wave, "v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_fmac_row, "v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_add_row, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_add_wave, "v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DPP_KERNEL(k_mov_wave, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1 ; %2")
DPP_KERNEL(k_fmac_plain, "v_fmac_f32 %0, %1, %2")
__global__ __launch_bounds__(256) void k_cvt64(float* out, float a, float b) {
  float x[8]; double y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = 0; }
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
  double s = 0; for (int i = 0; i < 8; ++i) s += y[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_dmul(float* out, float a, float b) {
  double y[8], da = a, db = b;
  for (int i = 0; i < 8; ++i) y[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(y[i]) : "v"(da), "v"(db));
  double s = 0; for (int i = 0; i < 8; ++i) s += y[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_dfma(float* out, float a, float b) {
  double y[8], da = a, db = b;
  for (int i = 0; i < 8; ++i) y[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y[i]) : "v"(da), "v"(db));
  double s = 0; for (int i = 0; i < 8; ++i) s += y[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
template <typename K> void run(const char* name, K k, int blocks, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double winstr = (double)blocks * 4 * ITERS * 8;  // wave-instructions
  double per_simd = winstr / 1024.0;
  printf("%-10s blocks %5d: %.3f ms, %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, blocks, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  for (int blocks : {256, 512, 2048}) {   // 1, 2, 8 waves per SIMD
    run("v_fma", k_fma, blocks, out); run("v_pk_fma", k_pkfma, blocks, out); run("add_dpp", k_dpp, blocks, out); run("v_rcp", k_rcp, blocks, out);
    run("fmac", k_fmac_plain, blocks, out); run("fmac_wave", k_fmac_wave, blocks, out); run("fmac_row", k_fmac_row, blocks, out);
    run("cvt_f64_f32", k_cvt64, blocks, out); run("mul_f64", k_dmul, blocks, out); run("fma_f64", k_dfma, blocks, out);
    run("add_row", k_add_row, blocks, out); run("add_wave", k_add_wave, blocks, out); run("mov_wave", k_mov_wave, blocks, out);
  }
  return 0;
}
