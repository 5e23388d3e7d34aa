// Micro-benchmark: LDS float / int atomic throughput on gfx950 (design input for the splat tile kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 8192;  // floats of LDS per block (32 KiB)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  __shared__ float a[N];
  __shared__ int ai[N];
  for (int i = threadIdx.x; i < N; i += 256) { a[i] = 0.f; ai[i] = 0; }
  __syncthreads();
  int idx = (threadIdx.x * stride) & (N - 1);
  float v = 1.0f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) atomicAdd(&a[idx], v);
    if (MODE == 1) atomicAdd(&ai[idx], (int)v);
    if (MODE == 2) { a[idx] += v; }
    if (MODE == 3) atomicAdd(&a[0], v);
    if (MODE == 4) { float o = atomicAdd(&a[idx], v); v += o * 1e-30f; }
    if (MODE == 5) { __hip_atomic_fetch_add(&a[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    if (MODE == 7) { double* p = (double*)&a[(idx * 2) & (N - 2)]; atomicAdd(p, (double)v); }
    if (MODE == 8) { unsigned long long* p = (unsigned long long*)&a[(idx * 2) & (N - 2)]; unsigned long long o = atomicAdd(p, (unsigned long long)threadIdx.x); v += (float)o * 1e-30f; }
    if (MODE == 6) { unsigned long long* p = (unsigned long long*)&a[(idx * 2) & (N - 2)]; atomicAdd(p, (unsigned long long)threadIdx.x); }
    idx = (idx + 256 * stride + 17) & (N - 1);
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < N; i += 256) s += a[i] + ai[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int stride) {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  int iters = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<2048, 256>>>(out, iters, stride);
  hipEventRecord(e0);
  k<MODE><<<2048, 256>>>(out, iters, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_cu = 2048.0 / 256 * 4 * iters;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s stride %d: %.3f ms, %.1f cycles per wave-instruction per CU, %.2f G lane-ops/s\n", name, stride, ms,
         cyc / wave_instr_per_cu, 2048.0 * 256 * iters / (ms * 1e-3) / 1e9);
  hipFree(out);
}
int main() {
  run<0>("ds_add_f32 no-return", 1);
  run<0>("ds_add_f32 no-return", 4);
  run<0>("ds_add_f32 no-return", 33);
  run<1>("ds_add_u32 no-return", 1);
  run<2>("plain LDS rmw (non-atomic)", 1);
  run<3>("ds_add_f32 same address", 1);
  run<4>("ds_add_rtn_f32", 1);
  run<5>("hip_atomic wg-scope f32", 1);
  run<6>("ds_add_u64", 1);
  run<6>("ds_add_u64", 3);
  run<7>("ds_add_f64", 1);
  run<8>("ds_add_rtn_u64", 1);
  return 0;
}
