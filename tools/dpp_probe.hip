// Probe: semantics of the DPP wave shifts on gfx950 (lane i reads lane i-1 / i+1, edge lanes read 0).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float lane_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__global__ void k(float* l, float* r) {
  float v = (float)(threadIdx.x + 1);
  l[threadIdx.x] = lane_left(v);
  r[threadIdx.x] = lane_right(v);
}
int main() {
  float *l, *r, hl[128], hr[128];
  hipMalloc(&l, 512); hipMalloc(&r, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, l, r);
  hipMemcpy(hl, l, 512, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) {
    float el = (i % 64 == 0) ? 0.f : (float)i, er = (i % 64 == 63) ? 0.f : (float)(i + 2);
    if (hl[i] != el || hr[i] != er) { if (bad < 8) printf("lane %d: left %g (want %g) right %g (want %g)\n", i, hl[i], el, hr[i], er); ++bad; }
  }
  printf("dpp probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
