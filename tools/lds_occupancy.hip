// How many workgroups of a given LDS size does the dispatcher keep on one CU at a time?  (gfx950: 160 KB of LDS per CU)
// Each workgroup spins ~40 us and records its start; workgroups that start after the first one ended ran in a later round.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_occupancy.hip -o tools/bin/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(long long* t, int spin_ticks) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    lds[0] = (float)t0;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    t[blockIdx.x * 2] = t0;
    t[blockIdx.x * 2 + 1] = wall_clock64();
  }
  __syncthreads();
}
int main() {
  const int threads_list[] = {256, 384, 512};
  long long* d;
  hipMalloc(&d, 4096 * 2 * sizeof(long long));
  for (int threads : threads_list)
    for (int kb : {16, 32, 40, 48, 52, 56, 60, 64, 72, 80}) {
      const int nwg = 1024;
      hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(spin, dim3(nwg), dim3(threads), kb * 1024, 0, d, 4000);  // 40 us at 100 MHz
        hipDeviceSynchronize();
      }
      std::vector<long long> h(nwg * 2);
      hipMemcpy(h.data(), d, nwg * 2 * sizeof(long long), hipMemcpyDeviceToHost);
      long long t0 = h[0], first_end = h[1];
      for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[2 * i]); first_end = std::min(first_end, h[2 * i + 1]); }
      int first_round = 0;
      for (int i = 0; i < nwg; ++i) first_round += h[2 * i] < first_end;
      printf("threads %d lds %2d KB: %4d of %d workgroups resident at once = %.2f per CU\n", threads, kb, first_round, nwg, first_round / 256.0);
    }
  return 0;
}
