// Host-side helpers private to the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#define SFM_OK 0
#define SFM_ERR_ARG (-1)        // null pointer / non-positive size / unsupported shape
#define SFM_ERR_TOO_LARGE (-2)  // an image does not fit 32-bit per-image indexing

// Launch status as the C ABI reports it: 0, or the hipError_t of the failed launch.
static inline int sfm_last_error() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFM_OK : (int)e;
}

// linspace step of the TF-1 LinSpaceOp in fp32: (1 - (-1)) / (n - 1).
static inline float sfm_lin_step(int n) { return n > 1 ? 2.0f / (float)(n - 1) : 0.0f; }

// every per-image element count must stay below 2^31 (kernels index an image with int)
static inline bool sfm_fits(long h, long w, long per_pixel) { return h * w * per_pixel < 2147483647L; }

// pose.hip: sfm_pose_grad plus an optional list of words to clear (used by the fused loss)
int sfm_launch_pose_grad(const float* pose, long pose_sb, long pose_si, const float* K, const float* partials, int n_sets,
                         int nblk, int B, int n_src, int inverse_pose, float gscale, float* d_pose, long dpose_sb,
                         long dpose_si, int* zero, int n_zero, hipStream_t stream);

// loss.hip: the tile-owned LDS splat for one image set (the stand-alone projective_warp forward)
namespace sfm {
long splat_solo_table_int4s(int B, int H, int W);
bool launch_splat_solo(const float* img, const float* depth, const float* trans, const float* params, int B, int H, int W,
                       float* acc, void* table, hipStream_t stream);
}
