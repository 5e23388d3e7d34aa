// Snippet unpack + augmentation + preprocessing: the device side of data_loader.py.
//
// A training example on disk is ONE jpeg holding the seq = N+1 frames of a snippet side by side
// (data/prepare_train_data.py:33, 47) and one line of 9 comma-separated intrinsics (:49-51).  The reference
// decodes it, slices it into the centre (target) frame and the N source frames stacked along the channels
// (data_loader.py:296-324), batches, augments (fetch_train_batch :126-144 -> data_augmentation :157-280:
// brightness :239-242, per-channel colour :232-238, a uint8 round trip :272-276, a mirror along the width for the
// whole batch :227-231, then tf.image.random_contrast :137), builds the per-scale intrinsics (:348-359) and, in the
// trainer, maps uint8 to [-1, 1] (SfMLearner_uni.py:449-452).  Here the decoded uint8 snippets [B, H, (N+1) W, 3]
// are taken from HBM to the fp32 tensors the loss reads (tgt [B,H,W,3], src [B,H,W,3N]) in two launches:
//   1. per-(item, frame, channel) sums of the uint8-quantised augmented image (the contrast step needs the
//      per-image channel means, AdjustContrastv2), chunk partials in fp64 in a fixed order;
//   2. every workgroup re-adds its item's partials in the same order, then recomputes the quantised value of its
//      pixels, applies contrast, the second uint8 quantisation, the mirror and the [-1, 1] mapping and writes
//      both outputs with 12-byte-per-frame stores.  Block 0 also writes the per-scale intrinsics.
// Byte work bounded by HBM: 3 (N+1) B in (twice), 12 (N+1) B out per pixel.  No fp contraction (Makefile default):
// the two uint8 truncations make every product's rounding visible.
#include "sfm_internal.h"

namespace sfm {
namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4096;   // pixels per workgroup of the sum pass
constexpr int kMaxFrames = 5;  // seq <= 5 (SURVEY config c4)

struct AugParams {
  int augment, swap;
  float bright, col[3], contrast;
};

// tf.image.convert_image_dtype(uint8 -> float32): cast * (1 / 255) in fp32
__device__ __forceinline__ float u8_to_unit(unsigned v) { return (float)v * (1.0f / 255.0f); }
// tf.image.convert_image_dtype(float32 -> uint8, saturate=True): x * 255.5, clamped to [0, 255], truncated
__device__ __forceinline__ unsigned unit_to_u8(float v) {
  const float s = v * 255.5f;
  return (unsigned)fminf(fmaxf(s, 0.0f), 255.0f);
}
// data_augmentation up to the uint8 round trip (data_loader.py:272-276)
__device__ __forceinline__ unsigned aug_quant(unsigned raw, float bright, float col) {
  return unit_to_u8((u8_to_unit(raw) * bright) * col);
}

__device__ __forceinline__ void load_rgb(const unsigned char* __restrict__ p, unsigned& r, unsigned& g, unsigned& b) {
  r = p[0]; g = p[1]; b = p[2];
}

// frame slot in the side-by-side jpeg of output frame k (k = 0: target, k = 1..N: sources in stack order):
// the target is the centre frame N/2, the sources are the frames before it then the frames after it (:298-314)
__device__ __forceinline__ int frame_of(int k, int n_src) {
  const int half = n_src / 2;
  if (k == 0) return half;
  return (k - 1 < half) ? (k - 1) : k;
}

template <int NF>
__global__ __launch_bounds__(kThreads) void snippet_sum_kernel(const unsigned char* __restrict__ seq, int H, int W,
                                                               int chunks, AugParams ap, double* __restrict__ part) {
  constexpr int K = 3 * NF;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  const int HW = H * W;
  const int p0 = chunk * kChunk, p1 = min(p0 + kChunk, HW);
  const unsigned char* __restrict__ base = seq + (size_t)b * HW * K;
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.0;
  for (int p = p0 + (int)threadIdx.x; p < p1; p += kThreads) {
    const int y = p / W, x = p - y * W;
    const unsigned char* row = base + (size_t)y * W * K;
#pragma unroll
    for (int f = 0; f < NF; ++f) {  // summed per jpeg frame slot; pass 2 maps slots to outputs
      unsigned c[3];
      load_rgb(row + ((size_t)f * W + x) * 3, c[0], c[1], c[2]);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) acc[3 * f + ch] += (double)u8_to_unit(aug_quant(c[ch], ap.bright, ap.col[ch]));
    }
  }
  // fixed-order block reduction: lanes by DPP-free shuffles, waves through LDS
  __shared__ double red[kThreads / 64][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
    part[(size_t)blockIdx.x * K + threadIdx.x] = v;
  }
}

template <int NF>
__global__ __launch_bounds__(kThreads) void snippet_unpack_kernel(const unsigned char* __restrict__ seq,
                                                                  const float* __restrict__ Kin, int B, int H, int W,
                                                                  int S, int chunks, AugParams ap,
                                                                  const double* __restrict__ part,
                                                                  float* __restrict__ tgt, float* __restrict__ src,
                                                                  float* __restrict__ Kms) {
  constexpr int K = 3 * NF, N = NF - 1;
  const int HW = H * W;
  const int blocks_per_img = (HW + kThreads - 1) / kThreads;
  const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x - b * blocks_per_img;
  __shared__ float mean[K];
  if (ap.augment && threadIdx.x < K) {  // AdjustContrastv2's per-image channel mean
    double v = 0.0;
    for (int c = 0; c < chunks; ++c) v += part[((size_t)b * chunks + c) * K + threadIdx.x];
    mean[threadIdx.x] = (float)(v / (double)HW);
  }
  if (blockIdx.x == 0) {  // get_multi_scale_intrinsics (data_loader.py:348-359): [B, S, 3, 3]
    for (int i = threadIdx.x; i < B * S; i += kThreads) {
      const int bb = i / S, s = i - bb * S;
      const float div = (float)(1 << s);
      const float* k = Kin + bb * 9;
      float* o = Kms + (size_t)i * 9;
      o[0] = k[0] / div; o[1] = 0.f; o[2] = k[2] / div;
      o[3] = 0.f; o[4] = k[4] / div; o[5] = k[5] / div;
      o[6] = 0.f; o[7] = 0.f; o[8] = 1.f;
    }
  }
  __syncthreads();
  const int p = blk * kThreads + (int)threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const int xs = (ap.augment && ap.swap) ? (W - 1 - x) : x;  // tf.reverse(im, axis=[2]) (:229): out[x] = in[W-1-x]
  const unsigned char* row = seq + ((size_t)b * H + y) * W * K;
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int f = frame_of(k, N);
    unsigned c[3];
    load_rgb(row + ((size_t)f * W + xs) * 3, c[0], c[1], c[2]);
    float o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      unsigned q = c[ch];
      if (ap.augment) {
        q = aug_quant(q, ap.bright, ap.col[ch]);
        const float m = mean[3 * f + ch];
        q = unit_to_u8((u8_to_unit(q) - m) * ap.contrast + m);  // adjust_contrast on uint8 images (:137)
      }
      o[ch] = u8_to_unit(q) * 2.0f - 1.0f;  // preprocess_image, SfMLearner_uni.py:449-452
    }
    float* dst = (k == 0) ? tgt + ((size_t)b * HW + p) * 3 : src + ((size_t)b * HW + p) * (3 * N) + 3 * (k - 1);
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
  }
}

inline int chunks_of(int H, int W) { return (H * W + kChunk - 1) / kChunk; }

template <int NF>
int launch(const unsigned char* seq, const float* K, int B, int H, int W, int S, const AugParams& ap, float* tgt,
           float* src, float* Kms, double* part, hipStream_t st) {
  const int chunks = chunks_of(H, W);
  if (ap.augment) {
    hipLaunchKernelGGL(snippet_sum_kernel<NF>, dim3((unsigned)(B * chunks)), dim3(kThreads), 0, st, seq, H, W, chunks, ap,
                       part);
  }
  const int bpi = (H * W + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(snippet_unpack_kernel<NF>, dim3((unsigned)(B * bpi)), dim3(kThreads), 0, st, seq, K, B, H, W, S,
                     chunks, ap, part, tgt, src, Kms);
  return sfm_last_error();
}

}  // namespace
}  // namespace sfm

extern "C" {

long long sfm_snippet_workspace_floats(int B, int H, int W, int n_src) {
  // even source counts only: unpack_image_sequence takes N/2 frames either side of the centre one (:298-309)
  if (B <= 0 || H <= 0 || W <= 0 || n_src < 2 || (n_src & 1) || n_src + 1 > sfm::kMaxFrames) return -1;
  if (!sfm_fits(H, (long)W * (n_src + 1), 12)) return -1;
  return 2LL * B * sfm::chunks_of(H, W) * 3 * (n_src + 1);  // doubles
}

int sfm_snippet_unpack(const unsigned char* seq, const float* K, int B, int H, int W, int n_src, int S, int augment,
                       float brightness, float col_r, float col_g, float col_b, int swap, float contrast, float* tgt,
                       float* src, float* K_ms, float* ws, void* stream) {
  if (!seq || !K || !tgt || !src || !K_ms || S < 1 || S > 16) return SFM_ERR_ARG;
  if (sfm_snippet_workspace_floats(B, H, W, n_src) < 0) return SFM_ERR_ARG;
  if ((long long)B * ((H * W + sfm::kThreads - 1) / sfm::kThreads) > 2147483647LL) return SFM_ERR_TOO_LARGE;
  if (augment && (!ws || (reinterpret_cast<uintptr_t>(ws) & 7))) return SFM_ERR_ARG;
  sfm::AugParams ap{augment ? 1 : 0, swap ? 1 : 0, brightness, {col_r, col_g, col_b}, contrast};
  double* part = reinterpret_cast<double*>(ws);
  hipStream_t st = (hipStream_t)stream;
  switch (n_src + 1) {
    case 3: return sfm::launch<3>(seq, K, B, H, W, S, ap, tgt, src, K_ms, part, st);
    case 5: return sfm::launch<5>(seq, K, B, H, W, S, ap, tgt, src, K_ms, part, st);
  }
  return SFM_ERR_ARG;
}

}  // extern "C"
