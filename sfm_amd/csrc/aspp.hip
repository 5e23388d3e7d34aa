// Dilated 3x3 convolutions of the ASPP decoder (nets.py:451-471 `disp_bottleneck_dense`, the
// `slim.conv2d(x, out_channels, [3, 3], rate=r)` calls at :457, :458, :463; called five times from
// `disp_aspp_u_dense`, nets.py:675-686) as fp32-input MFMA implicit GEMMs - the only dense
// contraction of the hot path.  NHWC activations, HWIO weights,
// SAME padding (= `rate` zero pixels on every side), stride 1, fp32 in / fp32 accumulate:
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain, so the result differs from the
// reference's fp32 convolution only by summation order.
//
//   forward          out[p, co]  = sum_{tap, ci} in[p + off(tap), ci] * w[tap, ci, co]         M = B*H*W, N = Cout, K = 9*Cin
//   backward data    gin[p, ci]  = sum_{tap, co} gout[p - off(tap), co] * w[tap, ci, co]       the forward kernel on flipped,
//                                                                                              transposed weights (Cin <-> Cout)
//   backward weight  gw[tap, ci, co] = sum_p in[p + off(tap), ci] * gout[p, co]                M = Cin, N = Cout, K = B*H*W per tap
//
// One tile core for all three: a workgroup owns a BM x BN output tile, waves own WM x WN sub-tiles
// made of 32x32 MFMA tiles; per K-step of 32 the A tile is staged in LDS K-major ([k][m], so the
// MFMA operand read "lane l -> A[m = l & 31][k = l >> 5]" is a conflict-free ds_read_b32) and the B
// tile as [k][n]; the next step's global loads are issued into registers before the MFMAs of the
// current one.  The im2col matrix never exists: a tap is a shifted (zero-padded) view of the input.
// Small-M / small-output problems split K over workgroups (taps for the forward, pixel ranges for
// the weight gradient) into partial slabs that a second kernel sums in a fixed order: results are
// bitwise reproducible, no atomics.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

// acc[i][j] += A[wm0 + 32 i .. +32][0..BK) * B[0..BK)[wn0 + 32 j .. +32]
template <int TM, int TN>
__device__ __forceinline__ void mma_step(const float* __restrict__ As, const float* __restrict__ Bs, int lda, int ldb,
                                         int wm0, int wn0, f32x16 (&acc)[TM][TN]) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = As[(kk + h) * lda + wm0 + i * 32 + r];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + h) * ldb + wn0 + j * 32 + r];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// Optional per-output-channel epilogue of the forward convolution, applied where the finished sum is stored (the MFMA
// kernel's store loop, or the slab reduction when the taps are split): y = act(x * scale[c] + shift[c]) - batch norm in
// INFERENCE mode folded to one FMA (scale = gamma / sqrt(moving_var + eps), shift = beta - moving_mean * scale) and ELU.
// scale == nullptr: none.  (Training-mode batch norm needs the statistics of the whole output first: csrc/bn_elu.hip.)
struct Epilogue {
  const float* scale;
  const float* shift;
  int act;
};
__device__ __forceinline__ float ep_apply(float v, float sc, float sh, int act) {
  const float z = v * sc + sh;
  return (act && z <= 0.0f) ? expf(z) - 1.0f : z;
}

// C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
template <int TM, int TN>
__device__ __forceinline__ void store_acc(const f32x16 (&acc)[TM][TN], float* __restrict__ out, long ld, long row0,
                                          long rows, int col0, const Epilogue ep = Epilogue{nullptr, nullptr, 0}) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + j * 32 + c;
      const float sc = ep.scale ? ep.scale[col] : 1.0f, sh = ep.scale ? ep.shift[col] : 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long row = row0 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (row < rows) out[row * ld + col] = ep.scale ? ep_apply(acc[i][j][reg], sc, sh, ep.act) : acc[i][j][reg];
      }
    }
}

// ------------------------------------------------------------------ forward / backward data
// grid (ceil(M / BM), Cout / BN, splits); a split covers taps [z * tps, (z + 1) * tps) and writes
// its own slab out + z * split_stride.
template <int BM, int BN, int WM, int WN, bool EP>  // EP: apply the epilogue (its own instantiation: the plain kernel stays as it was)
// waves_per_eu(2, 3): without the upper bound hipcc caps the 128x128 configuration at 128 VGPRs (4 waves per
// SIMD) and spills the prefetch registers to scratch inside the K loop
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) __attribute__((amdgpu_waves_per_eu(2, 3))) void aspp_conv_kernel(
    const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ out, int B, int H, int W, int Cin,
    int Cout, int rate, int tps, long split_stride, Epilogue ep) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int LDA = BM + 1, LDB = BN + 4;
  constexpr int A_IT = BM * BK / 4 / NT, B_IT = BK * BN / 4 / NT;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(A_IT >= 1 && B_IT >= 1 && (BM * BK / 4) % NT == 0 && (BK * BN / 4) % NT == 0, "tile / thread mismatch");
  __shared__ float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int wm0 = (wave % (BM / WM)) * WM, wn0 = (wave / (BM / WM)) * WN;
  const long M = (long)B * H * W;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int tap0 = blockIdx.z * tps, tap1 = min(9, tap0 + tps);

  // the A rows this thread stages: row = it * (NT / 8) + tid / 8, channels kc .. kc + 3 of the K-step
  const int kc = (tid & 7) * 4;
  int py[A_IT], px[A_IT];
  long pb[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const long m = m0 + it * (NT / 8) + (tid >> 3);
    if (m < M) {
      const long bi = m / ((long)H * W);
      const int rem = (int)(m - bi * H * W);
      py[it] = rem / W;
      px[it] = rem - py[it] * W;
      pb[it] = bi * H * W;
    } else {
      py[it] = px[it] = 0;
      pb[it] = -1;
    }
  }
  float4 ra[A_IT], rb[B_IT];
  auto load_tiles = [&](int tap, int c0) {
    const int dy = (tap / 3 - 1) * rate, dx = (tap % 3 - 1) * rate;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int y = py[it] + dy, x = px[it] + dx;
      ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pb[it] >= 0 && y >= 0 && y < H && x >= 0 && x < W)
        ra[it] = *reinterpret_cast<const float4*>(in + (pb[it] + (long)y * W + x) * Cin + c0 + kc);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BN / 4), n4 = idx % (BN / 4);
      rb[it] = *reinterpret_cast<const float4*>(wt + ((long)tap * Cin + c0 + k) * Cout + n0 + n4 * 4);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int r = it * (NT / 8) + (tid >> 3);
      As[(kc + 0) * LDA + r] = ra[it].x;
      As[(kc + 1) * LDA + r] = ra[it].y;
      As[(kc + 2) * LDA + r] = ra[it].z;
      As[(kc + 3) * LDA + r] = ra[it].w;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BN / 4), n4 = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Bs[k * LDB + n4 * 4]) = rb[it];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] = 0.f;

  const int csteps = Cin / BK, steps = (tap1 - tap0) * csteps;
  if (steps > 0) load_tiles(tap0, 0);
  for (int s = 0; s < steps; ++s) {
    store_tiles();
    __syncthreads();
    if (s + 1 < steps) load_tiles(tap0 + (s + 1) / csteps, ((s + 1) % csteps) * BK);
    mma_step<TM, TN>(As, Bs, LDA, LDB, wm0, wn0, acc);
    __syncthreads();
  }
  if constexpr (EP)
    store_acc<TM, TN>(acc, out, Cout, m0 + wm0, M, n0 + wn0, ep);  // only launched with gridDim.z == 1
  else
    store_acc<TM, TN>(acc, out + blockIdx.z * split_stride, Cout, m0 + wm0, M, n0 + wn0);
}

// ------------------------------------------------------------------ backward weight
// grid ((Cin / BM) * (Cout / BN), 9, splits); a split covers pixels [z * pps, (z + 1) * pps) and
// writes its own slab gw + z * 9*Cin*Cout.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void aspp_wgrad_kernel(
    const float* __restrict__ in, const float* __restrict__ gout, float* __restrict__ gw, int B, int H, int W, int Cin,
    int Cout, int rate, long pps) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A_IT = BK * BM / 4 / NT, B_IT = BK * BN / 4 / NT;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(A_IT >= 1 && B_IT >= 1 && (BK * BM / 4) % NT == 0 && (BK * BN / 4) % NT == 0, "tile / thread mismatch");
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int wm0 = (wave % (BM / WM)) * WM, wn0 = (wave / (BM / WM)) * WN;
  const int ntn = Cout / BN;
  const int ci0 = (blockIdx.x / ntn) * BM, co0 = (blockIdx.x % ntn) * BN;
  const int tap = blockIdx.y;
  const int dy = (tap / 3 - 1) * rate, dx = (tap % 3 - 1) * rate;
  const long M = (long)B * H * W;
  const long p0 = (long)blockIdx.z * pps, p1 = min(M, p0 + pps);

  // pixel coordinates of the A rows this thread stages, advanced by BK pixels per K-step (no divisions
  // in the loop).  Row k of the step = it * (NT / (BM / 4)) + tid / (BM / 4).
  int ay[A_IT], ax[A_IT];
  long ab[A_IT];  // b * H * W
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const long p = p0 + (it * NT + tid) / (BM / 4);
    const long bi = p / ((long)H * W);
    const int rem = (int)(p - bi * H * W);
    ay[it] = rem / W;
    ax[it] = rem - ay[it] * W;
    ab[it] = bi * H * W;
  }
  float4 ra[A_IT], rb[B_IT];
  auto load_tiles = [&](long pstep) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BM / 4), c4 = idx % (BM / 4);
      ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int y = ay[it] + dy, x = ax[it] + dx;
      if (pstep + k < p1 && y >= 0 && y < H && x >= 0 && x < W)
        ra[it] = *reinterpret_cast<const float4*>(in + (ab[it] + (long)y * W + x) * Cin + ci0 + c4 * 4);
      ax[it] += BK;  // next K-step
      while (ax[it] >= W) { ax[it] -= W; ++ay[it]; }
      while (ay[it] >= H) { ay[it] -= H; ab[it] += (long)H * W; }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BN / 4), n4 = idx % (BN / 4);
      const long p = pstep + k;
      rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < p1) rb[it] = *reinterpret_cast<const float4*>(gout + p * Cout + co0 + n4 * 4);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BM / 4), c4 = idx % (BM / 4);
      *reinterpret_cast<float4*>(&As[k * LDA + c4 * 4]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = it * NT + tid, k = idx / (BN / 4), n4 = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Bs[k * LDB + n4 * 4]) = rb[it];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] = 0.f;

  if (p0 < p1) load_tiles(p0);
  for (long ps = p0; ps < p1; ps += BK) {
    store_tiles();
    __syncthreads();
    if (ps + BK < p1) load_tiles(ps + BK);
    mma_step<TM, TN>(As, Bs, LDA, LDB, wm0, wn0, acc);
    __syncthreads();
  }
  float* o = gw + ((long)blockIdx.z * gridDim.y + tap) * Cin * Cout;  // gridDim.y taps: 9, or 1 for a 1x1 convolution
  store_acc<TM, TN>(acc, o, Cout, ci0 + wm0, Cin, co0 + wn0);
}

// ------------------------------------------------------------------ backward weight, all 9 taps at once
// For layers whose whole Cin x Cout block is small (the last decoder levels: 64 -> 32) one tap per
// workgroup leaves a 64x32 tile with 10 flop/byte - L2-bandwidth bound.  Here a workgroup owns a
// 64 (ci) x 32 (co) block of ALL NINE taps: a K-step is a segment of 32 consecutive pixels of one
// image row; the input patch (3 rows x (32 + 2 rate) pixels x 64 channels) and the 32 x 32 gout
// tile are staged once and every tap reads the patch at its own offset.  3 waves, wave w = tap
// row dy = (w - 1) rate, 3 taps x 2 MFMA tiles each.  36 flop/byte at rate 3.
// grid ((Cin / 64) * (Cout / 32), splits); a split covers row segments [z * sps, (z + 1) * sps).
constexpr int kSegMaxRate = 16;
__global__ __launch_bounds__(192, 2) void aspp_wgrad9_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                          float* __restrict__ gw, int B, int H, int W, int Cin, int Cout,
                                                          int rate, long sps) {
  constexpr int BM = 64, BN = 32, LDA = BM + 4, LDB = BN + 4, NT = 192;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PW = BK + 2 * rate;              // patch width in pixels
  float* As = lds;                           // [3][PW][LDA]
  float* Bs = lds + 3 * PW * LDA;            // [BK][LDB]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int ntn = Cout / BN;
  const int ci0 = (blockIdx.x / ntn) * BM, co0 = (blockIdx.x % ntn) * BN;
  const int nseg = (W + BK - 1) / BK;
  const long nsteps = (long)B * H * nseg;
  const long s0 = (long)blockIdx.y * sps, s1 = min(nsteps, s0 + sps);

  f32x16 acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) acc[j][i][reg] = 0.f;

  // staging: every thread owns up to kAIt float4 of the patch and kBIt of the gout tile; the next step's
  // global loads are issued into registers before the MFMAs of the current one
  constexpr int kAIt = 3 * (BK + 2 * kSegMaxRate) * (BM / 4) / NT, kBIt = (BK * (BN / 4) + NT - 1) / NT;
  const int na4 = 3 * PW * (BM / 4);
  float4 ra[kAIt], rb[kBIt];
  auto load_step = [&](long row, int seg) {   // row = b * H + y
    const int y = (int)(row % H);
    const long rbase = row - y;
    const int xs = seg * BK;
#pragma unroll
    for (int it = 0; it < kAIt; ++it) {
      const int idx = it * NT + tid;
      const int c4 = idx % (BM / 4), pp = idx / (BM / 4), px = pp % PW, pr = pp / PW;
      const int yy = y + (pr - 1) * rate, xx = xs - rate + px;
      ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < na4 && yy >= 0 && yy < H && xx >= 0 && xx < W)
        ra[it] = *reinterpret_cast<const float4*>(in + ((rbase + yy) * W + xx) * Cin + ci0 + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < kBIt; ++it) {
      const int idx = it * NT + tid, n4 = idx % (BN / 4), k = idx / (BN / 4);
      rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < BK && xs + k < W) rb[it] = *reinterpret_cast<const float4*>(gout + (row * W + xs + k) * Cout + co0 + n4 * 4);
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int it = 0; it < kAIt; ++it) {
      const int idx = it * NT + tid;
      if (idx < na4) *reinterpret_cast<float4*>(&As[(idx / (BM / 4)) * LDA + (idx % (BM / 4)) * 4]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < kBIt; ++it) {
      const int idx = it * NT + tid, n4 = idx % (BN / 4), k = idx / (BN / 4);
      if (k < BK) *reinterpret_cast<float4*>(&Bs[k * LDB + n4 * 4]) = rb[it];
    }
  };

  long row = s0 / nseg;              // (image row, segment) of the step being loaded, advanced incrementally
  int seg = (int)(s0 - row * nseg);
  if (s0 < s1) load_step(row, seg);
  for (long s = s0; s < s1; ++s) {
    store_step();
    __syncthreads();
    if (++seg == nseg) { seg = 0; ++row; }
    if (s + 1 < s1) load_step(row, seg);
    const float* Aw = As + wave * PW * LDA;
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 2) {
      const float b = Bs[(kk + h) * LDB + r];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float* ap = Aw + (kk + h + j * rate) * LDA + r;   // input pixel k + (j - 1) rate = patch pixel k + j rate
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], b, acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[32], b, acc[j][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* o = gw + (long)blockIdx.y * 9 * Cin * Cout;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float* ot = o + (long)(wave * 3 + j) * Cin * Cout;
    const int c = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rr = ci0 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        ot[(long)rr * Cout + co0 + c] = acc[j][i][reg];
      }
  }
}

// out[e] = sum_z part[z][e] in a fixed order: a workgroup owns 16 float4 outputs, 16 thread groups
// each add every 16th slab (independent loads in flight), LDS combines the groups in group order.
__device__ __forceinline__ float4 ep_apply4(float4 s, long e, int Cout, const Epilogue ep) {
  if (!ep.scale) return s;
  const int c = (int)((e * 4) % Cout);  // Cout is a multiple of 4: the four lanes of a float4 are channels c .. c + 3
  const float4 sc = *reinterpret_cast<const float4*>(ep.scale + c), sh = *reinterpret_cast<const float4*>(ep.shift + c);
  return make_float4(ep_apply(s.x, sc.x, sh.x, ep.act), ep_apply(s.y, sc.y, sh.y, ep.act), ep_apply(s.z, sc.z, sh.z, ep.act),
                     ep_apply(s.w, sc.w, sh.w, ep.act));
}

__global__ __launch_bounds__(kBlock) void aspp_reduce_kernel(const float* __restrict__ part, long n4, int splits,
                                                             float* __restrict__ out, int Cout, Epilogue ep) {
  __shared__ float4 red[16][16];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + o;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n4)
    for (int z = g; z < splits; z += 16) {
      const float4 v = reinterpret_cast<const float4*>(part)[(long)z * n4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[g][o] = s;
  __syncthreads();
  if (g == 0 && e < n4) {
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][o];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(out)[e] = ep_apply4(s, e, Cout, ep);
  }
}

// the same for a handful of slabs: one thread per float4, slabs in order
__global__ __launch_bounds__(kBlock) void aspp_reduce_few_kernel(const float* __restrict__ part, long n4, int splits,
                                                                 float* __restrict__ out, int Cout, Epilogue ep) {
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n4) return;
  float4 s = reinterpret_cast<const float4*>(part)[e];
  for (int z = 1; z < splits; ++z) {
    const float4 v = reinterpret_cast<const float4*>(part)[(long)z * n4 + e];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out)[e] = ep_apply4(s, e, Cout, ep);
}

static void launch_reduce(const float* part, long n4, int splits, float* out, hipStream_t stream, int Cout = 4,
                          Epilogue ep = Epilogue{nullptr, nullptr, 0}) {
  if (splits <= 16)
    hipLaunchKernelGGL(aspp_reduce_few_kernel, dim3((unsigned)((n4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, part, n4,
                       splits, out, Cout, ep);
  else
    hipLaunchKernelGGL(aspp_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(kBlock), 0, stream, part, n4, splits, out,
                       Cout, ep);
}

// wt[t][co][ci] = w[8 - t][ci][co]: the backward-data problem as a forward convolution
__global__ __launch_bounds__(kBlock) void aspp_flip_kernel(const float* __restrict__ w, int Cin, int Cout,
                                                           float* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((long)((int)gridDim.z - 1 - t) * Cin + ci0 + r) * Cout + co0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) wt[((long)t * Cout + co0 + r) * Cin + ci0 + tx] = tile[tx][r];
}

struct ConvPlan {
  int cfg;      // 0: 128x128, 1: 256x64, 2: 256x32
  int bm, bn;
  int splits;   // over taps
  int tps;      // taps per split
};

static ConvPlan conv_plan(long M, int Cin, int Cout) {
  ConvPlan p;
  if (Cout % 128 == 0) { p.cfg = 0; p.bm = 128; p.bn = 128; }
  else if (Cout % 64 == 0) { p.cfg = 1; p.bm = 256; p.bn = 64; }
  else { p.cfg = 2; p.bm = 256; p.bn = 32; }
  // taps per workgroup.  Model: the fullest CU gets n = ceil(base * splits / 256) workgroups of 9 / splits
  // taps; up to 3 are co-resident and overlap each other's stalls (measured: one workgroup alone on a CU
  // runs at ~0.66 of the rate three reach).  A split also writes and re-reads `splits` M x N slabs (~4 TB/s).
  const long base = ((M + p.bm - 1) / p.bm) * (Cout / p.bn);
  const double t_tap = 2.0 * p.bm * p.bn * (double)Cin / (0.6 * 157.3e12 / 256.0);
  double best = 1e30;
  p.splits = 1;
  for (int sp : {1, 3, 9}) {
    const long n = (base * sp + 255) / 256;
    const double eff = n >= 3 ? 1.0 : (n == 2 ? 0.9 : 0.66);
    const double t = (double)n * (9 / sp) * t_tap / eff + (sp > 1 ? 2.0 * sp * (double)M * Cout * 4.0 / 4.0e12 : 0.0);
    if (t < best * 0.97) { best = t; p.splits = sp; }
  }
  p.tps = 9 / p.splits;
  return p;
}

struct WgradPlan {
  int cfg;  // 0: 128x128, 1: 128x64, 2: 64x32, 3: 32x32, 4: 64x32 x 9 taps on row segments (pps = segments per split)
  int bm, bn;
  int splits;
  long pps;  // pixels per split, multiple of BK
};

static WgradPlan wgrad_plan(int B, int H, int W, int Cin, int Cout, int rate, int ntaps = 9) {
  const long M = (long)B * H * W;
  WgradPlan p;
  if (ntaps == 9 && Cin % 64 == 0 && (Cin % 128 != 0 || Cout % 64 != 0) && rate <= kSegMaxRate && W >= BK) {
    p.cfg = 4; p.bm = 64; p.bn = 32;
    const long nsteps = (long)B * H * ((W + BK - 1) / BK), base = (long)(Cin / 64) * (Cout / 32);
    long splits = (768 + base - 1) / base;               // ~3 workgroups of 3 waves per CU
    if (splits > nsteps / 8) splits = nsteps / 8 > 0 ? nsteps / 8 : 1;
    p.pps = (nsteps + splits - 1) / splits;
    p.splits = (int)((nsteps + p.pps - 1) / p.pps);
    return p;
  }
  if (Cin % 128 == 0 && Cout % 128 == 0) { p.cfg = 0; p.bm = 128; p.bn = 128; }
  else if (Cin % 128 == 0 && Cout % 64 == 0) { p.cfg = 1; p.bm = 128; p.bn = 64; }
  else if (Cin % 64 == 0) { p.cfg = 2; p.bm = 64; p.bn = 32; }
  else { p.cfg = 3; p.bm = 32; p.bn = 32; }
  const long base = (long)(Cin / p.bm) * (Cout / p.bn) * ntaps;
  long splits = (512 + base - 1) / base;
  const long max_splits = (M + 8 * BK - 1) / (8 * BK);  // at least 8 K-steps per split
  if (splits > max_splits) splits = max_splits;
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  p.pps = ((M + splits - 1) / splits + BK - 1) / BK * BK;
  p.splits = (int)((M + p.pps - 1) / p.pps);
  return p;
}

static bool conv_dims_ok(int B, int H, int W, int Cin, int Cout, int rate) {
  return B > 0 && H > 0 && W > 0 && rate > 0 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0 &&
         (long)B * H * W * (Cin > Cout ? Cin : Cout) < (1L << 40);
}

// ntaps = 9: the dilated 3x3; ntaps = 1: a 1x1 convolution = the same kernels on the centre tap alone (rate 0 makes every
// tap offset zero, the weights are [1][Cin][Cout]).
static int launch_conv(const float* in, const float* wt, float* out, float* slabs, int B, int H, int W, int Cin, int Cout,
                       int rate, hipStream_t stream, int ntaps = 9, Epilogue ep = Epilogue{nullptr, nullptr, 0}) {
  const long M = (long)B * H * W;
  ConvPlan p = conv_plan(M, Cin, Cout);
  if (ntaps == 1) { p.splits = 1; p.tps = 1; rate = 0; }
  float* dst = p.splits > 1 ? slabs : out;
  const dim3 grid((unsigned)((M + p.bm - 1) / p.bm), Cout / p.bn, p.splits);
  const long stride = M * Cout;
  const Epilogue kep = p.splits > 1 ? Epilogue{nullptr, nullptr, 0} : ep;  // split taps: the slab reduction applies it
#define SFM_CONV_LAUNCH(...)                                                                                            \
  do {                                                                                                                  \
    if (kep.scale)                                                                                                      \
      hipLaunchKernelGGL((aspp_conv_kernel<__VA_ARGS__, true>), grid, dim3(256), 0, stream, in, wt, dst, B, H, W, Cin, Cout, \
                         rate, p.tps, stride, kep);                                                                     \
    else                                                                                                                \
      hipLaunchKernelGGL((aspp_conv_kernel<__VA_ARGS__, false>), grid, dim3(256), 0, stream, in, wt, dst, B, H, W, Cin,  \
                         Cout, rate, p.tps, stride, kep);                                                               \
  } while (0)
  if (p.cfg == 0)
    SFM_CONV_LAUNCH(128, 128, 64, 64);
  else if (p.cfg == 1)
    SFM_CONV_LAUNCH(256, 64, 64, 64);
  else
    SFM_CONV_LAUNCH(256, 32, 64, 32);
#undef SFM_CONV_LAUNCH
  if (p.splits > 1) {
    const long n4 = stride / 4;
    launch_reduce(slabs, n4, p.splits, out, stream, Cout, ep);
  }
  return sfm_last_error();
}

// scale / shift of inference-mode batch norm, one thread per channel
__global__ void aspp_bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps, int C,
                                    float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = (gamma ? gamma[c] : 1.0f) * (1.0f / sqrtf(var[c] + eps));
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.0f) - mean[c] * sc;
}

}  // namespace

}  // namespace sfm

using namespace sfm;

// Workspace (floats) shared by the three calls below for one layer shape: the larger of the
// split-K slabs of the forward / backward-data problems plus the flipped weights, and the slabs of
// the weight gradient.
extern "C" long long sfm_aspp_conv3x3_workspace_floats(int B, int H, int W, int Cin, int Cout) {
  if (!conv_dims_ok(B, H, W, Cin, Cout, 1)) return SFM_ERR_ARG;
  const long M = (long)B * H * W;
  const ConvPlan f = conv_plan(M, Cin, Cout), d = conv_plan(M, Cout, Cin);
  const WgradPlan g = wgrad_plan(B, H, W, Cin, Cout, 1), g2 = wgrad_plan(B, H, W, Cin, Cout, kSegMaxRate + 1);
  const long fwd = f.splits > 1 ? (long)f.splits * M * Cout : 0;
  const long bwd = (d.splits > 1 ? (long)d.splits * M * Cin : 0) + 9L * Cin * Cout;
  const long wg = (long)(g.splits > g2.splits ? g.splits : g2.splits) * 9 * Cin * Cout;  // either weight-gradient kernel
  long n = fwd > bwd ? fwd : bwd;
  if (wg > n) n = wg;
  n = (n + 3) & ~3L;
  return n + 2L * Cout + 4;  // + the folded scale / shift of sfm_aspp_conv3x3_bn_act_infer, 16-byte aligned
}

// slim.conv2d(in, Cout, [3, 3], rate=rate) without its normaliser / activation (nets.py:457):
// in [B,H,W,Cin], w [3,3,Cin,Cout] (HWIO), out [B,H,W,Cout].
extern "C" int sfm_aspp_conv3x3_fwd(const float* in, const float* w, int B, int H, int W, int Cin, int Cout, int rate,
                                    float* out, float* ws, void* stream) {
  if (!in || !w || !out || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate)) return SFM_ERR_ARG;
  return launch_conv(in, w, out, ws, B, H, W, Cin, Cout, rate, (hipStream_t)stream);
}

// slim.conv2d(in, Cout, [3, 3], rate=rate) under the arg-scope with is_training=False (the inference graphs,
// test_kitti_depth_con.py:19-65): convolution, batch norm on the MOVING statistics, ELU (act != 0) - one pass, the
// normalisation and activation applied where the MFMA kernel (or its slab reduction) stores the finished sums.
extern "C" int sfm_aspp_conv3x3_bn_act_infer(const float* in, const float* w, const float* gamma, const float* beta,
                                             const float* moving_mean, const float* moving_var, float eps, int act, int B,
                                             int H, int W, int Cin, int Cout, int rate, float* out, float* ws,
                                             void* stream_) {
  if (!in || !w || !moving_mean || !moving_var || !out || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate)) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = sfm_aspp_conv3x3_workspace_floats(B, H, W, Cin, Cout);
  float* scale = ws + (total - 4 - 2L * Cout);
  float* shift = scale + Cout;
  hipLaunchKernelGGL(aspp_bn_fold_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, gamma, beta, moving_mean, moving_var,
                     eps, Cout, scale, shift);
  return launch_conv(in, w, out, ws, B, H, W, Cin, Cout, rate, stream, 9, Epilogue{scale, shift, act});
}

// dL/d in [B,H,W,Cin] from gout = dL/d out [B,H,W,Cout] (TF autodiff of the same call).
extern "C" int sfm_aspp_conv3x3_bwd_data(const float* gout, const float* w, int B, int H, int W, int Cin, int Cout,
                                         int rate, float* gin, float* ws, void* stream_) {
  if (!gout || !w || !gin || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate)) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  float* wt = ws;  // [9][Cout][Cin]
  hipLaunchKernelGGL(aspp_flip_kernel, dim3(Cin / 32, Cout / 32, 9), dim3(kBlock), 0, stream, w, Cin, Cout, wt);
  const long wsz = (9L * Cin * Cout + 3) & ~3L;
  return launch_conv(gout, wt, gin, ws + wsz, B, H, W, Cout, Cin, rate, stream);
}

static int launch_wgrad(const float* in, const float* gout, int B, int H, int W, int Cin, int Cout, int rate, float* gw,
                        float* ws, hipStream_t stream, int ntaps) {
  if (ntaps == 1) rate = 0;
  const WgradPlan p = wgrad_plan(B, H, W, Cin, Cout, rate, ntaps);
  float* dst = p.splits > 1 ? ws : gw;
  const dim3 grid((Cin / p.bm) * (Cout / p.bn), ntaps, p.splits);
  if (p.cfg == 4) {
    const size_t lds = (size_t)(3 * (BK + 2 * rate) * 68 + BK * 36) * sizeof(float);
    hipLaunchKernelGGL(aspp_wgrad9_kernel, dim3((Cin / 64) * (Cout / 32), p.splits), dim3(192), lds, stream, in, gout, dst, B, H,
                       W, Cin, Cout, rate, p.pps);
  } else if (p.cfg == 0)
    hipLaunchKernelGGL((aspp_wgrad_kernel<128, 128, 64, 64>), grid, dim3(256), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps);
  else if (p.cfg == 1)
    hipLaunchKernelGGL((aspp_wgrad_kernel<128, 64, 64, 32>), grid, dim3(256), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps);
  else if (p.cfg == 2)
    hipLaunchKernelGGL((aspp_wgrad_kernel<64, 32, 32, 32>), grid, dim3(128), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps);
  else
    hipLaunchKernelGGL((aspp_wgrad_kernel<32, 32, 32, 32>), grid, dim3(64), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps);
  if (p.splits > 1) {
    const long n4 = (long)ntaps * Cin * Cout / 4;
    launch_reduce(ws, n4, p.splits, gw, stream);
  }
  return sfm_last_error();
}

// dL/d w [3,3,Cin,Cout].
extern "C" int sfm_aspp_conv3x3_bwd_weight(const float* in, const float* gout, int B, int H, int W, int Cin, int Cout,
                                           int rate, float* gw, float* ws, void* stream_) {
  if (!in || !gout || !gw || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate)) return SFM_ERR_ARG;
  return launch_wgrad(in, gout, B, H, W, Cin, Cout, rate, gw, ws, (hipStream_t)stream_, 9);
}

// ---- 1x1 convolutions of the same block (nets.py:455, 456, 461, 466, 470): slim.conv2d(x, Cout, [1, 1]) itself, NHWC x
// [1,1,Cin,Cout], as the same implicit-GEMM kernels restricted to the centre tap.  Workspace: the 3x3 one is enough.
extern "C" int sfm_aspp_conv1x1_fwd(const float* in, const float* w, int B, int H, int W, int Cin, int Cout, float* out,
                                    float* ws, void* stream) {
  if (!in || !w || !out || !ws || !conv_dims_ok(B, H, W, Cin, Cout, 1)) return SFM_ERR_ARG;
  return launch_conv(in, w, out, ws, B, H, W, Cin, Cout, 0, (hipStream_t)stream, 1);
}

extern "C" int sfm_aspp_conv1x1_bwd_data(const float* gout, const float* w, int B, int H, int W, int Cin, int Cout,
                                         float* gin, float* ws, void* stream_) {
  if (!gout || !w || !gin || !ws || !conv_dims_ok(B, H, W, Cin, Cout, 1)) return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  float* wt = ws;  // [Cout][Cin]
  hipLaunchKernelGGL(aspp_flip_kernel, dim3(Cin / 32, Cout / 32, 1), dim3(kBlock), 0, stream, w, Cin, Cout, wt);
  const long wsz = ((long)Cin * Cout + 3) & ~3L;
  return launch_conv(gout, wt, gin, ws + wsz, B, H, W, Cout, Cin, 0, stream, 1);
}

extern "C" int sfm_aspp_conv1x1_bwd_weight(const float* in, const float* gout, int B, int H, int W, int Cin, int Cout,
                                           float* gw, float* ws, void* stream) {
  if (!in || !gout || !gw || !ws || !conv_dims_ok(B, H, W, Cin, Cout, 1)) return SFM_ERR_ARG;
  return launch_wgrad(in, gout, B, H, W, Cin, Cout, 0, gw, ws, (hipStream_t)stream, 1);
}
