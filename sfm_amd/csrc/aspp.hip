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
// Forward and backward-data (aspp_conv_sk_kernel) and the weight gradient of the >= 128-channel levels
// (aspp_wgrad_sk_kernel) share one design: a workgroup owns BM x BN output tiles (128x128, 256x64, 256x32), waves own
// WM x WN sub-tiles made of 32x32 MFMA tiles; K advances in steps of 32.  Tiles are staged by LDS-DMA
// (global_load_lds_dwordx4: no staging registers) into two buffers with ONE barrier per K-step, by dedicated LOADER
// waves, while the MULTIPLIER waves only read LDS and issue MFMAs (a DMA issued beside a busy matrix pipe stalls for
// hundreds of cycles, so the two jobs do not share a wave).  The im2col matrix never exists: a tap is a shifted
// (zero-padded) view of the input.  Work is dealt stream-K: the tiles x K-steps units of a layer are cut into ONE
// contiguous range per workgroup, 2 workgroups per CU; a range that ends inside a tile leaves a partial slab and
// aspp_sk_fixup_kernel adds a tile's slabs in workgroup order - bitwise reproducible, no atomics.  The 64 -> 32
// level's weight gradient keeps its own kernel (aspp_wgrad9_kernel: all nine taps of a 64x32 block per workgroup on a
// shared input patch, register staging; one tap per workgroup is L2-bound there at 10 flop/byte), smaller odd shapes
// the plain split-K kernel (aspp_wgrad_kernel).  Measured MFMA-pipe utilisation: profiles/r03_aspp_pmc.txt.
#include "sfm_device.h"
#include "sfm_internal.h"
#include <algorithm>
#include <cstdlib>

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
// kernel's store loop, or the stream-K fix-up when several ranges cut the tile): y = act(x * scale[c] + shift[c]) - batch norm in
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
                                          long rows, int col0, const Epilogue ep = Epilogue{nullptr, nullptr, 0},
                                          int lane = threadIdx.x & 63) {
  const int c = lane & 31, h = lane >> 5;
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
// Staging: global_load_lds_dwordx4 (LDS-DMA, gfx950) writes 16 bytes per lane straight into LDS, two tile buffers
// alternate, and a K-step has ONE barrier - the DMA of step s + 1 is issued before the MFMAs of step s and lands under
// them (no global -> VGPR -> ds_write phase between the MFMA phases, no prefetch registers).  What the DMA dictates: a
// wave instruction fills 64 consecutive 16-byte slots, so the LDS image is decided by WHICH element each lane fetches,
// not by where it stores it.
//   A tile [BM rows][8 slots of 4 channels]: slot(m, g) = m * 8 + (g ^ ((m >> 1) & 7)).  An MFMA operand read is one
//     ds_read_b128 per lane - lane (r, h) takes the four channels of group 2 c + h of row r - and the xor spreads the 16
//     lanes of every b128 service group over all 16 slot columns: conflict-free.  The four values feed four MFMAs, whose
//     k index is therefore 8 c + 4 h + j: lanes 0-31 and 32-63 of one MFMA carry k and k + 4 (any pairing of the K
//     elements is a valid GEMM as long as the B operand pairs them the same way).
//   B tile [BK][BN] row-major, as the weights lie in memory: lane (r, h) reads B[8 c + 4 h + j][n0 + r] (ds_read_b32,
//     consecutive lanes on consecutive words).
// Rows beyond the image / the padding ring fetch a 16-byte block of zeros instead.
//
// Work split: stream-K.  All 11 convolutions of the decoder have the same FLOPs but 52 ... 1664 output tiles; the
// tiles x K-steps units of a problem are dealt in ONE contiguous range per workgroup to 2 workgroups per CU (what the
// 64-80 KB of LDS admit), so every CU runs two workgroups from start to end and all finish together - with whole-tile
// work items the last round ran half empty and cost 120 -> 176 us.  A workgroup whose range ends inside a tile
// writes the partial tile to a slab of its own (at most two per workgroup: head and tail); aspp_sk_fixup_kernel adds
// the slabs of a tile in workgroup order (fixed order: bitwise reproducible, no atomics) and applies the epilogue.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef SFM_TRACE
__device__ long long g_trace[32768];
extern "C" int sfm_debug_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(long long) * 32768); }
#endif

// First unit of (logical) workgroup w: equal ranges.  (Of the two workgroups of a CU the one dispatched first finishes
// the same range ~8 % sooner; handing it 25/24 or 26/24 of the mean made every layer slower, profiles/r02_aspp_kloop.txt.)
__host__ __device__ inline long sk_first_unit(long w, long total, long nwg) { return w * total / nwg; }

// Wave roles.  A DMA instruction that has to be issued while the SIMD's matrix pipe is busy waits for it: beside two
// waves of back-to-back MFMAs one global_load_lds took 600-900 cycles to issue against ~110 beside the same loop with
// the MFMAs replaced by v_fma (cycle stamps, tools/trace_conv.py; profiles/r02_aspp_kloop.txt).  With every wave doing
// both jobs the issue block (8 DMAs, 1100-2400 cycles) sat in front of each wave's MFMAs, the two workgroups of a CU
// drifted into the same phase, and a second workgroup per CU bought nothing (171 us at 256 and at 512 workgroups);
// dealing the DMAs between the MFMAs was slower still.  So a workgroup has NW multiplier waves, which never touch
// global memory inside the K-loop, and NW loader waves that do nothing else (one of each per SIMD and workgroup,
// 4 waves per SIMD with the CU's second workgroup: 128 registers each): the loaders crawl while the multipliers run
// and finish in the gap before the barrier, 171 -> 150 us per convolution.  The loaders' per-step VALU work is cut to
// an add and a select per DMA (byte offsets and tap-validity bits are formed once per tile).
constexpr int conv_threads(int bm, int bn, int wm, int wn) { return (bm / wm) * (bn / wn) * 64 * 2; }

// STAGES = 2: NW multiplier + NW loader waves, two tile buffers, 2 workgroups per CU (4 waves per SIMD).
// STAGES = 3: the 256x128 tile - 8 multiplier + 4 loader waves, three tile buffers (144 KB of LDS), ONE workgroup per CU
//   (3 waves per SIMD).  A third fewer DMA instructions per FLOP than two 128x128 workgroups (48 against 64 per 8192
//   MFMA cycles of a SIMD), and the loaders run two steps ahead: s_waitcnt vmcnt(n) with the newest step's n DMAs still
//   in flight, so a step's landing time is covered by a whole step of MFMAs.
template <int BM, int BN, int WM, int WN, bool EP, int STAGES = 2>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64 + (STAGES == 3 ? 256 : (BM / WM) * (BN / WN) * 64))
__attribute__((amdgpu_waves_per_eu(STAGES == 3 ? 3 : 4, STAGES == 3 ? 3 : 4)))
void aspp_conv_sk_kernel(const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ out,
                         float* __restrict__ partial, int B, int H, int W, int Cin, int Cout, int rate, int ntaps, Epilogue ep,
                         int lda,    // floats between the rows of `in` (Cin, or more when `in` is a channel slice)
                         int ldc) {  // ... and of `out` (Cout, or more when the outputs are a channel slice of a wider buffer)
  constexpr int NW = (BM / WM) * (BN / WN), NL = STAGES == 3 ? 4 : NW;  // multiplier waves, loader waves
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN;
  constexpr int A_IT = A_FLOATS / 256 / NL, B_IT = B_FLOATS / 256 / NL;  // DMA instructions per loader wave and K-step
  static_assert(A_IT >= 1 && B_IT >= 1 && A_FLOATS % (256 * NL) == 0 && B_FLOATS % (256 * NL) == 0, "tile / wave mismatch");
  static_assert(STAGES == 2 || STAGES == 3, "two or three tile buffers");
  // (no wave both issues LDS-DMA and reads LDS, so the compiler's rule - every LDS read waits for every LDS-DMA it
  // cannot prove disjoint - never bites: one array holds all the tile buffers)
  __shared__ __attribute__((aligned(1024))) float lds_all[STAGES * (A_FLOATS + B_FLOATS)];
  float* const lds0 = lds_all;
  float* const lds1 = lds_all + (A_FLOATS + B_FLOATS);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave_all >= NW;
  const int wave = loader ? wave_all - NW : wave_all;  // index within the role
  const int wm0 = (wave % (BM / WM)) * WM, wn0 = (wave / (BM / WM)) * WN;
  const long M = (long)B * H * W;
  const int csteps = Cin / BK, K = ntaps * csteps, ntn = Cout / BN;
  const long total = ((M + BM - 1) / BM) * ntn * K;
  // workgroups are dealt to the 8 XCDs round-robin: each XCD (= each L2) takes one contiguous eighth of the unit range
  const long wg = gridDim.x % 8 == 0 ? (long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (long)blockIdx.x;
  const long u0 = sk_first_unit(wg, total, gridDim.x), u1 = sk_first_unit(wg + 1, total, gridDim.x);

  // B: instruction q of this wave fills floats (wave * B_IT + q) * 256 .. + 256 of the [BK][BN] tile
  int boff[B_IT];
#pragma unroll
  for (int q = 0; q < B_IT; ++q) {
    const int f = (wave * B_IT + q) * 256 + lane * 4;
    boff[q] = ((f / BN) * Cout + f % BN) * 4;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#ifdef SFM_TRACE
#define TR(i) do { if (trace_on && tn < 1020) g_trace[tb + tn++] = clock64(); } while (0)
  const bool trace_on = lane == 0 && (blockIdx.x == 0 || blockIdx.x == 300);
  const int tb = (blockIdx.x == 0 ? 0 : 8192) + wave_all * 1024;
  if (tid == 0 && blockIdx.x < 1024) {
    g_trace[16384 + blockIdx.x * 4 + 0] = wall_clock64();
    g_trace[16384 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492);                 // HW_ID
    g_trace[16384 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
  }
#else
#define TR(i)
#endif

  // the two roles run the tile loop separately: inside one loop the compiler keeps the loop-invariant address tables of
  // BOTH roles live in every wave (spills at the 168 registers that 3 waves per SIMD leave)
  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    for (long u = u0; u < u1;) {
      const long t = u / K;
      const int kb = (int)(u - t * K), ke = (int)((long)K < kb + (u1 - u) ? (long)K : kb + (u1 - u));
      const long tm = t / ntn;
      const long m0 = tm * BM;
      const int n0 = (int)(t - tm * ntn) * BN;
      const int steps = ke - kb;
  #ifdef SFM_TRACE
      int tn = 0;
  #endif
      // A: instruction q of this wave fills rows (wave * A_IT + q) * 8 .. + 8; this lane: row + (lane >> 3), slot lane & 7.
      // Everything a lane needs per K-step is a byte offset and the 9 tap-validity bits of its pixel: the loaders share
      // their SIMD's VALU port with the multipliers' MFMAs and get about one instruction in per MFMA (64 cycles), so
      // the ~15 VALU instructions per DMA of the plain form (coordinates, bounds, 64-bit address) made the loaders the
      // critical path (5-7 K cycles of issue per step against 4.4 K of MFMAs)
      long aoff[A_IT];
      unsigned vmask[A_IT];
      {
        const int row0 = wave * A_IT * 8 + (lane >> 3);
        const unsigned hw = (unsigned)H * (unsigned)W, mm = (unsigned)(m0 + row0);
        const unsigned rem = mm - (mm / hw) * hw;
        int y = (int)(rem / (unsigned)W), x = (int)(rem - (unsigned)y * (unsigned)W);
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
          const int row = row0 + q * 8;
          const int kg4 = 4 * ((lane & 7) ^ ((row >> 1) & 7));
          unsigned mask = 0;
          if (m0 + row < M) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
              const int yy = y + (tp / 3 - 1) * rate, xx = x + (tp % 3 - 1) * rate;
              mask |= (unsigned)(yy >= 0 && yy < H && xx >= 0 && xx < W) << tp;
            }
          }
          vmask[q] = mask;
          aoff[q] = ((m0 + row) * lda + kg4) * 4;
          x += 8;  // the next instruction's row: 8 pixels on
          while (x >= W) { x -= W; ++y; }
          while (y >= H) y -= H;
        }
      }
      auto issue = [&](int g, float* As) __attribute__((always_inline)) {  // K-step g of the tile -> tile buffer As
        const int tap = g / csteps, c0 = (g - tap * csteps) * BK;
        const int dy = (tap / 3 - 1) * rate, dx = (tap % 3 - 1) * rate;
        const char* abase = reinterpret_cast<const char*>(in) + ((long)(dy * W + dx) * lda + c0) * 4;  // uniform
        const char* bbase = reinterpret_cast<const char*>(wt) + (((long)tap * Cin + c0) * Cout + n0) * 4;
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
          const char* gp = ((vmask[q] >> tap) & 1u) ? abase + aoff[q] : reinterpret_cast<const char*>(g_zero16);
          __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(As + (wave * A_IT + q) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_IT; ++q)
          __builtin_amdgcn_global_load_lds((gptr_t)(bbase + boff[q]), (lptr_t)(As + A_FLOATS + (wave * B_IT + q) * 256), 16, 0, 0);
      };
      if constexpr (STAGES == 3) {
        // two steps ahead: when step s is being multiplied, step s + 1 has landed and step s + 2 is in flight
        constexpr int NDMA = A_IT + B_IT;
        issue(kb, lds_all);
        if (steps > 1) {
          issue(kb + 1, lds1);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // step 0 has landed
        int nb = 2;       // buffer of step s + 2
        for (int s = 0; s < steps; ++s) {
          TR(0);
          if (s + 2 < steps) {
            issue(kb + s + 2, lds_all + nb * (A_FLOATS + B_FLOATS));  // (its last readers passed the barrier of step s - 1)
            TR(1);
            TR(2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // step s + 1 has landed
          } else {
            TR(1);
            TR(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          TR(3);
          __syncthreads();
          nb = nb == 2 ? 0 : nb + 1;
        }
        TR(0);
        u += steps;
        continue;
      }
      issue(kb, lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // one barrier per K-step, as in the multiplier loop below; steps taken in pairs so that the buffers are named
      auto land = [&]() __attribute__((always_inline)) {
        TR(1);
        TR(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of the next step has landed ...
        TR(3);
        __syncthreads();  // ... and everybody's; the multipliers are done with the other buffer
      };
      int s = 0;
      for (; s + 1 < steps; s += 2) {
        TR(0);
        issue(kb + s + 1, lds1);
        land();
        TR(0);
        if (s + 2 < steps) issue(kb + s + 2, lds0);
        land();
      }
      if (s < steps) {
        TR(0);
        land();
      }
      TR(0);
      u += steps;
    }
    return;
  }
  for (long u = u0; u < u1;) {
    const long t = u / K;
    const int kb = (int)(u - t * K), ke = (int)((long)K < kb + (u1 - u) ? (long)K : kb + (u1 - u));
    const long tm = t / ntn;
    const long m0 = tm * BM;
    const int n0 = (int)(t - tm * ntn) * BN;
    const int steps = ke - kb;
#ifdef SFM_TRACE
    int tn = 0;
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] = 0.f;

    // one K-step on a landed buffer: the operands of k-group c + 1 are requested before the 4 x TM x TN MFMAs of group
    // c (sched_barrier pins the order: the scheduler otherwise sinks every ds_read to just in front of its first use)
    struct Frag {
      float4 a[TM];
      float b[4][TN];
    };
    auto fetch = [&](const float* As, int c, Frag& f) __attribute__((always_inline)) {
      const float* Bs = As + A_FLOATS;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = wm0 + i * 32 + r;
        f.a[i] = *reinterpret_cast<const float4*>(As + (m * 8 + ((2 * c + h) ^ ((m >> 1) & 7))) * 4);
      }
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j4][j] = Bs[(8 * c + 4 * h + j4) * BN + wn0 + j * 32 + r];
    };
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float a = j4 == 0 ? f.a[i].x : (j4 == 1 ? f.a[i].y : (j4 == 2 ? f.a[i].z : f.a[i].w));
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.b[j4][j], acc[i][j], 0, 0, 0);
        }
    };
    auto compute = [&](const float* As) __attribute__((always_inline)) {
      Frag f0, f1;
      fetch(As, 0, f0);
#pragma unroll
      for (int c = 0; c < BK / 8; c += 2) {
        fetch(As, c + 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < BK / 8) fetch(As, c + 2, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    __syncthreads();
    auto step = [&](const float* As) __attribute__((always_inline)) {
      TR(0);
      TR(1);
      compute(As);
      TR(2);
      TR(3);
      __syncthreads();  // the loaders' DMA of the next step has landed; everybody is done reading this buffer
    };
    if constexpr (STAGES == 3) {
      int cb = 0;
      for (int s = 0; s < steps; ++s) {
        step(lds_all + cb * (A_FLOATS + B_FLOATS));
        cb = cb == 2 ? 0 : cb + 1;
      }
    } else {
      int s = 0;
      for (; s + 1 < steps; s += 2) {
        step(lds0);
        step(lds1);
      }
      if (s < steps) step(lds0);
    }
    TR(0);
    // (opaque copies: the compiler otherwise forms the 64 row addresses of the store in front of the K-loop and
    // spills them across it)
    int lane_o = lane, wm0_o = wm0;
    asm volatile("" : "+v"(lane_o), "+s"(wm0_o));
    if (kb == 0 && ke == K) {  // the whole tile: straight to the output
      if constexpr (EP) store_acc<TM, TN>(acc, out, ldc, m0 + wm0_o, M, n0 + wn0, ep, lane_o);
      else store_acc<TM, TN>(acc, out, ldc, m0 + wm0_o, M, n0 + wn0, Epilogue{nullptr, nullptr, 0}, lane_o);
    } else {  // part of a tile: this workgroup's head (slab 0: the range starts inside the tile) or tail (slab 1)
      float* slab = partial + (wg * 2 + (u == u0 ? 0 : 1)) * (BM * BN);
      store_acc<TM, TN>(acc, slab, BN, wm0_o, BM, wn0, Epilogue{nullptr, nullptr, 0}, lane_o);
    }
    u += steps;
  }
#ifdef SFM_TRACE
  if (tid == 0 && blockIdx.x < 1024) g_trace[16384 + blockIdx.x * 4 + 1] = wall_clock64();
#endif
}

// out tile = sum of the slabs of the workgroups whose ranges cut the tile, in a fixed order (bitwise reproducible).
// WGRAD: a "tile" is a (ci tile, co tile, tap) block of the weight gradient, K its pixel steps, rows = Cin, taps = ntaps.
// WIDE = false: grid (tiles, BM * BN / 4 / 256), a thread owns a float4 and adds the slabs in workgroup order - the
//   convolutions, where a tile is cut by a handful of ranges.
// WIDE = true: grid (tiles, BM * BN / 4 / 16), 16 float4 per block, 16 thread groups each add every 16th slab and LDS
//   combines the groups in group order - weight gradients of few tiles, where hundreds of ranges cut one tile (a 1x1
//   convolution 128 -> 64 is ONE tile cut by all 512 workgroups: the serial form took 105 us for it, longer than the
//   MFMA kernel).
template <int BM, int BN, bool WGRAD, bool WIDE>
__global__ __launch_bounds__(kBlock) void aspp_sk_fixup_kernel(const float* __restrict__ partial, float* __restrict__ out, long M,
                                                               int Cout, int K, long nwg, long total, int taps, Epilogue ep) {
  __shared__ float4 red[WIDE ? 16 : 1][16];
  const long t = blockIdx.x;
  const int ntn = Cout / BN;
  const long ta = t * K, tb = ta + K;
  long w = ta * nwg / total;  // a workgroup at or before the owner of unit ta
  while (w > 0 && sk_first_unit(w, total, nwg) > ta) --w;
  while (sk_first_unit(w + 1, total, nwg) <= ta) ++w;
  {
    const long a0 = sk_first_unit(w, total, nwg), a1 = sk_first_unit(w + 1, total, nwg);
    if (a0 <= ta && a1 >= tb) return;  // one workgroup did the whole tile and stored it itself
  }
  const int grp = WIDE ? threadIdx.x >> 4 : 0, ngrp = WIDE ? 16 : 1;
  const int e = WIDE ? blockIdx.y * 16 + (threadIdx.x & 15) : blockIdx.y * kBlock + threadIdx.x;  // float4 of the tile
  const int row = e / (BN / 4), c4 = (e % (BN / 4)) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (w += grp; w < nwg; w += ngrp) {
    const long a0 = sk_first_unit(w, total, nwg);
    if (a0 >= tb) break;
    const long seg0 = a0 > ta ? a0 : ta;
    const float4 v = *reinterpret_cast<const float4*>(partial + ((w * 2 + (seg0 == a0 ? 0 : 1)) * (long)(BM * BN)) + (long)row * BN + c4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if constexpr (WIDE) {
    red[grp][threadIdx.x & 15] = s;
    __syncthreads();
    if (grp != 0) return;
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][threadIdx.x & 15];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  long orow;  // row of the [rows][Cout] output matrix
  int n0;
  if constexpr (WGRAD) {
    const long tile = t / taps;
    const int tap = (int)(t - tile * taps);
    orow = (long)tap * M + (tile / ntn) * BM + row;  // M = Cin here
    n0 = (int)(tile % ntn) * BN;
  } else {
    const long tm = t / ntn;
    orow = tm * BM + row;
    n0 = (int)(t - tm * ntn) * BN;
    if (orow >= M) return;
  }
  if (ep.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(ep.scale + n0 + c4), sh = *reinterpret_cast<const float4*>(ep.shift + n0 + c4);
    s = make_float4(ep_apply(s.x, sc.x, sh.x, ep.act), ep_apply(s.y, sc.y, sh.y, ep.act), ep_apply(s.z, sc.z, sh.z, ep.act),
                    ep_apply(s.w, sc.w, sh.w, ep.act));
  }
  *reinterpret_cast<float4*>(out + orow * Cout + n0 + c4) = s;
}

// The same sums for the convolutions, launched over the BOUNDARIES between consecutive workgroups' ranges instead of over
// the tiles: only a boundary that falls inside a tile cuts one (at most nwg - 1 tiles of up to thousands), and the first
// boundary inside a tile handles it.  grid (nwg - 1, BM * BN / 4 / 256).  (The tile-indexed form spent 13-25 us per
// convolution mostly on workgroups that found their tile uncut after a round of 64-bit divisions.)
template <int BM, int BN>
__global__ __launch_bounds__(kBlock) void aspp_sk_fixup_cut_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                   long M, int Cout, int K, long nwg, long total, Epilogue ep,
                                                                   int ldc) {
  const long b = blockIdx.x;  // the boundary between workgroups b and b + 1
  const long u = sk_first_unit(b + 1, total, nwg);
  const long t = u / K, ta = t * K, tb = ta + K;
  if (u == ta || u >= total) return;               // a tile edge: nothing is cut here
  if (sk_first_unit(b, total, nwg) > ta) return;   // workgroup b starts inside the same tile: an earlier boundary handles it
  const int e = blockIdx.y * kBlock + threadIdx.x;  // float4 of the tile
  const int row = e / (BN / 4), c4 = (e % (BN / 4)) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long w = b; w < nwg; ++w) {  // workgroup b holds the tile's first piece; pieces in workgroup order
    const long a0 = sk_first_unit(w, total, nwg);
    if (a0 >= tb) break;
    const long seg0 = a0 > ta ? a0 : ta;
    const float4 v = *reinterpret_cast<const float4*>(partial + ((w * 2 + (seg0 == a0 ? 0 : 1)) * (long)(BM * BN)) + (long)row * BN + c4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int ntn = Cout / BN;
  const long tm = t / ntn, orow = tm * BM + row;
  const int n0 = (int)(t - tm * ntn) * BN;
  if (orow >= M) return;
  if (ep.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(ep.scale + n0 + c4), sh = *reinterpret_cast<const float4*>(ep.shift + n0 + c4);
    s = make_float4(ep_apply(s.x, sc.x, sh.x, ep.act), ep_apply(s.y, sc.y, sh.y, ep.act), ep_apply(s.z, sc.z, sh.z, ep.act),
                    ep_apply(s.w, sc.w, sh.w, ep.act));
  }
  *reinterpret_cast<float4*>(out + orow * ldc + n0 + c4) = s;
}

// ------------------------------------------------------------------ backward weight, stream-K + LDS-DMA + loader waves
// The same machinery for dL/dw[tap][ci][co] = sum_p in[p + tap offset][ci] * gout[p][co]: M = Cin, N = Cout, K = the
// B*H*W pixels in steps of 32.  Both operands lie K-major in memory ([pixel][channel]), so both tiles are dense
// [BK][BM] / [BK][BN] LDS images filled by 1 KB DMA pieces and read with ds_read_b32 (lane (r, h): row k + h, word r).
// A unit = one K-step of one (ci tile, co tile, tap) block; 512 workgroups take equal contiguous ranges (the old grid
// of blocks x taps x pixel splits left 416-576 workgroups for 512-1024 slots: CUs with 2 and CUs with 3).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(conv_threads(BM, BN, WM, WN)) __attribute__((amdgpu_waves_per_eu(4, 4)))
void aspp_wgrad_sk_kernel(const float* __restrict__ in, const float* __restrict__ gout, float* __restrict__ gw,
                          float* __restrict__ partial, int B, int H, int W, int Cin, int Cout, int rate, int ntaps, int lda) {
  constexpr int NW = (BM / WM) * (BN / WN), NL = NW;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * BN;
  constexpr int A_IT = A_FLOATS / 256 / NL, B_IT = B_FLOATS / 256 / NL;
  static_assert(A_IT >= 1 && B_IT >= 1 && A_FLOATS % (256 * NL) == 0 && B_FLOATS % (256 * NL) == 0, "tile / wave mismatch");
  __shared__ __attribute__((aligned(1024))) float lds0[A_FLOATS + B_FLOATS];
  __shared__ __attribute__((aligned(1024))) float lds1[A_FLOATS + B_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave_all >= NW;
  const int wave = loader ? wave_all - NW : wave_all;
  const int wm0 = (wave % (BM / WM)) * WM, wn0 = (wave / (BM / WM)) * WN;
  const long M = (long)B * H * W;
  const int K = (int)((M + BK - 1) / BK), ntn = Cout / BN;
  const long total = (long)(Cin / BM) * ntn * ntaps * K;
  const long wg = gridDim.x % 8 == 0 ? (long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (long)blockIdx.x;
  const long u0 = sk_first_unit(wg, total, gridDim.x), u1 = sk_first_unit(wg + 1, total, gridDim.x);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    // lane constants: row k and byte offset of its 16 bytes inside a K-step's A / B block
    int ak[A_IT], aoff[A_IT], bk_[B_IT], boff[B_IT];
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      const int f = (wave * A_IT + q) * 256 + lane * 4;
      ak[q] = f / BM;
      aoff[q] = (ak[q] * lda + f % BM) * 4;
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
      const int f = (wave * B_IT + q) * 256 + lane * 4;
      bk_[q] = f / BN;
      boff[q] = (bk_[q] * Cout + f % BN) * 4;
    }
    for (long u = u0; u < u1;) {
      const long t = u / K;
      const int kb = (int)(u - t * K), ke = (int)((long)K < kb + (u1 - u) ? (long)K : kb + (u1 - u));
      const long tile = t / ntaps;
      const int tap = (int)(t - tile * ntaps);
      const int ci0 = (int)(tile / ntn) * BM, co0 = (int)(tile % ntn) * BN;
      const int dy = (tap / 3 - 1) * rate, dx = (tap % 3 - 1) * rate;
      const int steps = ke - kb;
      // image coordinates of this lane's pixel of each A instruction, advanced by BK pixels per step
      int ay[A_IT], ax[A_IT];
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        const unsigned hw = (unsigned)H * (unsigned)W, pp = (unsigned)((long)kb * BK + ak[q]);
        const unsigned rem = pp % hw;
        ay[q] = (int)(rem / (unsigned)W);
        ax[q] = (int)(rem - (unsigned)ay[q] * (unsigned)W);
      }
      const char* abase0 = reinterpret_cast<const char*>(in) + ((long)(dy * W + dx) * lda + ci0) * 4;
      const char* bbase0 = reinterpret_cast<const char*>(gout) + (long)co0 * 4;
      auto issue = [&](int g, float* As) __attribute__((always_inline)) {  // K-step g -> tile buffer As
        const char* abase = abase0 + (long)g * BK * lda * 4;
        const char* bbase = bbase0 + (long)g * BK * Cout * 4;
        const long left = M - (long)g * BK;  // pixels from the step's first one to the end
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
          const int y = ay[q] + dy, x = ax[q] + dx;
          const bool ok = ak[q] < left && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
          const char* gp = ok ? abase + aoff[q] : reinterpret_cast<const char*>(g_zero16);
          __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(As + (wave * A_IT + q) * 256), 16, 0, 0);
          ax[q] += BK;
          while (ax[q] >= W) { ax[q] -= W; ++ay[q]; }
          while (ay[q] >= H) ay[q] -= H;
        }
#pragma unroll
        for (int q = 0; q < B_IT; ++q) {
          const char* gp = bk_[q] < left ? bbase + boff[q] : reinterpret_cast<const char*>(g_zero16);
          __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(As + A_FLOATS + (wave * B_IT + q) * 256), 16, 0, 0);
        }
      };
      auto land = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      };
      issue(kb, lds0);
      land();
      int s = 0;
      for (; s + 1 < steps; s += 2) {
        issue(kb + s + 1, lds1);
        land();
        if (s + 2 < steps) issue(kb + s + 2, lds0);
        land();
      }
      if (s < steps) land();
      u += steps;
    }
    return;
  }
  for (long u = u0; u < u1;) {
    const long t = u / K;
    const int kb = (int)(u - t * K), ke = (int)((long)K < kb + (u1 - u) ? (long)K : kb + (u1 - u));
    const int steps = ke - kb;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] = 0.f;
    __syncthreads();
    int s = 0;
    for (; s + 1 < steps; s += 2) {
      mma_step<TM, TN>(lds0, lds0 + A_FLOATS, BM, BN, wm0, wn0, acc);
      __syncthreads();
      mma_step<TM, TN>(lds1, lds1 + A_FLOATS, BM, BN, wm0, wn0, acc);
      __syncthreads();
    }
    if (s < steps) {
      mma_step<TM, TN>(lds0, lds0 + A_FLOATS, BM, BN, wm0, wn0, acc);
      __syncthreads();
    }
    int lane_o = lane, wm0_o = wm0;
    asm volatile("" : "+v"(lane_o), "+s"(wm0_o));
    if (kb == 0 && ke == K) {
      const long tile = t / ntaps;
      const int tap = (int)(t - tile * ntaps);
      const int ci0 = (int)(tile / ntn) * BM, co0 = (int)(tile % ntn) * BN;
      store_acc<TM, TN>(acc, gw + (long)tap * Cin * Cout, Cout, ci0 + wm0_o, Cin, co0 + wn0, Epilogue{nullptr, nullptr, 0}, lane_o);
    } else {
      float* slab = partial + (wg * 2 + (u == u0 ? 0 : 1)) * (BM * BN);
      store_acc<TM, TN>(acc, slab, BN, wm0_o, BM, wn0, Epilogue{nullptr, nullptr, 0}, lane_o);
    }
    u += steps;
  }
}

// ------------------------------------------------------------------ backward weight
// grid ((Cin / BM) * (Cout / BN), 9, splits); a split covers pixels [z * pps, (z + 1) * pps) and
// writes its own slab gw + z * 9*Cin*Cout.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void aspp_wgrad_kernel(
    const float* __restrict__ in, const float* __restrict__ gout, float* __restrict__ gw, int B, int H, int W, int Cin,
    int Cout, int rate, long pps, int lda) {
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
        ra[it] = *reinterpret_cast<const float4*>(in + (ab[it] + (long)y * W + x) * lda + ci0 + c4 * 4);
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
__global__ __launch_bounds__(384, 2) void aspp_wgrad9_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                          float* __restrict__ gw, int B, int H, int W, int Cin, int Cout,
                                                          int rate, long sps, int lda) {
  constexpr int BM = 64, BN = 32, LDA = BM + 4, LDB = BN + 4, NT = 384;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PW = BK + 2 * rate;              // patch width in pixels
  float* As = lds;                           // [3][PW][LDA]
  float* Bs = lds + 3 * PW * LDA;            // [BK][LDB]
  // six waves: wave = tap row + 3 * (half of the 64 input channels) - one 32x32 MFMA tile per tap and wave.  (Three
  // waves with both halves each held 96 accumulator + 64 staging registers: two waves per SIMD, 6 waves over 4 SIMDs.)
  const int tid = threadIdx.x, wave6 = tid >> 6, wave = wave6 % 3, half = wave6 / 3, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int ntn = Cout / BN;
  const int ci0 = (blockIdx.x / ntn) * BM, co0 = (blockIdx.x % ntn) * BN;
  const int nseg = (W + BK - 1) / BK;
  const long nsteps = (long)B * H * nseg;
  const long s0 = (long)blockIdx.y * sps, s1 = min(nsteps, s0 + sps);

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[j][reg] = 0.f;

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
        ra[it] = *reinterpret_cast<const float4*>(in + ((rbase + yy) * W + xx) * lda + ci0 + c4 * 4);
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
        const float* ap = Aw + (kk + h + j * rate) * LDA + 32 * half + r;   // input pixel k + (j - 1) rate = patch pixel k + j rate
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], b, acc[j], 0, 0, 0);
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
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = ci0 + half * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      ot[(long)rr * Cout + co0 + c] = acc[j][reg];
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
  int cfg;      // 0: 128x128, 1: 256x64, 2: 256x32, 3: 256x128 with three tile buffers
  int bm, bn;
  long nwg;     // workgroups of the stream-K launch
};

constexpr int kSkResident = 512;  // 2 workgroups per CU x 256 CUs
static ConvPlan conv_plan(long M, int Cin, int Cout, int ntaps) {
  ConvPlan p;
  // SFM_ASPP_TILE256=1 selects the 256x128 three-buffer configuration (measured 5-12 % SLOWER than two 128x128 workgroups
  // per CU on every layer, profiles/r02_aspp_kloop.txt: kept as an experiment, off by default)
  static const bool wide3 = [] { const char* e = getenv("SFM_ASPP_TILE256"); return e && e[0] == '1'; }();
  long resident_default = kSkResident;
  if (Cout % 128 == 0 && wide3) { p.cfg = 3; p.bm = 256; p.bn = 128; resident_default = kSkResident / 2; }  // one workgroup per CU
  else if (Cout % 128 == 0) { p.cfg = 0; p.bm = 128; p.bn = 128; }
  else if (Cout % 64 == 0) { p.cfg = 1; p.bm = 256; p.bn = 64; }
  else { p.cfg = 2; p.bm = 256; p.bn = 32; }
  const long total = ((M + p.bm - 1) / p.bm) * (Cout / p.bn) * ntaps * (Cin / BK);
  // one resident set of workgroups with equal ranges; small problems: at least 4 K-steps per workgroup
  static const long forced = [] { const char* e = getenv("SFM_ASPP_NWG"); return e ? atol(e) : 0L; }();  // experiments
  const long resident = forced > 0 ? forced : resident_default;
  p.nwg = total / 4 < resident ? (total / 4 > 0 ? total / 4 : 1) : resident;
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
    long splits = (512 + base - 1) / base;               // 2 workgroups of 6 waves per CU: one resident round (768: 240 us, 512: 225)
    if (splits > nsteps / 8) splits = nsteps / 8 > 0 ? nsteps / 8 : 1;
    p.pps = (nsteps + splits - 1) / splits;
    p.splits = (int)((nsteps + p.pps - 1) / p.pps);
    return p;
  }
  if (Cin % 128 == 0 && Cout % 64 == 0) {  // stream-K kernels: splits = 0, pps = workgroups
    const bool wide = Cout % 128 == 0;
    p.cfg = wide ? 0 : 1; p.bm = 128; p.bn = wide ? 128 : 64;
    const long total = (long)(Cin / p.bm) * (Cout / p.bn) * ntaps * ((M + BK - 1) / BK);
    p.splits = 0;
    p.pps = total / 4 < kSkResident ? (total / 4 > 0 ? total / 4 : 1) : kSkResident;
    return p;
  }
  if (Cin % 64 == 0) { p.cfg = 2; p.bm = 64; p.bn = 32; }
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
         (long)B * H * W < (1L << 31) && (long)B * H * W * (Cin > Cout ? Cin : Cout) < (1L << 40);
}

// ntaps = 9: the dilated 3x3; ntaps = 1: a 1x1 convolution = the same kernels on the centre tap alone (rate 0 makes every
// tap offset zero, the weights are [1][Cin][Cout]).
static int launch_conv(const float* in, const float* wt, float* out, float* slabs, int B, int H, int W, int Cin, int Cout,
                       int rate, hipStream_t stream, int ntaps = 9, Epilogue ep = Epilogue{nullptr, nullptr, 0}, int lda = 0,
                       int ldc = 0) {
  const long M = (long)B * H * W;
  if (lda == 0) lda = Cin;
  if (ldc == 0) ldc = Cout;
  const ConvPlan p = conv_plan(M, Cin, Cout, ntaps);
  if (ntaps == 1) rate = 0;
  const int K = ntaps * (Cin / BK);
  const long tiles = ((M + p.bm - 1) / p.bm) * (Cout / p.bn);
#define SFM_CONV_LAUNCH(BM_, BN_, WM_, WN_, ST_, NT_)                                                                    \
  do {                                                                                                                  \
    if (ep.scale)                                                                                                       \
      hipLaunchKernelGGL((aspp_conv_sk_kernel<BM_, BN_, WM_, WN_, true, ST_>), dim3((unsigned)p.nwg), dim3(NT_), 0, stream, in, wt, \
                         out, slabs, B, H, W, Cin, Cout, rate, ntaps, ep, lda, ldc);                                    \
    else                                                                                                                \
      hipLaunchKernelGGL((aspp_conv_sk_kernel<BM_, BN_, WM_, WN_, false, ST_>), dim3((unsigned)p.nwg), dim3(NT_), 0, stream, in,    \
                         wt, out, slabs, B, H, W, Cin, Cout, rate, ntaps, ep, lda, ldc);                                \
    if (p.nwg > 1)                                                                                                      \
      hipLaunchKernelGGL((aspp_sk_fixup_cut_kernel<BM_, BN_>), dim3((unsigned)(p.nwg - 1), BM_ * BN_ / 4 / kBlock), dim3(kBlock), \
                         0, stream, slabs, out, M, Cout, K, p.nwg, tiles * K, ep, ldc);                                 \
  } while (0)
  if (p.cfg == 3)
    SFM_CONV_LAUNCH(256, 128, 64, 64, 3, 768);
  else if (p.cfg == 0)
    SFM_CONV_LAUNCH(128, 128, 64, 64, 2, 512);
  else if (p.cfg == 1)
    SFM_CONV_LAUNCH(256, 64, 64, 64, 2, 512);
  else
    SFM_CONV_LAUNCH(256, 32, 64, 32, 2, 512);
#undef SFM_CONV_LAUNCH
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
  const ConvPlan f = conv_plan(M, Cin, Cout, 9), d = conv_plan(M, Cout, Cin, 9);
  const WgradPlan g = wgrad_plan(B, H, W, Cin, Cout, 1), g2 = wgrad_plan(B, H, W, Cin, Cout, kSegMaxRate + 1);
  // stream-K slabs: two partial tiles per workgroup (the 1x1 convolutions launch at most as many workgroups)
  const long fwd = 2L * f.nwg * f.bm * f.bn;
  const long bwd = 2L * d.nwg * d.bm * d.bn + 9L * Cin * Cout;
  long wg = (long)(g.splits > g2.splits ? g.splits : g2.splits) * 9 * Cin * Cout;  // either split weight-gradient kernel
  if (g.splits == 0) wg = 2L * g.pps * g.bm * g.bn;                                 // stream-K slabs
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

// The same with `in` a CHANNEL SLICE of a wider NHWC buffer: in points at the slice's first channel of pixel 0, ld_in =
// floats between consecutive pixels (the buffer's channel count).  What the concat-free decoder level reads:
// tf.concat([supnet, aspp_up], 3) (nets.py:458) is the first 2C channels of the level's one concat buffer.
static bool slice_ok(const float* in, int ld_in, int Cin, long M) {
  return ld_in >= Cin && ld_in % 4 == 0 && ((uintptr_t)in & 15) == 0 && M * ld_in < (1L << 40);
}
extern "C" int sfm_aspp_conv3x3_fwd_ld(const float* in, int ld_in, const float* w, int B, int H, int W, int Cin, int Cout,
                                       int rate, float* out, float* ws, void* stream) {
  if (!in || !w || !out || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate) || !slice_ok(in, ld_in, Cin, (long)B * H * W))
    return SFM_ERR_ARG;
  return launch_conv(in, w, out, ws, B, H, W, Cin, Cout, rate, (hipStream_t)stream, 9, Epilogue{nullptr, nullptr, 0}, ld_in);
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

// The inference decoder level without a concat (the serving path: test_kitti_depth_con.py:19-65 builds the net with
// is_training=False): the same one-pass convolution + batch norm (moving statistics) + ELU with `in` a channel slice of a
// wider buffer (ld_in) and `out` written as a channel slice of the level's concat buffer (ld_out) - one slim.conv2d = one
// kernel pass, the concat of nets.py:458, 463, 467, 469 never materialised.  ntaps = 9 (3x3, `rate`) or 1 (1x1).
extern "C" int sfm_aspp_conv_bn_act_infer_ld(const float* in, int ld_in, const float* w, int ntaps, const float* gamma,
                                             const float* beta, const float* moving_mean, const float* moving_var, float eps,
                                             int act, int B, int H, int W, int Cin, int Cout, int rate, float* out, int ld_out,
                                             float* ws, void* stream_) {
  if ((ntaps != 1 && ntaps != 9) || !in || !w || !moving_mean || !moving_var || !out || !ws ||
      !conv_dims_ok(B, H, W, Cin, Cout, ntaps == 9 ? rate : 1) || !slice_ok(in, ld_in, Cin, (long)B * H * W) ||
      !slice_ok(out, ld_out, Cout, (long)B * H * W))
    return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = sfm_aspp_conv3x3_workspace_floats(B, H, W, Cin, Cout);
  float* scale = ws + (total - 4 - 2L * Cout);
  float* shift = scale + Cout;
  hipLaunchKernelGGL(aspp_bn_fold_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, gamma, beta, moving_mean, moving_var,
                     eps, Cout, scale, shift);
  return launch_conv(in, w, out, ws, B, H, W, Cin, Cout, rate, stream, ntaps, Epilogue{scale, shift, act}, ld_in, ld_out);
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
                        float* ws, hipStream_t stream, int ntaps, int lda = 0) {
  if (ntaps == 1) rate = 0;
  if (lda == 0) lda = Cin;
  const WgradPlan p = wgrad_plan(B, H, W, Cin, Cout, rate, ntaps);
  float* dst = p.splits > 1 ? ws : gw;  // (stream-K plans: splits = 0, the slabs live in ws)
  const dim3 grid((Cin / p.bm) * (Cout / p.bn), ntaps, p.splits);
  if (p.cfg == 4) {
    const size_t lds = (size_t)(3 * (BK + 2 * rate) * 68 + BK * 36) * sizeof(float);
    hipLaunchKernelGGL(aspp_wgrad9_kernel, dim3((Cin / 64) * (Cout / 32), p.splits), dim3(384), lds, stream, in, gout, dst, B, H,
                       W, Cin, Cout, rate, p.pps, lda);
  } else if (p.cfg == 0 || p.cfg == 1) {
    const long M = (long)B * H * W;
    const int K = (int)((M + BK - 1) / BK);
    const long tiles = (long)(Cin / p.bm) * (Cout / p.bn) * ntaps;
    const Epilogue none{nullptr, nullptr, 0};
    const bool wide = p.pps > 8 * tiles;  // more than ~8 ranges cut a tile
#define SFM_WGRAD_LAUNCH(BM_, BN_, WM_, WN_)                                                                                  \
  do {                                                                                                                        \
    hipLaunchKernelGGL((aspp_wgrad_sk_kernel<BM_, BN_, WM_, WN_>), dim3((unsigned)p.pps), dim3(conv_threads(BM_, BN_, WM_, WN_)), \
                       0, stream, in, gout, gw, ws, B, H, W, Cin, Cout, rate, ntaps, lda);                                    \
    if (wide)                                                                                                                 \
      hipLaunchKernelGGL((aspp_sk_fixup_kernel<BM_, BN_, true, true>), dim3((unsigned)tiles, BM_ * BN_ / 4 / 16), dim3(kBlock),  \
                         0, stream, ws, gw, (long)Cin, Cout, K, p.pps, tiles * K, ntaps, none);                               \
    else                                                                                                                      \
      hipLaunchKernelGGL((aspp_sk_fixup_kernel<BM_, BN_, true, false>), dim3((unsigned)tiles, BM_ * BN_ / 4 / kBlock),           \
                         dim3(kBlock), 0, stream, ws, gw, (long)Cin, Cout, K, p.pps, tiles * K, ntaps, none);                 \
  } while (0)
    if (p.cfg == 0)
      SFM_WGRAD_LAUNCH(128, 128, 64, 64);
    else
      SFM_WGRAD_LAUNCH(128, 64, 64, 32);
#undef SFM_WGRAD_LAUNCH
    return sfm_last_error();
  } else if (p.cfg == 2)
    hipLaunchKernelGGL((aspp_wgrad_kernel<64, 32, 32, 32>), grid, dim3(128), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps, lda);
  else
    hipLaunchKernelGGL((aspp_wgrad_kernel<32, 32, 32, 32>), grid, dim3(64), 0, stream, in, gout, dst, B, H, W, Cin, Cout,
                       rate, p.pps, lda);
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

// dL/d w with `in` a channel slice (see sfm_aspp_conv3x3_fwd_ld)
extern "C" int sfm_aspp_conv3x3_bwd_weight_ld(const float* in, int ld_in, const float* gout, int B, int H, int W, int Cin,
                                              int Cout, int rate, float* gw, float* ws, void* stream_) {
  if (!in || !gout || !gw || !ws || !conv_dims_ok(B, H, W, Cin, Cout, rate) || !slice_ok(in, ld_in, Cin, (long)B * H * W))
    return SFM_ERR_ARG;
  return launch_wgrad(in, gout, B, H, W, Cin, Cout, rate, gw, ws, (hipStream_t)stream_, 9, ld_in);
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
