// Per-(source, batch item) set-up and its adjoint.
//   pose_setup_kernel : pose [6] + K [3x3]  ->  Kinv [9] and P = [[K,0],[0,1]] * T  [3x4]
//   pose_grad_kernel  : block partial sums of dL/dP  ->  dL/dpose [6]
// Reference: euler2mat utils.py:27-75, pose_vec2mat utils.py:77-96, tf.matrix_inverse
// utils.py:112/189/224, K4 * T utils.py:195-201 == 230-236.
//
// The forward follows the canonical fp32 op order of oracle/ref_graph.py exactly
// (k-ordered non-fused dot products, partial-pivot LU inverse, correctly rounded
// sin/cos via fp64) because the projected integer indices must be bit-exact.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {

// Partial-pivot LU inverse, N x N, fp32, same op order as matrix_inverse_lu().
template <int N, typename T>
__device__ void lu_inverse(const T* __restrict__ a, T* __restrict__ inv) {
  T lu[N][N], rhs[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      lu[i][j] = a[i * N + j];
      rhs[i][j] = (i == j) ? T(1) : T(0);
    }
  for (int k = 0; k < N; ++k) {
    int p = k;
    T best = lu[k][k] < T(0) ? -lu[k][k] : lu[k][k];
    for (int i = k + 1; i < N; ++i) {
      T v = lu[i][k] < T(0) ? -lu[i][k] : lu[i][k];
      if (v > best) { best = v; p = i; }
    }
    if (p != k)
      for (int j = 0; j < N; ++j) {
        T t = lu[k][j]; lu[k][j] = lu[p][j]; lu[p][j] = t;
        T u = rhs[k][j]; rhs[k][j] = rhs[p][j]; rhs[p][j] = u;
      }
    for (int i = k + 1; i < N; ++i) {
      T l = lu[i][k] / lu[k][k];
      lu[i][k] = l;
      for (int j = k + 1; j < N; ++j) lu[i][j] = lu[i][j] - l * lu[k][j];
    }
  }
  T y[N][N];
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < N; ++c) {
      T acc = rhs[i][c];
      for (int j = 0; j < i; ++j) acc = acc - lu[i][j] * y[j][c];
      y[i][c] = acc;
    }
  for (int i = N - 1; i >= 0; --i)
    for (int c = 0; c < N; ++c) {
      T acc = y[i][c];
      for (int j = i + 1; j < N; ++j) acc = acc - lu[i][j] * inv[j * N + c];
      inv[i * N + c] = acc / lu[i][i];
    }
}

template <int N, typename T>
__device__ void matmul_seq(const T* a, const T* b, T* o) {  // k-ordered, no FMA
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      T acc = a[i * N + 0] * b[0 * N + j];
      for (int k = 1; k < N; ++k) acc = acc + a[i * N + k] * b[k * N + j];
      o[i * N + j] = acc;
    }
}

constexpr float kPiF = 3.14159274101257324f;  // fp32(np.pi)

// rotation factors from the six correctly rounded sines / cosines of the clipped angles
__device__ void euler_mats(float sx, float cx, float sy, float cy, float sz, float cz, float* X, float* Y, float* Z) {
  float z_[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
  float y_[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
  float x_[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
  for (int k = 0; k < 9; ++k) { X[k] = x_[k]; Y[k] = y_[k]; Z[k] = z_[k]; }
}

// items = n_src * B, item j = i * B + b.  pose element (b, i, k) at pose[b*pose_sb + i*pose_si + k].
// Four lanes per item: lanes 0..2 of a group each evaluate the fp64 sincos of one angle (the kernel is one wave's
// latency chain in front of every step; the three evaluations were 2/3 of it when one thread walked them), lane 0
// collects them by shuffles and does the rest.
__global__ __launch_bounds__(64) void pose_setup_kernel(const float* __restrict__ pose, long pose_sb, long pose_si,
                                                        const float* __restrict__ K, int B, int n_src, int inverse_pose,
                                                        float* __restrict__ params) {
  const int items = B * n_src;
  const int jr = blockIdx.x * 16 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const int j = jr < items ? jr : items - 1;  // idle groups redo the last item: every lane takes part in the shuffles
  int i = j / B, b = j - i * B;
  const float* pv = pose + b * pose_sb + i * pose_si;
  const float* k3 = K + (long)b * 9;
  const float ang = fminf(fmaxf(pv[3 + (sub < 3 ? sub : 0)], -kPiF), kPiF);  // sub 0, 1, 2: rx, ry, rz
  double sd, cd;
  sincos((double)ang, &sd, &cd);
  const float sf = (float)sd, cf = (float)cd;
  const int base = threadIdx.x & ~3;
  const float sx = __shfl(sf, base, 64), cx = __shfl(cf, base, 64);
  const float sy = __shfl(sf, base + 1, 64), cy = __shfl(cf, base + 1, 64);
  const float sz = __shfl(sf, base + 2, 64), cz = __shfl(cf, base + 2, 64);
  if (sub != 0 || jr >= items) return;
  float X[9], Y[9], Z[9], XY[9], R[9];
  euler_mats(sx, cx, sy, cy, sz, cz, X, Y, Z);
  matmul_seq<3>(X, Y, XY);
  matmul_seq<3>(XY, Z, R);
  float T[16] = {R[0], R[1], R[2], pv[0], R[3], R[4], R[5], pv[1], R[6], R[7], R[8], pv[2], 0.f, 0.f, 0.f, 1.f};
  float Tm[16];
  if (inverse_pose) {
    lu_inverse<4>(T, Tm);
  } else {
    for (int k = 0; k < 16; ++k) Tm[k] = T[k];
  }
  float K4[16] = {k3[0], k3[1], k3[2], 0.f, k3[3], k3[4], k3[5], 0.f, k3[6], k3[7], k3[8], 0.f, 0.f, 0.f, 0.f, 1.f};
  float P[16];
  matmul_seq<4>(K4, Tm, P);
  float Ki[9];
  lu_inverse<3>(k3, Ki);
  float* o = params + (long)j * SFM_PARAM_STRIDE;
  for (int k = 0; k < 9; ++k) o[k] = Ki[k];
  for (int k = 0; k < 12; ++k) o[9 + k] = P[k];
  for (int k = 21; k < SFM_PARAM_STRIDE; ++k) o[k] = 0.f;
}

// dL/dP block partials [n_sets][items][nblk][12]  ->  d_pose (b, i, 0..5) at d_pose[b*sb + i*si + k].
// One 256-thread workgroup per item.  The n_sets slabs (the scales of the fused loss) and the
// nblk block partials are summed in fp64 in a fixed order (strided per thread, then a fixed
// shuffle / LDS tree), so the result is bitwise reproducible; thread 0 applies the chain rule.
__global__ __launch_bounds__(kBlock) void pose_grad_kernel(const float* __restrict__ pose, long pose_sb, long pose_si,
                                                           const float* __restrict__ K,
                                                           const float* __restrict__ partials, int n_sets, int nblk,
                                                           int B, int n_src, int inverse_pose, float gscale,
                                                           float* __restrict__ d_pose, long dpose_sb, long dpose_si,
                                                           int* __restrict__ zero, int n_zero) {
  __shared__ double red[4][12];
  if (zero)  // optional: words the caller wants cleared by the last kernel of its step
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n_zero; k += gridDim.x * kBlock) zero[k] = 0;
  const int j = blockIdx.x;
  const int i = j / B, b = j - i * B;
  const float* pv = pose + b * pose_sb + i * pose_si;
  const float* k3 = K + (long)b * 9;
  double dP[12];
  for (int k = 0; k < 12; ++k) dP[k] = 0.0;
  const long items = (long)B * n_src;
  const int rows = n_sets * nblk;
  for (int r = threadIdx.x; r < rows; r += kBlock) {
    const int s = r / nblk, q = r - s * nblk;
    const float* pp = partials + ((((long)s * items + j) * nblk) + q) * 12;
    const float4 a = *reinterpret_cast<const float4*>(pp), bq = *reinterpret_cast<const float4*>(pp + 4),
                 c = *reinterpret_cast<const float4*>(pp + 8);
    dP[0] += (double)a.x; dP[1] += (double)a.y; dP[2] += (double)a.z; dP[3] += (double)a.w;
    dP[4] += (double)bq.x; dP[5] += (double)bq.y; dP[6] += (double)bq.z; dP[7] += (double)bq.w;
    dP[8] += (double)c.x; dP[9] += (double)c.y; dP[10] += (double)c.z; dP[11] += (double)c.w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    double v = dP[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 12; ++k) dP[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  // dTm[r][c] = sum_k K[k][r] dP[k][c]   (rows 0..2; row 3 of K4 only feeds q3, unused)
  double dTm[16];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      dTm[r * 4 + c] = (double)k3[0 * 3 + r] * dP[0 * 4 + c] + (double)k3[1 * 3 + r] * dP[1 * 4 + c] +
                       (double)k3[2 * 3 + r] * dP[2 * 4 + c];
  for (int c = 0; c < 4; ++c) dTm[12 + c] = 0.0;
  // rebuild the rotation in fp64 from the clipped fp32 angles
  double rx = fmin(fmax((double)pv[3], -(double)kPiF), (double)kPiF);
  double ry = fmin(fmax((double)pv[4], -(double)kPiF), (double)kPiF);
  double rz = fmin(fmax((double)pv[5], -(double)kPiF), (double)kPiF);
  // fp32 sin / cos (1 ulp) widened: the forward's own values are fp32, and fp64 sincos is ~4 us of serial
  // single-thread latency at the end of every step
  float sxf, cxf, syf, cyf, szf, czf;
  sincosf((float)rx, &sxf, &cxf);
  sincosf((float)ry, &syf, &cyf);
  sincosf((float)rz, &szf, &czf);
  const double sx = sxf, cx = cxf, sy = syf, cy = cyf, sz = szf, cz = czf;
  double X[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  double Y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  double Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double XY[9], R[9], YZ[9];
  matmul_seq<3>(X, Y, XY);
  matmul_seq<3>(XY, Z, R);
  matmul_seq<3>(Y, Z, YZ);
  double dT[16];
  if (inverse_pose) {
    // Tm = inv(T):  dT = -Tm^T * dTm * Tm^T
    double T[16] = {R[0], R[1], R[2], pv[0], R[3], R[4], R[5], pv[1], R[6], R[7], R[8], pv[2], 0, 0, 0, 1};
    double Tm[16], tmp[16];
    lu_inverse<4>(T, Tm);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += Tm[k * 4 + r] * dTm[k * 4 + c];
        tmp[r * 4 + c] = acc;
      }
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += tmp[r * 4 + k] * Tm[c * 4 + k];
        dT[r * 4 + c] = -acc;
      }
  } else {
    for (int k = 0; k < 16; ++k) dT[k] = dTm[k];
  }
  double dR[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) dR[r * 3 + c] = dT[r * 4 + c];
  // R = X * (Y*Z):  dX = dR * (YZ)^T ;  dYZ = X^T * dR ;  dY = dYZ * Z^T ; dZ = Y^T * dYZ
  double dX[9], dYZ[9], dY[9], dZ[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double a0 = 0, a1 = 0;
      for (int k = 0; k < 3; ++k) {
        a0 += dR[r * 3 + k] * YZ[c * 3 + k];
        a1 += X[k * 3 + r] * dR[k * 3 + c];
      }
      dX[r * 3 + c] = a0;
      dYZ[r * 3 + c] = a1;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double a0 = 0, a1 = 0;
      for (int k = 0; k < 3; ++k) {
        a0 += dYZ[r * 3 + k] * Z[c * 3 + k];
        a1 += Y[k * 3 + r] * dYZ[k * 3 + c];
      }
      dY[r * 3 + c] = a0;
      dZ[r * 3 + c] = a1;
    }
  double g_rx = -sx * dX[4] - cx * dX[5] + cx * dX[7] - sx * dX[8];
  double g_ry = -sy * dY[0] + cy * dY[2] - cy * dY[6] - sy * dY[8];
  double g_rz = -sz * dZ[0] - cz * dZ[1] + cz * dZ[3] - sz * dZ[4];
  // tf.clip_by_value passes the gradient inside the closed interval
  if (!(pv[3] >= -kPiF && pv[3] <= kPiF)) g_rx = 0;
  if (!(pv[4] >= -kPiF && pv[4] <= kPiF)) g_ry = 0;
  if (!(pv[5] >= -kPiF && pv[5] <= kPiF)) g_rz = 0;
  float* o = d_pose + b * dpose_sb + i * dpose_si;
  const double gs = (double)gscale;
  o[0] = (float)(gs * dT[3]);
  o[1] = (float)(gs * dT[7]);
  o[2] = (float)(gs * dT[11]);
  o[3] = (float)(gs * g_rx);
  o[4] = (float)(gs * g_ry);
  o[5] = (float)(gs * g_rz);
}

}  // namespace sfm

using namespace sfm;

extern "C" int sfm_pose_setup(const float* pose, long pose_sb, long pose_si, const float* K, int B, int n_src,
                              int inverse_pose, float* params, void* stream) {
  if (!pose || !K || !params || B <= 0 || n_src <= 0) return SFM_ERR_ARG;
  int items = B * n_src;
  hipLaunchKernelGGL(pose_setup_kernel, dim3((items + 15) / 16), dim3(64), 0, (hipStream_t)stream, pose, pose_sb,
                     pose_si, K, B, n_src, inverse_pose, params);
  return sfm_last_error();
}

int sfm_launch_pose_grad(const float* pose, long pose_sb, long pose_si, const float* K, const float* partials, int n_sets,
                         int nblk, int B, int n_src, int inverse_pose, float gscale, float* d_pose, long dpose_sb,
                         long dpose_si, int* zero, int n_zero, hipStream_t stream) {
  if (!pose || !K || !partials || !d_pose || B <= 0 || n_src <= 0 || nblk <= 0 || n_sets <= 0) return SFM_ERR_ARG;
  int items = B * n_src;
  hipLaunchKernelGGL(sfm::pose_grad_kernel, dim3(items), dim3(sfm::kBlock), 0, stream, pose, pose_sb, pose_si, K, partials,
                     n_sets, nblk, B, n_src, inverse_pose, gscale, d_pose, dpose_sb, dpose_si, zero, n_zero);
  return sfm_last_error();
}

extern "C" int sfm_pose_grad(const float* pose, long pose_sb, long pose_si, const float* K, const float* partials,
                             int n_sets, int nblk, int B, int n_src, int inverse_pose, float gscale, float* d_pose,
                             long dpose_sb, long dpose_si, void* stream) {
  return sfm_launch_pose_grad(pose, pose_sb, pose_si, K, partials, n_sets, nblk, B, n_src, inverse_pose, gscale, d_pose,
                              dpose_sb, dpose_si, nullptr, 0, (hipStream_t)stream);
}
