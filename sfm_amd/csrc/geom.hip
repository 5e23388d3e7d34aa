// The geometry helpers of utils.py as stand-alone operators (forward), in the canonical fp32 op order of the oracle:
//   euler2mat utils.py:27-75, pose_vec2mat :77-96, meshgrid :146-170, pixel2cam :98-122, cam2pixel :124-144.
// The fused kernels never call these (they keep the same arithmetic in registers: sfm_device.h geom_fwd, pose.hip); the
// reference exports the helpers from utils.py, so the drop-in does too.  Layouts are the reference's: rotation
// [B,1,3,3], transform [B,4,4], pixel grid [B,3|2,H,W] and camera points [B,4|3,H,W] channel-major (NCHW),
// coordinates [B,H,W,2] and depth [B,H,W,1] NHWC.  Compiled with -ffp-contract=off like every index-feeding file.
#include "sfm_device.h"
#include "sfm_internal.h"

namespace sfm {
namespace {

constexpr float kPiG = 3.14159274101257324f;  // fp32(np.pi)

// R = (X . Y) . Z from the clipped angles; sin / cos correctly rounded (fp64, one rounding)
__device__ void euler_rot(float az, float ay, float ax, float* __restrict__ R) {
  const float z = fminf(fmaxf(az, -kPiG), kPiG), y = fminf(fmaxf(ay, -kPiG), kPiG), x = fminf(fmaxf(ax, -kPiG), kPiG);
  double sd, cd;
  sincos((double)z, &sd, &cd);
  const float sz = (float)sd, cz = (float)cd;
  sincos((double)y, &sd, &cd);
  const float sy = (float)sd, cy = (float)cd;
  sincos((double)x, &sd, &cd);
  const float sx = (float)sd, cx = (float)cd;
  const float Z[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};
  const float Y[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy};
  const float X[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx};
  float XY[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = X[i * 3] * Y[j];
      for (int k = 1; k < 3; ++k) acc = acc + X[i * 3 + k] * Y[k * 3 + j];
      XY[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = XY[i * 3] * Z[j];
      for (int k = 1; k < 3; ++k) acc = acc + XY[i * 3 + k] * Z[k * 3 + j];
      R[i * 3 + j] = acc;
    }
}

__global__ __launch_bounds__(64) void euler2mat_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                       const float* __restrict__ x, int B, float* __restrict__ out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float R[9];
  euler_rot(z[b], y[b], x[b], R);
  for (int k = 0; k < 9; ++k) out[(size_t)b * 9 + k] = R[k];
}

__global__ __launch_bounds__(64) void pose_vec2mat_kernel(const float* __restrict__ vec, int B, float* __restrict__ out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* v = vec + (size_t)b * 6;
  float R[9];
  euler_rot(v[5], v[4], v[3], R);  // utils.py:90: euler2mat(rz, ry, rx)
  float* o = out + (size_t)b * 16;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o[r * 4 + c] = R[r * 3 + c];
    o[r * 4 + 3] = v[r];
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

// [B, nch, H, W]: x, y (, 1)
__global__ __launch_bounds__(kBlock) void meshgrid_kernel(int B, int H, int W, int nch, float step_x, float step_y,
                                                          float* __restrict__ out) {
  const long n = (long)B * H * W;
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const int u = (int)(e % W), v = (int)((e / W) % H);
  const long b = e / ((long)W * H);
  const long hw = (long)H * W, p = (long)v * W + u;
  float* o = out + b * nch * hw + p;
  o[0] = grid_coord(u, step_x, (float)(W - 1));
  o[hw] = grid_coord(v, step_y, (float)(H - 1));
  if (nch == 3) o[2 * hw] = 1.0f;
}

// 3x3 inverse of the intrinsics in the oracle's LU order (same routine as pose.hip; kept local: 3x3 only)
__device__ void inv3_lu(const float* __restrict__ a, float* __restrict__ inv) {
  float lu[3][3], rhs[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { lu[i][j] = a[i * 3 + j]; rhs[i][j] = (i == j) ? 1.f : 0.f; }
  for (int k = 0; k < 3; ++k) {
    int p = k;
    float best = fabsf(lu[k][k]);
    for (int i = k + 1; i < 3; ++i) { const float v = fabsf(lu[i][k]); if (v > best) { best = v; p = i; } }
    if (p != k)
      for (int j = 0; j < 3; ++j) {
        const float t = lu[k][j]; lu[k][j] = lu[p][j]; lu[p][j] = t;
        const float u = rhs[k][j]; rhs[k][j] = rhs[p][j]; rhs[p][j] = u;
      }
    for (int i = k + 1; i < 3; ++i) {
      const float l = lu[i][k] / lu[k][k];
      lu[i][k] = l;
      for (int j = k + 1; j < 3; ++j) lu[i][j] = lu[i][j] - l * lu[k][j];
    }
  }
  float y[3][3];
  for (int i = 0; i < 3; ++i)
    for (int c = 0; c < 3; ++c) {
      float acc = rhs[i][c];
      for (int j = 0; j < i; ++j) acc = acc - lu[i][j] * y[j][c];
      y[i][c] = acc;
    }
  for (int i = 2; i >= 0; --i)
    for (int c = 0; c < 3; ++c) {
      float acc = y[i][c];
      for (int j = i + 1; j < 3; ++j) acc = acc - lu[i][j] * inv[j * 3 + c];
      inv[i * 3 + c] = acc / lu[i][i];
    }
}

__global__ __launch_bounds__(64) void inv3_kernel(const float* __restrict__ K, int B, float* __restrict__ Kinv) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float o[9];
  inv3_lu(K + (size_t)b * 9, o);
  for (int k = 0; k < 9; ++k) Kinv[(size_t)b * 9 + k] = o[k];
}

// cam = (Kinv . pix) * depth (+ t (.) cam) (, 1): pix [B,3,H,W], depth [B,H,W], t [B,H,W,3] -> [B,nch,H,W]
__global__ __launch_bounds__(kBlock) void pixel2cam_kernel(const float* __restrict__ depth, const float* __restrict__ pix,
                                                           const float* __restrict__ Kinv, const float* __restrict__ trans,
                                                           int B, int H, int W, int nch, float* __restrict__ out) {
  const long hw = (long)H * W, n = (long)B * hw;
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const long b = e / hw, p = e - b * hw;
  const float* ki = Kinv + b * 9;
  const float* px = pix + b * 3 * hw + p;
  const float p0 = px[0], p1 = px[hw], p2 = px[2 * hw], d = depth[e];
  float c[3];
  for (int r = 0; r < 3; ++r) c[r] = ((ki[r * 3] * p0 + ki[r * 3 + 1] * p1) + ki[r * 3 + 2] * p2) * d;
  if (trans) {
    const f3 t = ld3(trans + e * 3);
    c[0] = c[0] + t.x * c[0];
    c[1] = c[1] + t.y * c[1];
    c[2] = c[2] + t.z * c[2];
  }
  float* o = out + b * nch * hw + p;
  o[0] = c[0]; o[hw] = c[1]; o[2 * hw] = c[2];
  if (nch == 4) o[3 * hw] = 1.0f;
}

// unnorm = proj . cam; coords = (x/(z+1e-10), y/(z+1e-10)), depth = z: cam [B,4,H,W], proj [B,4,4]
__global__ __launch_bounds__(kBlock) void cam2pixel_kernel(const float* __restrict__ cam, const float* __restrict__ proj,
                                                           int B, int H, int W, float* __restrict__ coords,
                                                           float* __restrict__ z) {
  const long hw = (long)H * W, n = (long)B * hw;
  const long e = (long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const long b = e / hw, p = e - b * hw;
  const float* pr = proj + b * 16;
  const float* cm = cam + b * 4 * hw + p;
  const float c0 = cm[0], c1 = cm[hw], c2 = cm[2 * hw], c3 = cm[3 * hw];
  float q[3];
  for (int r = 0; r < 3; ++r) q[r] = ((pr[r * 4] * c0 + pr[r * 4 + 1] * c1) + pr[r * 4 + 2] * c2) + pr[r * 4 + 3] * c3;
  const float den = q[2] + 1e-10f;
  coords[e * 2] = q[0] / den;
  coords[e * 2 + 1] = q[1] / den;
  z[e] = q[2];
}

// ------------------------------------------------------------------ adjoints
// What tf.gradients (SfMLearner_uni.py:306) derives for a caller that composes the reference's pieces itself.  The fused
// kernels never call these (their adjoint is geom_bwd in sfm_device.h and pose.hip); they exist so that the drop-in
// helpers are hand-written in both directions.  Reductions over the pixels (dL/dK, dL/dproj) go through per-workgroup
// partial sums added in a fixed order: no atomics, reproducible.
constexpr float kPiB = 3.14159274101257324f;  // fp32(np.pi)

__device__ __forceinline__ void mm3(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ float dot9(const float* a, const float* b) {
  float s = 0.f;
  for (int k = 0; k < 9; ++k) s += a[k] * b[k];
  return s;
}

// R = X(x) Y(y) Z(z) with the angles clipped to [-pi, pi] (gradient 1 inside the closed interval, 0 outside:
// clip_by_value).  gR [9] -> (gz, gy, gx)
__device__ void euler_rot_bwd(float az, float ay, float ax, const float* __restrict__ gR, float& gz, float& gy, float& gx) {
  const float z = fminf(fmaxf(az, -kPiB), kPiB), y = fminf(fmaxf(ay, -kPiB), kPiB), x = fminf(fmaxf(ax, -kPiB), kPiB);
  double sd, cd;
  sincos((double)z, &sd, &cd);
  const float sz = (float)sd, cz = (float)cd;
  sincos((double)y, &sd, &cd);
  const float sy = (float)sd, cy = (float)cd;
  sincos((double)x, &sd, &cd);
  const float sx = (float)sd, cx = (float)cd;
  const float Z[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f}, dZ[9] = {-sz, -cz, 0.f, cz, -sz, 0.f, 0.f, 0.f, 0.f};
  const float Y[9] = {cy, 0.f, sy, 0.f, 1.f, 0.f, -sy, 0.f, cy}, dY[9] = {-sy, 0.f, cy, 0.f, 0.f, 0.f, -cy, 0.f, -sy};
  const float X[9] = {1.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, sx, cx}, dX[9] = {0.f, 0.f, 0.f, 0.f, -sx, -cx, 0.f, cx, -sx};
  float t[9], d[9];
  mm3(dX, Y, t); mm3(t, Z, d);
  gx = (ax >= -kPiB && ax <= kPiB) ? dot9(gR, d) : 0.f;
  mm3(X, dY, t); mm3(t, Z, d);
  gy = (ay >= -kPiB && ay <= kPiB) ? dot9(gR, d) : 0.f;
  mm3(X, Y, t); mm3(t, dZ, d);
  gz = (az >= -kPiB && az <= kPiB) ? dot9(gR, d) : 0.f;
}

__global__ __launch_bounds__(64) void euler2mat_bwd_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ gR, int B,
                                                           float* __restrict__ gz, float* __restrict__ gy,
                                                           float* __restrict__ gx) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float g[9], a, c, d;
  for (int k = 0; k < 9; ++k) g[k] = gR[(size_t)b * 9 + k];
  euler_rot_bwd(z[b], y[b], x[b], g, a, c, d);
  gz[b] = a; gy[b] = c; gx[b] = d;
}

__global__ __launch_bounds__(64) void pose_vec2mat_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ gT, int B,
                                                              float* __restrict__ gvec) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* v = vec + (size_t)b * 6;
  const float* g = gT + (size_t)b * 16;
  float gR[9], gz, gy, gx;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) gR[r * 3 + c] = g[r * 4 + c];
  euler_rot_bwd(v[5], v[4], v[3], gR, gz, gy, gx);
  float* o = gvec + (size_t)b * 6;
  o[0] = g[3]; o[1] = g[7]; o[2] = g[11];   // the translation column
  o[3] = gx; o[4] = gy; o[5] = gz;          // euler2mat(rz, ry, rx), utils.py:90
}

// cam = (Kinv pix) d (1 + t): per pixel dL/dd, dL/dt, dL/dpix; per workgroup the partial sums of dL/dKinv (9 values)
__global__ __launch_bounds__(kBlock) void pixel2cam_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ pix,
                                                               const float* __restrict__ Kinv, const float* __restrict__ trans,
                                                               const float* __restrict__ gcam, int H, int W, int nch,
                                                               float* __restrict__ g_depth, float* __restrict__ g_pix,
                                                               float* __restrict__ g_trans, float* __restrict__ partials) {
  __shared__ float red[4 * 9];
  const long hw = (long)H * W;
  const long b = blockIdx.y, p = (long)blockIdx.x * kBlock + threadIdx.x;
  float gk[9];
  for (int k = 0; k < 9; ++k) gk[k] = 0.f;
  if (p < hw) {
    const long e = b * hw + p;
    const float* ki = Kinv + b * 9;
    const float* px = pix + b * 3 * hw + p;
    const float* gc = gcam + b * nch * hw + p;
    const float pv[3] = {px[0], px[hw], px[2 * hw]}, d = depth[e];
    float r[3], g_c[3] = {gc[0], gc[hw], gc[2 * hw]};
    for (int i = 0; i < 3; ++i) r[i] = (ki[i * 3] * pv[0] + ki[i * 3 + 1] * pv[1]) + ki[i * 3 + 2] * pv[2];
    if (trans) {
      const f3 t = ld3(trans + e * 3);
      const float tt[3] = {t.x, t.y, t.z};
      float gt[3];
      for (int i = 0; i < 3; ++i) {
        gt[i] = g_c[i] * (r[i] * d);       // dL/dt = dL/dcam (.) c
        g_c[i] = g_c[i] + g_c[i] * tt[i];  // dL/dc = dL/dcam (1 + t)
      }
      if (g_trans) st3(g_trans + e * 3, gt[0], gt[1], gt[2]);
    }
    if (g_depth) g_depth[e] = (g_c[0] * r[0] + g_c[1] * r[1]) + g_c[2] * r[2];
    float g_r[3];
    for (int i = 0; i < 3; ++i) g_r[i] = g_c[i] * d;
    if (g_pix) {
      float* o = g_pix + b * 3 * hw + p;
      for (int j = 0; j < 3; ++j) o[j * hw] = (ki[j] * g_r[0] + ki[3 + j] * g_r[1]) + ki[6 + j] * g_r[2];
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) gk[i * 3 + j] = g_r[i] * pv[j];
  }
  if (partials) block_sum_store<9>(gk, red, partials + (b * gridDim.x + blockIdx.x) * 9);
}

// dL/dKinv = sum of the workgroup partials (one wave per batch item, fixed order); dL/dK = -Kinv^T dL/dKinv Kinv^T
__global__ __launch_bounds__(64) void pixel2cam_gk_kernel(const float* __restrict__ partials, const float* __restrict__ Kinv,
                                                          int nblk, float* __restrict__ gK) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s[9];
  for (int k = 0; k < 9; ++k) s[k] = 0.f;
  for (int j = lane; j < nblk; j += 64)
    for (int k = 0; k < 9; ++k) s[k] += partials[((long)b * nblk + j) * 9 + k];
  for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
  if (lane != 0) return;
  const float* ki = Kinv + (long)b * 9;
  float kt[9], t[9], o[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) kt[i * 3 + j] = ki[j * 3 + i];
  mm3(kt, s, t);
  mm3(t, kt, o);
  for (int k = 0; k < 9; ++k) gK[(long)b * 9 + k] = -o[k];
}

// q = P cam, (x, y) = (q0, q1) / (q2 + 1e-10), z = q2: dL/dcam per pixel, per workgroup the partial sums of dL/dP (12 values)
__global__ __launch_bounds__(kBlock) void cam2pixel_bwd_kernel(const float* __restrict__ cam, const float* __restrict__ proj,
                                                               const float* __restrict__ g_coords, const float* __restrict__ g_z,
                                                               int H, int W, float* __restrict__ g_cam,
                                                               float* __restrict__ partials) {
  __shared__ float red[4 * 12];
  const long hw = (long)H * W;
  const long b = blockIdx.y, p = (long)blockIdx.x * kBlock + threadIdx.x;
  float gp[12];
  for (int k = 0; k < 12; ++k) gp[k] = 0.f;
  if (p < hw) {
    const long e = b * hw + p;
    const float* pr = proj + b * 16;
    const float* cm = cam + b * 4 * hw + p;
    const float c[4] = {cm[0], cm[hw], cm[2 * hw], cm[3 * hw]};
    float q[3];
    for (int r = 0; r < 3; ++r) q[r] = ((pr[r * 4] * c[0] + pr[r * 4 + 1] * c[1]) + pr[r * 4 + 2] * c[2]) + pr[r * 4 + 3] * c[3];
    const float den = q[2] + 1e-10f, iden = 1.0f / den;
    const float gx = g_coords ? g_coords[e * 2] : 0.f, gy = g_coords ? g_coords[e * 2 + 1] : 0.f, gz = g_z ? g_z[e] : 0.f;
    float gq[3];
    gq[0] = gx * iden;
    gq[1] = gy * iden;
    gq[2] = gz - (gx * q[0] + gy * q[1]) * (iden * iden);
    if (g_cam) {
      float* o = g_cam + b * 4 * hw + p;
      for (int j = 0; j < 4; ++j) o[j * hw] = (pr[j] * gq[0] + pr[4 + j] * gq[1]) + pr[8 + j] * gq[2];
    }
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < 4; ++j) gp[r * 4 + j] = gq[r] * c[j];
  }
  if (partials) block_sum_store<12>(gp, red, partials + (b * gridDim.x + blockIdx.x) * 12);
}

__global__ __launch_bounds__(64) void cam2pixel_gproj_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ g_proj) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s[12];
  for (int k = 0; k < 12; ++k) s[k] = 0.f;
  for (int j = lane; j < nblk; j += 64)
    for (int k = 0; k < 12; ++k) s[k] += partials[((long)b * nblk + j) * 12 + k];
  for (int k = 0; k < 12; ++k) s[k] = wave_sum(s[k]);
  if (lane != 0) return;
  for (int k = 0; k < 12; ++k) g_proj[(long)b * 16 + k] = s[k];
  for (int k = 12; k < 16; ++k) g_proj[(long)b * 16 + k] = 0.f;   // the last row of proj never enters the outputs
}

}  // namespace
}  // namespace sfm

using namespace sfm;

extern "C" int sfm_euler2mat(const float* z, const float* y, const float* x, int B, float* out, void* stream) {
  if (!z || !y || !x || !out || B <= 0) return SFM_ERR_ARG;
  hipLaunchKernelGGL(euler2mat_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, z, y, x, B, out);
  return sfm_last_error();
}

extern "C" int sfm_pose_vec2mat(const float* vec, int B, float* out, void* stream) {
  if (!vec || !out || B <= 0) return SFM_ERR_ARG;
  hipLaunchKernelGGL(pose_vec2mat_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, vec, B, out);
  return sfm_last_error();
}

extern "C" int sfm_meshgrid(int B, int H, int W, int is_homogeneous, float* out, void* stream) {
  if (!out || B <= 0 || H <= 0 || W <= 0) return SFM_ERR_ARG;
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(meshgrid_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, B, H, W,
                     is_homogeneous ? 3 : 2, sfm_lin_step(W), sfm_lin_step(H), out);
  return sfm_last_error();
}

extern "C" int sfm_pixel2cam(const float* depth, const float* pixel_coords, const float* K, const float* trans, int B,
                             int H, int W, int is_homogeneous, float* kinv_ws, float* out, void* stream) {
  if (!depth || !pixel_coords || !K || !kinv_ws || !out || B <= 0 || H <= 0 || W <= 0) return SFM_ERR_ARG;
  hipLaunchKernelGGL(inv3_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, K, B, kinv_ws);
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(pixel2cam_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, depth,
                     pixel_coords, kinv_ws, trans, B, H, W, is_homogeneous ? 4 : 3, out);
  return sfm_last_error();
}

extern "C" int sfm_cam2pixel(const float* cam_coords, const float* proj, int B, int H, int W, float* coords, float* z,
                             void* stream) {
  if (!cam_coords || !proj || !coords || !z || B <= 0 || H <= 0 || W <= 0) return SFM_ERR_ARG;
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(cam2pixel_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     cam_coords, proj, B, H, W, coords, z);
  return sfm_last_error();
}

extern "C" int sfm_euler2mat_bwd(const float* z, const float* y, const float* x, const float* g_out, int B, float* gz, float* gy,
                                 float* gx, void* stream) {
  if (!z || !y || !x || !g_out || !gz || !gy || !gx || B <= 0) return SFM_ERR_ARG;
  hipLaunchKernelGGL(euler2mat_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, z, y, x, g_out, B, gz, gy, gx);
  return sfm_last_error();
}

extern "C" int sfm_pose_vec2mat_bwd(const float* vec, const float* g_out, int B, float* g_vec, void* stream) {
  if (!vec || !g_out || !g_vec || B <= 0) return SFM_ERR_ARG;
  hipLaunchKernelGGL(pose_vec2mat_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, vec, g_out, B, g_vec);
  return sfm_last_error();
}

// workspace of the two per-pixel adjoints: 9 B floats (Kinv) + 12 floats per workgroup of 256 pixels and batch item
extern "C" long long sfm_geom_bwd_workspace_floats(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return SFM_ERR_ARG;
  const long nblk = ((long)H * W + kBlock - 1) / kBlock;
  return 12L * B + 12L * B * nblk;
}

// g_cam [B, 4|3, H, W] -> g_depth [B,H,W], g_pix [B,3,H,W], g_K [B,3,3], g_trans [B,H,W,3] (each may be NULL: not wanted)
extern "C" int sfm_pixel2cam_bwd(const float* depth, const float* pixel_coords, const float* K, const float* trans,
                                 const float* g_cam, int B, int H, int W, int is_homogeneous, float* g_depth, float* g_pix,
                                 float* g_K, float* g_trans, float* ws, void* stream_) {
  if (!depth || !pixel_coords || !K || !g_cam || !ws || B <= 0 || H <= 0 || W <= 0 || B > 65535 || (g_trans && !trans))
    return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  float* kinv = ws;
  float* partials = ws + 12L * B;
  hipLaunchKernelGGL(inv3_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, K, B, kinv);
  const int nblk = (int)(((long)H * W + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(pixel2cam_bwd_kernel, dim3(nblk, B), dim3(kBlock), 0, stream, depth, pixel_coords, kinv, trans, g_cam, H, W,
                     is_homogeneous ? 4 : 3, g_depth, g_pix, g_trans, g_K ? partials : (float*)nullptr);
  if (g_K) hipLaunchKernelGGL(pixel2cam_gk_kernel, dim3(B), dim3(64), 0, stream, partials, kinv, nblk, g_K);
  return sfm_last_error();
}

// g_coords [B,H,W,2], g_z [B,H,W,1] (either may be NULL: that output was not used) -> g_cam [B,4,H,W], g_proj [B,4,4]
extern "C" int sfm_cam2pixel_bwd(const float* cam_coords, const float* proj, const float* g_coords, const float* g_z, int B, int H,
                                 int W, float* g_cam, float* g_proj, float* ws, void* stream_) {
  if (!cam_coords || !proj || (!g_coords && !g_z) || (!g_cam && !g_proj) || !ws || B <= 0 || H <= 0 || W <= 0 || B > 65535)
    return SFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  float* partials = ws + 12L * B;
  const int nblk = (int)(((long)H * W + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(cam2pixel_bwd_kernel, dim3(nblk, B), dim3(kBlock), 0, stream, cam_coords, proj, g_coords, g_z, H, W, g_cam,
                     g_proj ? partials : (float*)nullptr);
  if (g_proj) hipLaunchKernelGGL(cam2pixel_gproj_kernel, dim3(B), dim3(64), 0, stream, partials, nblk, g_proj);
  return sfm_last_error();
}
