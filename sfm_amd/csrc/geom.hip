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
