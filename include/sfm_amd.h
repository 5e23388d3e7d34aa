/* sfm_amd.h - C ABI of libsfm_amd.so, the MI355X (gfx950) implementation of the
 * view-synthesis hot path of alncat/SFM.
 *
 * The reference has no FFI layer: its boundary for this path is the set of Python
 * functions in utils.py, called from SfMLearner_uni.py:120-144.  Each entry point
 * below names the reference function (file:line, relative to the reference repo) whose
 * arithmetic it replaces; `sfm_amd/utils.py` binds them with ctypes behind the
 * reference's own signatures (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - all tensors are dense, contiguous fp32 device (HBM) pointers, layout NHWC;
 *     depth is [B,H,W]; coords are [B,H,W,2] = (x, y) in pixels; int outputs int32;
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only enqueues
 *     work on it - no allocation, no host synchronisation, no global mutable state
 *     (state that outlives a call lives in the caller's workspace and is documented there);
 *   - return value: 0 on success, SFM_ERR_* (< 0) for rejected arguments, otherwise
 *     the hipError_t (> 0) of the failed launch.  Nothing throws across the ABI;
 *   - inputs are never written; workspaces are caller-provided and documented per call;
 *   - `params` is the [n_src*B, SFM_PARAM_STRIDE] table written by sfm_pose_setup
 *     (inverse intrinsics + projection matrix per source and batch item).
 */
#ifndef SFM_AMD_H
#define SFM_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_PARAM_STRIDE 32
#define SFM_ERR_ARG (-1)
#define SFM_ERR_TOO_LARGE (-2)

/* ---- per-batch set-up ------------------------------------------------------------
 * pose_vec2mat utils.py:77-96 (euler2mat 27-75), tf.matrix_inverse(intrinsics) utils.py:112,
 * optional tf.matrix_inverse(pose) utils.py:189/224, K4*T utils.py:195-201 == 230-236.
 * pose element (b, i, k) is read at pose[b*pose_sb + i*pose_si + k]; item i*B+b of
 * params receives Kinv[9], P[12]. */
int sfm_pose_setup(const float* pose, long pose_sb, long pose_si, const float* K, int B, int n_src,
                   int inverse_pose, float* params, void* stream);

/* Adjoint of sfm_pose_setup: sums the dL/dP block partials [n_sets][n_src*B][nblk][12]
 * written by the *_bwd calls and chains them through K4*T, the optional inverse and
 * euler2mat (TF autodiff of the same lines; implicit in SfMLearner_uni.py:306).
 * d_pose element (b, i, k) is written at d_pose[b*dpose_sb + i*dpose_si + k], scaled by gscale. */
int sfm_pose_grad(const float* pose, long pose_sb, long pose_si, const float* K, const float* partials,
                  int n_sets, int nblk, int B, int n_src, int inverse_pose, float gscale, float* d_pose,
                  long dpose_sb, long dpose_si, void* stream);

/* number of dL/dP partial blocks per batch item the *_bwd calls write for an HxW image */
int sfm_bwd_blocks(int H, int W);

/* ---- geometry only: meshgrid 146-170 -> pixel2cam 98-122 -> cam2pixel 124-144 ----
 * coords [B,H,W,2], z [B,H,W] (may be NULL).  trans (may be NULL) = add_trans=False. */
int sfm_project_coords(const float* depth, const float* trans, const float* params, int B, int H, int W,
                       float* coords, float* z, void* stream);

/* ---- the geometry helpers as stand-alone operators (utils.py exports them; the fused kernels keep the same
 * arithmetic in registers and never call these).  Forward: canonical fp32 op order (k-ordered dot products, no
 * FMA, sin / cos rounded once from fp64, partial-pivot LU inverse).
 * euler2mat utils.py:27-75: z, y, x [B] (the reference's [B,1]) -> out [B,1,3,3], R = X.Y.Z, angles clipped to +-pi.
 * pose_vec2mat utils.py:77-96: vec [B,6] = (tx,ty,tz,rx,ry,rz) -> out [B,4,4].
 * meshgrid utils.py:146-170: out [B, 3 (is_homogeneous) or 2, H, W] = (x, y[, 1]) with the fp32 linspace values.
 * pixel2cam utils.py:98-122: depth [B,H,W], pixel_coords [B,3,H,W], K [B,3,3], trans [B,H,W,3] or NULL
 *   (add_trans=False) -> out [B, 4 (is_homogeneous) or 3, H, W]; kinv_ws: scratch of 9*B floats.
 * cam2pixel utils.py:124-144: cam_coords [B,4,H,W], proj [B,4,4] -> coords [B,H,W,2], z [B,H,W,1]. */
int sfm_euler2mat(const float* z, const float* y, const float* x, int B, float* out, void* stream);
int sfm_pose_vec2mat(const float* vec, int B, float* out, void* stream);
int sfm_meshgrid(int B, int H, int W, int is_homogeneous, float* out, void* stream);
int sfm_pixel2cam(const float* depth, const float* pixel_coords, const float* K, const float* trans, int B, int H, int W,
                  int is_homogeneous, float* kinv_ws, float* out, void* stream);
int sfm_cam2pixel(const float* cam_coords, const float* proj, int B, int H, int W, float* coords, float* z, void* stream);
/* Their adjoints (what tf.gradients, SfMLearner_uni.py:306, derives when the caller composes utils.py:185-202 itself).
 * g_out: gradient of the forward's output, same layout.  euler2mat: clip_by_value passes the gradient inside [-pi, pi].
 * pixel2cam: any of g_depth [B,H,W], g_pix [B,3,H,W], g_K [B,3,3] (through the matrix inverse), g_trans [B,H,W,3] may be
 * NULL (not wanted).  cam2pixel: g_coords / g_z may be NULL (that output was unused; not both), g_cam [B,4,H,W] / g_proj
 * [B,4,4] may be NULL (not both).  The sums over the pixels (g_K, g_proj) go through per-workgroup partials added in a
 * fixed order.  ws: sfm_geom_bwd_workspace_floats(B, H, W) floats. */
int sfm_euler2mat_bwd(const float* z, const float* y, const float* x, const float* g_out, int B, float* gz, float* gy, float* gx,
                      void* stream);
int sfm_pose_vec2mat_bwd(const float* vec, const float* g_out, int B, float* g_vec, void* stream);
long long sfm_geom_bwd_workspace_floats(int B, int H, int W);
int sfm_pixel2cam_bwd(const float* depth, const float* pixel_coords, const float* K, const float* trans, const float* g_cam, int B,
                      int H, int W, int is_homogeneous, float* g_depth, float* g_pix, float* g_K, float* g_trans, float* ws,
                      void* stream);
int sfm_cam2pixel_bwd(const float* cam_coords, const float* proj, const float* g_coords, const float* g_z, int B, int H, int W,
                      float* g_cam, float* g_proj, float* ws, void* stream);

/* ---- gather direction --------------------------------------------------------------
 * bilinear_sampler utils.py:241-340.  out [B,Ht,Wt,C], mask [B,Ht,Wt,1].
 * idx_dump (may be NULL): int32 [B,Ht,Wt,4] = clipped (x0, x1, y0, y1) of utils.py:286-289. */
int sfm_bilinear_sampler_fwd(const float* img, const float* coords, int B, int Hs, int Ws, int C, int Ht,
                             int Wt, float* out, float* mask, int* idx_dump, void* stream);
/* d_coords [B,Ht,Wt,2]; d_img (may be NULL) must be zero-filled [B,Hs,Ws,C], accumulated atomically. */
int sfm_bilinear_sampler_bwd(const float* img, const float* coords, const float* gout, int B, int Hs, int Ws,
                             int C, int Ht, int Wt, float* d_coords, float* d_img, void* stream);

/* projective_inverse_warp utils.py:207-239 with the geometry fused (no meshgrid / cam /
 * coords tensors are materialised). */
int sfm_inverse_warp_fwd(const float* img, const float* depth, const float* trans, const float* params, int B,
                         int H, int W, int C, float* out, float* mask, int* idx_dump, void* stream);
/* d_depth [B,H,W], d_trans [B,H,W,3] (iff trans), partials [B][sfm_bwd_blocks][12]. */
int sfm_inverse_warp_bwd(const float* img, const float* depth, const float* trans, const float* params,
                         const float* gout, int B, int H, int W, int C, float* d_depth, float* d_trans,
                         float* partials, float* d_img, void* stream);

/* ---- splat direction ---------------------------------------------------------------
 * bilinear_project utils.py:342-502.  z (may be NULL) = use_depth=False.
 * acc: workspace [B,H,W,C+1] (zeroed by the call; the backward reads it again).
 * out [B,H,W,C], mask [B,H,W,C]. */
int sfm_bilinear_project_fwd(const float* img, const float* coords, const float* z, int B, int H, int W, int C,
                             float* acc, float* out, float* mask, void* stream);
int sfm_bilinear_project_bwd(const float* img, const float* coords, const float* z, const float* acc,
                             const float* gout, int B, int H, int W, int C, float* d_coords, float* d_z,
                             float* d_img, void* stream);

/* projective_warp utils.py:172-205 with the geometry fused. */
int sfm_splat_warp_fwd(const float* img, const float* depth, const float* trans, const float* params, int B,
                       int H, int W, int C, float* acc, float* out, float* mask, void* stream);
int sfm_splat_warp_bwd(const float* img, const float* depth, const float* trans, const float* params,
                       const float* acc, const float* gout, int B, int H, int W, int C, float* d_depth,
                       float* d_trans, float* partials, float* d_img, void* stream);

/* The allocation-free variant of sfm_splat_warp_fwd.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * Same arithmetic and kernels; the [B,H,W,C+1] accumulator lives in a 16 MiB device array inside the library
 * (one copy per GPU, held until the process ends) instead of a caller workspace, so B*H*W*(C+1) must not exceed
 * sfm_splat_warp_fwd_static_max_floats() = 4 Mi floats (SFM_ERR_TOO_LARGE otherwise).  Forward only: the accumulator
 * is not handed back, so there is no matching backward (use sfm_splat_warp_fwd / _bwd for training).  The call
 * allocates nothing (no hipMalloc / hipFree); it must be made with the device of its pointers current (the array's
 * address is looked up per current device on the first call).  Every call on a GPU shares that GPU's array: calls
 * that overlap on one GPU (two streams, two threads) race - serialise them. */
long long sfm_splat_warp_fwd_static_max_floats(void);
int sfm_splat_warp_fwd_static(const float* img, const float* depth, const float* trans, const float* params, int B,
                              int H, int W, int C, float* out, float* mask, void* stream);

/* ---- SSIM: tf_ssim utils.py:548-575 (window 504-519), 3 channels ---------------------
 * out = the raw SSIM map `value` of line 572, [B,H,W,3]. */
int sfm_ssim_fwd(const float* img1, const float* img2, int B, int H, int W, float* out, void* stream);
/* gv = dL/d value; coef = workspace B*H*W*15 floats; d1 / d2 may be NULL (not both). */
int sfm_ssim_bwd(const float* img1, const float* img2, const float* gv, int B, int H, int W, float* coef,
                 float* d1, float* d2, void* stream);

/* ---- fused multi-scale photometric loss: SfMLearner_uni.py:85-168 ----------------------
 * S scales x N sources (2 <= N <= 4) of {projective_inverse_warp :120, projective_warp :128,
 * L1 + SSIM :137-144, masked means :149, min-reprojection :168}, forward and backward,
 * in a fixed number of launches.  Layouts: tgt [B,H,W,3], src [B,H,W,3N] (the reference's
 * src_image_stack), depth [S,B,H,W] (pred_depth resized to HxW, :91), trans [S,B,H,W,3N]
 * (pred_exp_logits resized, :109-113), pose [B,N,6], K [B,3,3], params from
 * sfm_pose_setup(pose, 6N, 6, K, B, N, 0, ...).
 * ws: workspace of sfm_loss_workspace_floats() floats, shared by both calls of a step and
 * initialised ONCE by sfm_loss_workspace_init (it zeroes the splat's overflow accumulator and
 * flags; every forward/backward pair leaves them zero again, so later steps pay no memset).
 * Each sfm_loss_forward must be followed by sfm_loss_backward on the same ws (need_grad 0 or 1)
 * before the next sfm_loss_forward; calls sharing a ws must be stream-ordered.
 *
 * n_elem_global = B_global*H*W*3 is the denominator of every reduce_mean of the loss (the
 * batch is global under data parallelism); need_grad = 0 makes sfm_loss_forward skip the
 * backward seeds (it must then be paired with sfm_loss_backward(need_grad = 0)).
 *
 * sfm_loss_forward writes partials[S*(3N+1)] (fp64): per scale, per source the sums
 * (sum m|Sp-src|, sum m, sum ssim(Sp,src)) and then sum min(e_0,e_1).  Under data
 * parallelism these are all-reduced (sum) across ranks before sfm_loss_backward.
 * sfm_loss_backward turns the (global) sums into the loss (one float at loss_out; the
 * pixel_loss of :149+:168) and, if need_grad, into d_depth [S,B,H,W], d_trans [S,B,H,W,3N]
 * and d_pose [B,N,6], each scaled by gloss (dL/d loss).  n_elem_global = B_global*H*W*3.
 *
 * sfm_loss_forward is the SINGLE-DEVICE forward: the backward seeds of the SSIM term carry the factor
 * mean(proj_src_mask) of :144, 149, i.e. the global mask sums M(s, i) = sum m (utils.py:478-479), which
 * it takes from this device's images alone.  A data-parallel step makes the forward in three calls instead
 * (no counterpart in the single-device reference; same kernels, same arithmetic):
 *   sfm_loss_forward_warps   both warps + this rank's mask sums msum[S*N] (fp64)
 *   [all-reduce (sum) of msum over the ranks - on the communication stream, beside:]
 *   sfm_loss_forward_gather  the gather-side stencil (min-reprojection term)
 *   sfm_loss_forward_splat   the splat-side stencil with the GLOBAL mask sums, then partials[S*(3N+1)]
 *   [all-reduce (sum) of partials]  ->  sfm_loss_backward
 * (_gather before _splat; all on one stream or ordered by events.)
 *
 * The seed guard.  The seeds are written assuming clip((1 - mean ssim) / 2, 0, 1) (utils.py:575) passes its gradient,
 * which it does unless the global mean SSIM of a (scale, source) leaves [-1, 1] (degenerate inputs); for that case
 * the seeds of that pair are rewritten without the SSIM part once the sums are global.  sfm_loss_forward does this in
 * its own reduction (single device: its sums ARE the global ones); sfm_loss_backward does it for the three-call
 * forward, from the all-reduced partials.  need_grad of sfm_loss_backward: 0 = loss only, 1 = gradients (the guard is
 * applied first), 3 = gradients and `partials` are exactly what sfm_loss_forward wrote for this workspace (guard
 * already applied: one launch less). */
long long sfm_loss_workspace_floats(int B, int H, int W, int S, int N);
int sfm_loss_num_partials(int B, int H, int W, int S, int N);
int sfm_loss_workspace_init(float* ws, int B, int H, int W, int S, int N, void* stream);
int sfm_loss_forward(const float* tgt, const float* src, const float* depth, const float* trans,
                     const float* params, float* ws, int B, int H, int W, int S, int N, long long n_elem_global,
                     int need_grad, double* partials, void* stream);
int sfm_loss_forward_warps(const float* tgt, const float* src, const float* depth, const float* trans,
                           const float* params, float* ws, int B, int H, int W, int S, int N, double* msum, void* stream);
int sfm_loss_forward_gather(const float* tgt, float* ws, int B, int H, int W, int S, int N, long long n_elem_global,
                            int need_grad, void* stream);
int sfm_loss_forward_splat(const float* src, float* ws, int B, int H, int W, int S, int N, long long n_elem_global,
                           int need_grad, const double* msum_global, double* partials, void* stream);
int sfm_loss_backward(const float* tgt, const float* src, const float* depth, const float* trans,
                      const float* params, float* ws, int B, int H, int W, int S, int N,
                      long long n_elem_global, float gloss, int need_grad, const double* partials,
                      const float* pose, const float* K, float* loss_out, float* d_depth, float* d_trans,
                      float* d_pose, void* stream);
/* Test / debug aid (no counterpart in the reference): where region `which` of the workspace lies, in floats from ws -
 * 0 Acc, 1 Ovf (the splat's overflow side accumulator), 2 Hb, 3 the per-tile overflow counters (int32), 4 the mask
 * sums (fp64). */
int sfm_loss_workspace_region(int B, int H, int W, int S, int N, int which, long long* offset, long long* count);

/* The same step fed with the prediction PYRAMID instead of maps already resized to HxW: depth_s[s]
 * [B,h[s],w[s]] = pred_depth[s], trans_s[s] [B,h[s],w[s],3N] = pred_exp_logits[s] at their native sizes (HOST arrays of S
 * <= 8 device pointers / sizes, read at call time; h[s] <= H, w[s] <= W).  The kernels evaluate
 * tf.image.resize_bilinear(..., [H, W]) of SfMLearner_uni.py:91, 113 per pixel while loading (TF-1 defaults, the
 * arithmetic of sfm_pyramid_upsample_fwd bit for bit; a level at full size is read directly), so the [S,B,H,W,1+3N]
 * intermediate of sfm_pyramid_upsample_fwd is never written or read.  Everything else - workspace, partials, the
 * all-reduce between the two calls, loss_out, d_pose - as sfm_loss_forward / sfm_loss_backward; sfm_loss_forward_warps_pyr
 * is sfm_loss_forward_warps fed this way (the other two calls of the data-parallel forward read no prediction map).
 * Backward: the adjoint of the resize runs inside the kernel too.  d_depth_s[s] [B,h[s],w[s]] and d_trans_s[s]
 * [B,h[s],w[s],3N] (host arrays of device pointers) receive the gradients of the pyramid's own maps, ADDED to their
 * contents: zero-fill them first, or pass maps that already hold another term of the gradient (the regulariser's,
 * sfm_smooth_loss).  A tile's contributions are summed in LDS and reach a low-resolution map as one global float atomic
 * per cell and value (cells on tile borders have up to four adders: their last bits depend on the order); no
 * [S,B,H,W,1+3N] gradient intermediate exists either.  n_extra (0..16): fp64 values stored behind the S*(3N+1) sums
 * in `partials` - the other terms of total_loss, e.g. the regulariser's share written there by sfm_smooth_loss and
 * all-reduced with the sums - which are added to loss_out (fp32 adds, as total_loss = pixel_loss + smooth_loss is). */
int sfm_loss_forward_pyr(const float* tgt, const float* src, const float* const* depth_s, const float* const* trans_s,
                         const int* h, const int* w, const float* params, float* ws, int B, int H, int W, int S, int N,
                         long long n_elem_global, int need_grad, double* partials, void* stream);
int sfm_loss_forward_warps_pyr(const float* tgt, const float* src, const float* const* depth_s,
                               const float* const* trans_s, const int* h, const int* w, const float* params, float* ws, int B,
                               int H, int W, int S, int N, double* msum, void* stream);
int sfm_loss_backward_pyr(const float* tgt, const float* src, const float* const* depth_s, const float* const* trans_s,
                          const int* h, const int* w, const float* params, float* ws, int B, int H, int W, int S, int N,
                          long long n_elem_global, float gloss, int need_grad, const double* partials, int n_extra,
                          const float* pose, const float* K, float* loss_out, float* const* d_depth_s,
                          float* const* d_trans_s, float* d_pose, void* stream);

/* ---- the prediction pyramid either side of the warp loop ----------------------
 * All scales of a call go in one launch; the per-scale pointer / size arrays are HOST arrays read at call
 * time (S, n_maps <= 16).  No exchange step: under data parallelism pass the global batch and sum the
 * scalar over ranks.  Reductions run in a fixed order (reproducible).
 *
 * pred_disp = d / tf.reduce_mean(d, axis=[1,2,3], keep_dims=True), pred_depth = 1. / pred_disp
 * (SfMLearner_uni.py:40, 47).  disp[s], disp_norm[s], depth[s]: [B, n[s]] (n = h*w*1); mean [S,B];
 * ws: sfm_disp_normalize_workspace_floats(n, S, B) floats (chunk partial sums), undefined between calls. */
long long sfm_disp_normalize_workspace_floats(const int* n, int S, int B);
int sfm_disp_normalize_fwd(const float* const* disp, const int* n, int S, int B, float* const* disp_norm,
                           float* const* depth, float* mean, float* ws, void* stream);
/* g_disp_norm / g_depth (arrays or single entries may be NULL = zero) -> g_disp[s] [B, n[s]]. */
int sfm_disp_normalize_bwd(const float* const* disp, const float* mean, const float* const* g_disp_norm,
                           const float* const* g_depth, const int* n, int S, int B, float* const* g_disp, float* ws,
                           void* stream);

/* tf.image.resize_bilinear(x, [H, W]) with TF-1 defaults (align_corners=False), SfMLearner_uni.py:91, 113:
 * in[s] [B,h[s],w[s],C] -> out [S,B,H,W,C], the stacked layout sfm_loss_forward reads (C = 1 for pred_depth,
 * 3N for pred_exp_logits; C in {1,2,3,4,6,9,12}; h[s] <= H, w[s] <= W).  The adjoint is two separable gather
 * passes (no atomics): gout [S,B,H,W,C] -> gin[s] [B,h[s],w[s],C], written, or added to gin[s] if accumulate;
 * ws: sfm_pyramid_upsample_bwd_workspace_floats(...) floats (the row-reduced intermediate). */
int sfm_pyramid_upsample_fwd(const float* const* in, const int* h, const int* w, int S, int B, int C, int H, int W,
                             float* out, void* stream);
long long sfm_pyramid_upsample_bwd_workspace_floats(const int* h, const int* w, int S, int B, int C, int H, int W);
int sfm_pyramid_upsample_bwd(const float* gout, const int* h, const int* w, int S, int B, int C, int H, int W,
                             int accumulate, float* const* gin, float* ws, void* stream);
/* the same for one map; here C may also be any multiple of 4 (the x2 upsample of disp_bottleneck_dense's feature
 * maps, nets.py:454, 465); ws of _bwd: sfm_resize_bilinear_bwd_workspace_floats(...) floats, 16-byte aligned */
long long sfm_resize_bilinear_bwd_workspace_floats(int B, int h, int w, int C, int H, int W);
int sfm_resize_bilinear_fwd(const float* in, int B, int h, int w, int C, int H, int W, float* out, void* stream);
int sfm_resize_bilinear_bwd(const float* gout, int B, int h, int w, int C, int H, int W, float* gin, float* ws,
                            void* stream);

/* smooth_loss of SfMLearner_uni.py:94-101 with compute_smooth_loss :205-219:
 *   loss = sum_m  w_smooth[m] * compute_smooth_loss(maps[m]) + w_abs[m] * mean|maps[m]|
 * maps[m] [B,h[m],w[m],c[m]] (h, w >= 3); every mean divides by the GLOBAL element count (B_global in
 * place of B).  Value and gradient in one pass: d_maps (NULL, or per map NULL = not wanted) receives
 * gscale * dloss/dmaps[m]; loss_out one float (this rank's share of the global mean).
 * ws: sfm_smooth_workspace_floats(...) floats, 8-byte aligned (the block shares are fp64), contents undefined between
 * calls. */
long long sfm_smooth_workspace_floats(const int* h, const int* w, const int* c, int n_maps, int B);
int sfm_smooth_loss(const float* const* maps, float* const* d_maps, const int* h, const int* w, const int* c,
                    const float* w_smooth, const float* w_abs, int n_maps, int B, int B_global, float gscale, float* ws,
                    float* loss_out, double* loss_out_f64, void* stream);

/* Upstream-gradient scaling (no counterpart in the reference: TF's autodiff seeds d total_loss = 1).  The fused loss
 * computes its gradients together with the value; when the caller's autograd later supplies dL/d total_loss as a DEVICE
 * scalar *g, this multiplies the count listed buffers (bufs / n: host arrays, count <= 34) by *g in place.  *g == 1
 * (every plain backward) makes it a launch that exits at once instead of a pass over all gradients. */
int sfm_scale_grads(float* const* bufs, const long long* n, int count, const float* g, void* stream);

/* ---- ASPP dilated 3x3 convolutions: nets.py:451-471 (disp_bottleneck_dense) --------------------
 * slim.conv2d(x, Cout, [3, 3], rate=rate) of nets.py:457, 458, 463 - the convolution itself,
 * without the arg-scope's batch norm / ELU - as fp32-input MFMA implicit GEMM, and its two
 * gradients (TF autodiff; implicit in SfMLearner_uni.py:306).  in [B,H,W,Cin] NHWC, w [3,3,Cin,Cout]
 * HWIO, out [B,H,W,Cout]; SAME padding, stride 1; Cin and Cout multiples of 32.
 * ws: scratch of sfm_aspp_conv3x3_workspace_floats(B,H,W,Cin,Cout) floats (split-K slabs, flipped
 * weights), 16-byte aligned, contents undefined between calls; calls sharing it must be stream-ordered.
 * Results are bitwise reproducible (no atomics). */
long long sfm_aspp_conv3x3_workspace_floats(int B, int H, int W, int Cin, int Cout);
int sfm_aspp_conv3x3_fwd(const float* in, const float* w, int B, int H, int W, int Cin, int Cout, int rate,
                         float* out, float* ws, void* stream);
int sfm_aspp_conv3x3_bwd_data(const float* gout, const float* w, int B, int H, int W, int Cin, int Cout, int rate,
                              float* gin, float* ws, void* stream);
int sfm_aspp_conv3x3_bwd_weight(const float* in, const float* gout, int B, int H, int W, int Cin, int Cout, int rate,
                                float* gw, float* ws, void* stream);
/* The forward and the weight gradient with `in` a CHANNEL SLICE of a wider NHWC buffer: in = address of the slice's first
 * channel at pixel 0 (16-byte aligned), ld_in = floats between consecutive pixels (a multiple of 4, >= Cin).  This is how
 * the decoder level reads tf.concat([supnet, aspp_up], 3) (nets.py:458) and tf.concat([aspp_up, aspp_up2], 3) (:463)
 * without a concat: the operands already lie side by side in the level's one concat buffer (:467, 469). */
int sfm_aspp_conv3x3_fwd_ld(const float* in, int ld_in, const float* w, int B, int H, int W, int Cin, int Cout, int rate,
                            float* out, float* ws, void* stream);
int sfm_aspp_conv3x3_bwd_weight_ld(const float* in, int ld_in, const float* gout, int B, int H, int W, int Cin, int Cout,
                                   int rate, float* gw, float* ws, void* stream);

/* The same slim.conv2d under is_training=False (the inference graphs, test_kitti_depth_con.py:19-65; arg-scope
 * nets.py:653-656 -> resnet/resnet_utils.py:232-251): convolution, batch norm on the moving statistics
 * y = gamma (x - moving_mean) / sqrt(moving_var + eps) + beta (gamma / beta may be NULL = 1 / 0), tf.nn.elu if act != 0 -
 * in ONE pass: the normalisation and activation are applied where the MFMA kernel stores its sums.  ws as above. */
int sfm_aspp_conv3x3_bn_act_infer(const float* in, const float* w, const float* gamma, const float* beta,
                                  const float* moving_mean, const float* moving_var, float eps, int act, int B, int H,
                                  int W, int Cin, int Cout, int rate, float* out, float* ws, void* stream);
/* The same one-pass convolution + batch norm (moving statistics) + ELU for the inference decoder level without a concat:
 * ntaps = 9 (dilated 3x3) or 1 (1x1); `in` may be a channel slice of a wider NHWC buffer (ld_in floats between pixels) and
 * `out` is written as a channel slice of the level's concat buffer (ld_out) - tf.concat of nets.py:458, 463, 467, 469 is
 * never materialised and every slim.conv2d of the block is ONE kernel pass.  16-byte aligned, strides multiples of 4. */
int sfm_aspp_conv_bn_act_infer_ld(const float* in, int ld_in, const float* w, int ntaps, const float* gamma, const float* beta,
                                  const float* moving_mean, const float* moving_var, float eps, int act, int B, int H, int W,
                                  int Cin, int Cout, int rate, float* out, int ld_out, float* ws, void* stream);

/* The 1x1 convolutions of the same block (nets.py:455, 456, 461, 466, 470): slim.conv2d(x, Cout, [1, 1]) itself, w
 * [1,1,Cin,Cout], through the same implicit-GEMM kernels (centre tap only).  ws: the 3x3 workspace of the shape. */
int sfm_aspp_conv1x1_fwd(const float* in, const float* w, int B, int H, int W, int Cin, int Cout, float* out, float* ws,
                         void* stream);
int sfm_aspp_conv1x1_bwd_data(const float* gout, const float* w, int B, int H, int W, int Cin, int Cout, float* gin,
                              float* ws, void* stream);
int sfm_aspp_conv1x1_bwd_weight(const float* in, const float* gout, int B, int H, int W, int Cin, int Cout, float* gw,
                                float* ws, void* stream);

/* ---- the disparity heads of disp_aspp_u_dense (nets.py:678, 683, 689) ---------------------------------------
 * disp = DISP_SCALING * slim.conv2d(x, 1, [3, 3], activation_fn=tf.sigmoid, normalizer_fn=None) + MIN_DISP with
 * DISP_SCALING = 10, MIN_DISP = 0.01 (nets.py:11-12): x [B,H,W,Cin] NHWC, w [3,3,Cin,1] HWIO, bias [1] (no normaliser,
 * so slim adds a bias), SAME padding -> disp [B,H,W,1]; sig [B,H,W] receives the sigmoid for the backward.  Cin / 4
 * must be a power of two <= 64.  Backward (TF autodiff): gdisp = dL/d disp -> gx [B,H,W,Cin] (may be NULL), gw
 * [3,3,Cin,1] and gb [1] (both or neither); ws: sfm_disp_head_workspace_floats() floats of workgroup partials,
 * summed in a fixed order (reproducible). */
long long sfm_disp_head_workspace_floats(int B, int H, int W, int Cin);
int sfm_disp_head_fwd(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, float* disp,
                      float* sig, void* stream);
int sfm_disp_head_bwd(const float* x, const float* w, const float* sig, const float* gdisp, int B, int H, int W, int Cin,
                      float* gx, float* gw, float* gb, float* ws, void* stream);

/* ---- batch norm + ELU after each ASPP convolution -----------------------------------
 * What the arg-scope of disp_aspp_u_dense (nets.py:653-656 -> resnet/resnet_utils.py:232-251) appends to every
 * slim.conv2d of disp_bottleneck_dense (nets.py:455-470): slim.batch_norm(decay, epsilon, scale=True, fused=True,
 * is_training=True, updates_collections=None) then tf.nn.elu.  x [M, C] = the convolution's output (M = B*H*W rows,
 * NHWC; C a multiple of 4), gamma / beta [C].  Training mode: batch statistics (population
 * variance), saved in save_mean / save_invstd [C] for the backward; moving_mean / moving_var [C] (may be NULL) are
 * updated in place as assign_moving_average does: m -= (m - batch) * fp32(1 - decay), the variance with Bessel's
 * correction (fused_batch_norm).  Reductions in a fixed order.
 * ws: sfm_bn_elu_workspace_floats(M, C) floats, 8-byte aligned, undefined between calls.
 * act = 1: ELU; act = 0: the batch norm alone (the slim.conv2d(..., activation_fn=None) calls, nets.py:455, 470).
 * Backward: gy = dL/dy -> gx [M, C], ggamma [C], gbeta [C] (TF autodiff: EluGrad, FusedBatchNormGrad). */
long long sfm_bn_elu_workspace_floats(long long M, int C);
int sfm_bn_elu_fwd(const float* x, const float* gamma, const float* beta, long long M, int C, float eps, double decay,
                   int act, float* moving_mean, float* moving_var, float* y, float* save_mean, float* save_invstd,
                   float* ws, void* stream);
int sfm_bn_elu_bwd(const float* x, const float* y, const float* gy, const float* gamma, const float* save_mean,
                   const float* save_invstd, long long M, int C, int act, float* gx, float* ggamma, float* gbeta,
                   float* ws, void* stream);
/* The same pair for the concat-free decoder level (tf.concat of nets.py:458, 463, 467, 469 never materialised): the
 * forward writes y as a channel slice of the level's concat buffer (y = address of the slice at row 0, ld_y = floats
 * between rows: a multiple of 4, >= C, 16-byte aligned); the backward reads y from there and takes dL/dy as the SUM of
 * n_gy <= 3 tensors gy[k], each a channel slice with its own row stride ld_gy[k] - the slice of the concat buffer's
 * gradient plus the input gradients of the convolutions that read the same activation - which is the add_n TF's
 * autodiff inserts for a tensor with several consumers.  x, gx stay dense [M, C]. */
int sfm_bn_elu_fwd_ld(const float* x, const float* gamma, const float* beta, long long M, int C, float eps, double decay,
                      int act, float* moving_mean, float* moving_var, float* y, int ld_y, float* save_mean,
                      float* save_invstd, float* ws, void* stream);
/* slim.batch_norm(x, is_training=False) (+ tf.nn.elu if act): the moving statistics instead of the batch's - the batch
 * norms of the inference graph (test_kitti_depth_con.py:19-65) that do not follow a convolution (nets.py:672, 460; the
 * ones that do are the epilogue of sfm_aspp_conv_bn_act_infer_ld).  y may be a channel slice (ld_y). */
int sfm_bn_elu_infer(const float* x, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                     long long M, int C, float eps, int act, float* y, int ld_y, void* stream);
int sfm_bn_elu_bwd_ld(const float* x, const float* y, int ld_y, const float* const* gy, const int* ld_gy, int n_gy,
                      const float* gamma, const float* save_mean, const float* save_invstd, long long M, int C, int act,
                      float* gx, float* ggamma, float* gbeta, float* ws, void* stream);

/* The same batch norm with its statistics taken over ALL ranks of a data-parallel job (the batch of
 * SfMLearner_uni.py:36-45 sharded over GPUs; the reference is single-device, so this is its graph on the global
 * batch).  The collectives stay with the caller (torch.distributed over RCCL):
 *   forward : sfm_bn_local_stats -> all-gather of stats -> sfm_bn_elu_apply_stats
 *   backward: sfm_bn_elu_bwd_sums -> all-reduce (sum) of ggamma / gbeta copies -> sfm_bn_elu_bwd_apply
 * stats [3][C] fp64 = (row count, mean, sum of squared deviations) of this rank's M rows; stats_all [world][3][C] =
 * the gathered triples, merged in rank order (Chan's update: identical on every rank).  sfm_bn_elu_bwd_sums returns
 * this rank's share of dL/dgamma, dL/dbeta (to be summed over ranks with the other parameter gradients);
 * sfm_bn_elu_bwd_apply takes the sums over all ranks and the global row count.  ws as sfm_bn_elu_fwd. */
int sfm_bn_local_stats(const float* x, long long M, int C, double* stats, float* ws, void* stream);
int sfm_bn_elu_apply_stats(const float* x, const float* gamma, const float* beta, const double* stats_all, int world,
                           long long M, int C, float eps, double decay, int act, float* moving_mean, float* moving_var,
                           float* y, float* save_mean, float* save_invstd, void* stream);
int sfm_bn_elu_bwd_sums(const float* x, const float* y, const float* gy, const float* save_mean, const float* save_invstd,
                        long long M, int C, int act, float* ggamma, float* gbeta, float* ws, void* stream);
int sfm_bn_elu_bwd_apply(const float* x, const float* y, const float* gy, const float* gamma, const float* save_mean,
                         const float* save_invstd, const float* ggamma_all, const float* gbeta_all, long long M,
                         long long M_global, int C, int act, float* gx, void* stream);

/* ---- snippet unpack, augmentation and preprocessing (data_loader.py) ---------------
 * seq: the decoded uint8 snippets [B, H, (n_src+1) W, 3] exactly as tf.image.decode_jpeg returns the files
 * data/prepare_train_data.py:47 writes (frames side by side), already in device memory; K [B,3,3] from the
 * <frame>_cam.txt lines (:49-51).  Output: tgt [B,H,W,3] = the centre frame, src [B,H,W,3 n_src] = the frames
 * before then after it, stacked along the channels (unpack_image_sequence, data_loader.py:296-324), both mapped to
 * [-1, 1] fp32 (preprocess_image, SfMLearner_uni.py:449-452); K_ms [B,S,3,3] (get_multi_scale_intrinsics,
 * data_loader.py:348-359).  n_src even, 2 or 4.  augment = 0 is fetch_val_batch (:60-73); augment = 1 is
 * fetch_train_batch (:126-144) with the random draws made by the caller: x *= brightness (:239-242), x *= col_c
 * per colour channel (:232-238), uint8 round trip (:272-276), swap != 0 mirrors every frame along the width
 * (:227-231, intrinsics untouched as in the reference), adjust_contrast by `contrast` about the per-image channel
 * means (:137), uint8 again.  ws: sfm_snippet_workspace_floats(...) floats, 8-byte aligned (unused if !augment). */
long long sfm_snippet_workspace_floats(int B, int H, int W, int n_src);
int sfm_snippet_unpack(const unsigned char* seq, const float* K, int B, int H, int W, int n_src, int S, int augment,
                       float brightness, float col_r, float col_g, float col_b, int swap, float contrast, float* tgt,
                       float* src, float* K_ms, float* ws, void* stream);

/* ---- measurement aid (no counterpart in the reference; used by bench.py only) -----------
 * When enabled, sfm_loss_forward / sfm_loss_backward bracket each of their launches with
 * HIP events on the caller's stream; sfm_prof_read adds the last step's elapsed ms per stage
 * into ms[sfm_prof_stages()] (order: tile_shift, warp_fwd, stencil, reduce, finalize / seed guard,
 * warp_bwd, pose_grad, mask sums) after the caller synchronised.  Process-wide, off by default. */
int sfm_prof_enable(int on);
int sfm_prof_stages(void);
int sfm_prof_read(float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SFM_AMD_H */
