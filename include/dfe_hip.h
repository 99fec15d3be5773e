/* libdfe_hip.so -- C ABI of the MI355X (gfx950) photometric-warping loss stack.
 *
 * Drop-in boundary for the hot path of jianfenglihg/Unsupervised_depth_OpticalFlow_egomotion.
 * The reference has no FFI: its boundary is the Python signatures cited on each entry point
 * below (paths relative to the reference checkout).  A maintainer binds these symbols with
 * ctypes from the cited Python function (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 NCHW data unless it says "host";
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing here allocates,
 *     frees or synchronises; all work is enqueued on `stream` (a hipStream_t, NULL = default);
 *   - return value: DFE_OK (0) or a negative DFE_ERR_* code; no C++ exception crosses the ABI;
 *   - no module-global state: entry points are re-entrant and may be called from the autograd
 *     backward thread (contrast the racy pixel_coords cache, inverse_warp.py:6-18);
 *   - `align_corners` is the grid_sample convention (the reference leaves it to the installed
 *     torch; 0 is what torch >= 1.3 does, 1 is torch <= 1.2).
 */
#ifndef DFE_HIP_H
#define DFE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DFE_OK 0
#define DFE_ERR_NULL (-1)        /* a required pointer is NULL */
#define DFE_ERR_DIMS (-2)        /* a dimension is out of range */
#define DFE_ERR_LAUNCH (-3)      /* hipGetLastError() reported a launch failure */
#define DFE_ERR_UNSUPPORTED (-4) /* argument value outside what the kernels implement */
#define DFE_ERR_WORKSPACE (-5)   /* workspace too small */

#define DFE_MAX_SCALES 8
#define DFE_ABI_VERSION 1

int dfe_abi_version(void);
const char* dfe_error_string(int code);

/* ---- cameras ----------------------------------------------------------------------------
 * Per (sample, direction, scale) projection block: K_s = K with rows 0-1 / downscale[s]
 * (model_geometry.py:92-93,696-697), K_s^-1, A = K_s R, b = K_s t with R = Rx Ry Rz
 * (inverse_warp.py:110-145,172-187,284-291).  pose [B,ndir,6]; K [B,3,3];
 * cams: B*ndir*nscale*dfe_camera_floats() floats, index (b*ndir+d)*nscale+s;
 * downscale_host: nscale host floats. */
int dfe_camera_floats(void);
int dfe_prepare_cameras(const float* pose, const float* K, float* cams, int B, int ndir, int nscale,
                        const float* downscale_host, void* stream);

/* pose_vec2mat (inverse_warp.py:172-187) -> T34 [n,3,4] and/or compute_essential_matrix
 * (inverse_warp.py:354-364) -> E [n,3,3]; either output may be NULL. */
int dfe_pose_vec2mat_fwd(const float* vec, float* T34, float* E, int n, void* stream);
int dfe_pose_vec2mat_bwd(const float* vec, const float* gT34, const float* gE, float* gvec, int n, void* stream);

/* ---- warp_flow(x, flow, use_mask)  net_utils.py:16-54 --------------------------------------
 * x [B,C,H,W], flow [B,2,H,W] -> out [B,C,H,W].  Backward: gflow [B,2,H,W] (NULL to skip),
 * gx [B,C,H,W] scatter-added with atomics, must be zero-initialised by the caller (NULL to skip). */
int dfe_warp_flow_fwd(const float* x, const float* flow, float* out, int B, int C, int H, int W, int use_mask,
                      int align_corners, void* stream);
int dfe_warp_flow_bwd(const float* x, const float* flow, const float* gout, float* gflow, float* gx, int B, int C,
                      int H, int W, int use_mask, int align_corners, void* stream);

/* ---- inverse_warp2(img, depth, ref_depth, pose, intrinsics)  inverse_warp.py:263-303 -------
 * cams from dfe_prepare_cameras(pose[B,6], K, cams, B, 1, 1, {1}).  Outputs: projected image
 * [B,3,H,W], valid mask [B,1,H,W], projected depth [B,1,H,W], computed depth [B,1,H,W]
 * (the last three may be NULL).  Backward: g_depth [B,1,H,W], g_refdepth [B,1,H,W]
 * (zero-initialised by the caller, NULL to skip), g_pose [B,6]; partials: workspace of
 * dfe_pose_partials_floats(B,H,W) floats. */
int dfe_pose_partials_floats(int B, int H, int W);
int dfe_inverse_warp2_fwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          float* out_img, float* out_valid, float* out_pdepth, float* out_cdepth, int B, int H, int W,
                          int align_corners, void* stream);
int dfe_inverse_warp2_bwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          const float* g_img, const float* g_pdepth, const float* g_cdepth, float* g_depth,
                          float* g_refdepth, float* g_pose, float* partials, int B, int H, int W, int align_corners,
                          void* stream);

/* ---- calculate_rigid_flow(depth, pose, intrinsics)  inverse_warp.py:311-342 ---------------- */
int dfe_rigid_flow_fwd(const float* depth, const float* cams, float* out, int B, int H, int W, void* stream);
int dfe_rigid_flow_bwd(const float* depth, const float* cams, const float* gout, float* g_depth, float* g_pose,
                       float* partials, int B, int H, int W, void* stream);

/* ---- SSIM(x, y)  pytorch_ssim/ssim.py:4-19 -------------------------------------------------- */
int dfe_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, void* stream);
int dfe_ssim_bwd(const float* x, const float* y, const float* gout, float* gx, float* gy, int B, int C, int H, int W,
                 void* stream);

/* ---- PWC_tf.corr_naive(input1, input2, d=4)  pwc_tf.py:97-106 ------------------------------
 * out [B,(2d+1)^2,H,W]; only d == 4 is implemented (DFE_ERR_UNSUPPORTED otherwise). */
int dfe_corr_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int d, void* stream);
int dfe_corr_bwd(const float* f1, const float* f2, const float* gout, float* g1, float* g2, int B, int C, int H, int W,
                 int d, void* stream);

/* ---- pyramids: mode 0 = F.interpolate(bilinear, align_corners=False) (model_geometry.py:65-72),
 * mode 1 = F.interpolate(area) == adaptive_avg_pool2d (model_geometry.py:91, model_flow.py:58-64). */
int dfe_resize(const float* in, float* out, int planes, int inH, int inW, int outH, int outW, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFE_HIP_H */
