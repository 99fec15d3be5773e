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
/* Bumped whenever an exported signature changes, an entry point is removed or the host side starts to rely on a new one
 * (2: round 5 changed dfe_wino_wgrad3x3 and removed dfe_thin_conv3x3 / dfe_cast_*; 3: round 6 added dfe_pwc_level_map_bytes /
 * _fwd_map / _bwd_map, which ops.py calls).  _lib.py compares the library's value with this header's. */
#define DFE_ABI_VERSION 3

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

/* ---- Adam update of many tensors in one launch  train.py:85-87 (torch.optim.Adam, default betas / eps) ---------------
 * table: device array of ntensors records {float* p, const float* g, float* m, float* v, int64 n} (5 x 8 bytes);
 * blockmap: device int32 pairs (tensor, chunk), one per block of dfe_adam_chunk() elements; bias_correction1 / 2 =
 * 1 - beta^t of the step count t.  m = m + (1-b1)(g-m); v = b2 v + (1-b2) g g; p -= (lr/c1) m / (sqrt(v)/sqrt(c2) + eps). */
int dfe_adam_chunk(void);
int dfe_adam_step(const void* table, const int* blockmap, int nblocks, double lr, double beta1, double beta2, double eps,
                  double bias_correction1, double bias_correction2, void* stream);
/* the same update with the step count on the device (a step replayed from a hipGraph: train.py --graph): one extra one-thread
 * launch increments *step_count (a device double) and forms the two bias-correction coefficients in coef (two device floats),
 * which the update kernel reads. */
int dfe_adam_step_dev(const void* table, const int* blockmap, int nblocks, double lr, double beta1, double beta2, double eps,
                      double* step_count, float* coef, void* stream);

/* ---- order-independent scatter-add workspace (csrc/dfe_scatter.h) -----------------------------
 * Adjoint of a bilinear gather with respect to the sampled tensor: contributions are scaled by a power of two, rounded
 * once to 64-bit integers, added with integer atomics (associative: no dependence on the order the atomics retire in)
 * and converted back in one pass -- bitwise reproducible, no float atomics anywhere in the library.  A scatter into
 * n floats needs a 16-byte aligned workspace of dfe_scatter_ws_bytes(n) bytes (64 + 8 n); the library zero-fills it. */
long dfe_scatter_ws_bytes(long n);

/* ---- warp_flow(x, flow, use_mask)  net_utils.py:16-54 --------------------------------------
 * x [B,C,H,W], flow [B,2,H,W] -> out [B,C,H,W].  Backward: gflow [B,2,H,W] (NULL to skip),
 * gx [B,C,H,W] (NULL to skip; every element written, no zero-fill needed) through the scatter workspace gx_ws
 * (dfe_scatter_ws_bytes(B*C*H*W) bytes; may be NULL when gx is);
 * gflow is written once per pixel from fixed-order partial sums.  Both are bitwise reproducible. */
int dfe_warp_flow_fwd(const float* x, const float* flow, float* out, int B, int C, int H, int W, int use_mask,
                      int align_corners, void* stream);
int dfe_warp_flow_bwd(const float* x, const float* flow, const float* gout, float* gflow, float* gx, void* gx_ws, int B,
                      int C, int H, int W, int use_mask, int align_corners, void* stream);

/* ---- inverse_warp2(img, depth, ref_depth, pose, intrinsics)  inverse_warp.py:263-303 -------
 * cams from dfe_prepare_cameras(pose[B,6], K, cams, B, 1, 1, {1}).  Outputs: projected image
 * [B,3,H,W], valid mask [B,1,H,W], projected depth [B,1,H,W], computed depth [B,1,H,W]
 * (the last three may be NULL).  Backward: g_depth [B,1,H,W], g_refdepth [B,1,H,W]
 * (NULL to skip; every element written, through the scatter workspace g_refdepth_ws of
 * dfe_scatter_ws_bytes(B*H*W) bytes), g_pose [B,6]; partials: workspace of
 * dfe_pose_partials_floats(B,H,W) floats. */
int dfe_pose_partials_floats(int B, int H, int W);
int dfe_inverse_warp2_fwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          float* out_img, float* out_valid, float* out_pdepth, float* out_cdepth, int B, int H, int W,
                          int align_corners, void* stream);
int dfe_inverse_warp2_bwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          const float* g_img, const float* g_pdepth, const float* g_cdepth, float* g_depth,
                          float* g_refdepth, void* g_refdepth_ws, float* g_pose, float* partials, int B, int H, int W,
                          int align_corners, void* stream);

/* ---- calculate_rigid_flow(depth, pose, intrinsics)  inverse_warp.py:311-342 ---------------- */
int dfe_rigid_flow_fwd(const float* depth, const float* cams, float* out, int B, int H, int W, void* stream);
int dfe_rigid_flow_bwd(const float* depth, const float* cams, const float* gout, float* g_depth, float* g_pose,
                       float* partials, int B, int H, int W, void* stream);

/* ---- mask decisions of the per-method API -----------------------------------------------------
 * compute_occ_weight (model_geometry.py:105-132): from_l/tgt/from_r [B,3,H,W] (masked flow warps and the
 *   target level) -> occ_bwd/occ_fwd = (1 - softmax([dl, dr]) > 0.48), valid_bwd/valid_fwd = 1 - prod_c(warp == 0),
 *   all [B,1,H,W] in {0,1};
 * compute_texture_mask (model_geometry.py:134-140): (mean_c|img - warped| < mean_c|img - source|) -> [B,1,H,W];
 * compute_dynamic_mask (model_geometry.py:699-711): flow, rigid [B,2,H,W] -> mask = (n(|rigid-flow|)^2 <
 *   alpha (n(flow)^2 + n(rigid)^2) + beta), score = 1 / (1e-4 + n(|rigid-flow|)) (score may be NULL).
 * Same device arithmetic as the fused stack: masks are bit-identical to dfe_geom_loss_fwd's mask pack. */
int dfe_occ_masks(const float* from_l, const float* tgt, const float* from_r, float* occ_bwd, float* occ_fwd,
                  float* valid_bwd, float* valid_fwd, int B, int H, int W, void* stream);
int dfe_texture_mask(const float* img, const float* warped, const float* source, float* out, int B, int H, int W,
                     void* stream);
int dfe_dynamic_mask(const float* flow, const float* rigid, float* mask, float* score, float alpha, float beta, int B,
                     int H, int W, void* stream);

/* ---- device-side input pipeline  core/dataset/kitti_prepared.py:63-90,132-152, train.py:171 ------------------
 * in_u8: uint8 [B][3*H0][W0][3] raw stacked triplets (left / target / right along H, channel order as stored);
 * flip: B device bytes (non-zero = horizontal flip of that sample, NULL = none); out: fp32 [B][3][3*H][W] in [0,1]:
 * per frame bilinear resize (half-pixel centres, edge replication) to H x W, flip, / 255, HWC -> CHW. */
int dfe_prepare_triplets(const unsigned char* in_u8, const unsigned char* flip, float* out, int B, int H0, int W0,
                         int H, int W, void* stream);

/* ---- forward-splat occlusion map  core/networks/model_flow.py:33-39 (get_occlusion_mask_from_flow) -----------
 * The reference calls an undefined `transformerFwd`; this is its TrianFlow meaning: out [B,1,H,W] = bilinear forward
 * warp of a ones image by flow [B,2,H,W] (optionally clamped to [0,1]).  Order-independent scatter through ws
 * (dfe_scatter_ws_bytes(B*H*W) bytes); every element of out is written; bitwise reproducible. */
int dfe_forward_splat_ones(const float* flow, float* out, void* ws, int B, int H, int W, int clamp01, void* stream);

/* ---- self-test of the short correctly rounded fp32 sequences (csrc/loss_stack_exact.h) ------------------------
 * The mask-deciding expressions of the reference use IEEE division and square root (torch CPU); the pointwise kernels
 * evaluate them with 3 / 5-instruction sequences that must return the SAME bits.  This entry point checks that
 * exhaustively on the device: counts (4 device uint64) receive the numbers of mismatches of
 *   [0] RN(1/z) over every fp32 z whose reciprocal is normal, [1] sqrt over every non-negative finite fp32,
 *   [2] x/z over `npairs` pseudo-random pairs, [3] the same with all-ones divisor significands.  All must be 0. */
int dfe_exact_math_selftest(unsigned long long* counts, unsigned long long npairs, void* stream);

/* ---- SSIM(x, y)  pytorch_ssim/ssim.py:4-19 -------------------------------------------------- */
int dfe_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, void* stream);
int dfe_ssim_bwd(const float* x, const float* y, const float* gout, float* gx, float* gy, int B, int C, int H, int W,
                 void* stream);

/* ---- PWC_tf.corr_naive(input1, input2, d=4)  pwc_tf.py:97-106 ------------------------------
 * out [B,(2d+1)^2,H,W]; only d == 4 is implemented (DFE_ERR_UNSUPPORTED otherwise). */
int dfe_corr_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int d, void* stream);
int dfe_corr_bwd(const float* f1, const float* f2, const float* gout, float* g1, float* g2, int B, int C, int H, int W,
                 int d, void* stream);

/* ---- ResNet stem max pooling  nn.MaxPool2d(3, 2, 1) of torchvision's resnet as ResnetEncoder runs it
 * (core/networks/structures/depth_model.py:60-95).  x [planes,H,W] -> y [planes,Ho,Wo], Ho = (H-1)/2+1; `idx` keeps the
 * winning window position (0..8, row-major) as one byte per output.  Values, tie-breaking (first maximum in window
 * scan order, NaN wins) and the gradient's accumulation order are ATen's: bit-identical to F.max_pool2d. */
int dfe_maxpool3x3s2_out(int n);
int dfe_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int planes, int H, int W, void* stream);
int dfe_maxpool3x3s2_bwd(const float* gy, const unsigned char* idx, float* gx, int planes, int H, int W, void* stream);

/* ---- one PWC decoder level's input  pwc_tf.py:119-121 (and :131-133, :143-145, :155-157) --------
 *   warp = self.warp(c2, up_flow); corr = self.corr(c1, warp); x = torch.cat((corr, c1, up_flow), 1)
 * as ONE operator: the cost volume is written straight into planes [0,81) of x [B, 81+C+2, H, W] (caller-allocated),
 * c1 and the flow are copied behind it; `warped` [B,C,H,W] is kept for the backward pass.  The backward pass reads
 * the three channel slices of gx = dL/dx in place and returns complete gradients:
 *   g_c1 = dcorr/dc1 + gx[:,81:81+C];  g_flow = dwarp/dflow + gx[:,81+C:];  g_c2 (order-independent scatter through
 *   g_c2_ws, dfe_scatter_ws_bytes(B*C*H*W) bytes; every element of g_c2 is written).
 * g_warped [B,C,H,W] is scratch.  g_c2 (with g_c2_ws) or g_flow may be NULL (not both). */
int dfe_pwc_level_channels(int C);   /* 81 + C + 2 */
int dfe_pwc_level_fwd(const float* c1, const float* c2, const float* flow, float* warped, float* x, int B, int C, int H,
                      int W, int align_corners, void* stream);
int dfe_pwc_level_bwd(const float* c1, const float* c2, const float* flow, const float* warped, const float* gx,
                      float* g_warped, float* g_c1, float* g_c2, void* g_c2_ws, float* g_flow, int B, int C, int H,
                      int W, int align_corners, void* stream);
/* The same operator with the backward's inverse map built in the FORWARD pass (ABI 3; C >= 8): `map` (dfe_pwc_level_map_bytes
 * bytes, 16-byte aligned, caller-allocated, contents undefined on entry) receives, per pixel of c2, the list of (source pixel,
 * bilinear weight) taps of warp(c2, flow) that read it -- the feature warp counts them while it samples -- and must reach
 * dfe_pwc_level_bwd_map unchanged.  That backward is three launches at every level (correlation gradients, flow gradient, gather)
 * and returns the bits dfe_pwc_level_bwd returns.  DFE_ERR_UNSUPPORTED for C < 8 (use the pair above). */
long dfe_pwc_level_map_bytes(int B, int H, int W);
int dfe_pwc_level_fwd_map(const float* c1, const float* c2, const float* flow, float* warped, float* x, void* map, int B, int C,
                          int H, int W, int align_corners, void* stream);
int dfe_pwc_level_bwd_map(const float* c1, const float* c2, const float* flow, const float* warped, const float* gx,
                          float* g_warped, float* g_c1, float* g_c2, void* map, float* g_flow, int B, int C, int H, int W,
                          int align_corners, void* stream);

/* ---- pyramids: mode 0 = F.interpolate(bilinear, align_corners=False) (model_geometry.py:65-72),
 * mode 1 = F.interpolate(area) == adaptive_avg_pool2d (model_geometry.py:91, model_flow.py:58-64). */
int dfe_resize(const float* in, float* out, int planes, int inH, int inW, int outH, int outW, int mode, void* stream);
/* Differentiable bilinear resize with the scalar PWC_tf puts on either side of F.interpolate (pwc_tf.py:118-119:
 * `F.interpolate(flow6, scale_factor=2.0, mode='bilinear') * 2.0`; :175-178: `F.interpolate(flow2 * 4.0, [h, w])`):
 * pre_scale = 1: out = resize(in * mult); 0: out = resize(in) * mult -- each rounded like the ATen (CPU) composition.
 * The backward pass is the adjoint as a gather per input element (no atomics, unlike ATen's upsample backward);
 * DFE_ERR_UNSUPPORTED when the up-sampling ratio exceeds 4 per axis. */
int dfe_resize_bilinear_fwd(const float* in, float* out, int planes, int inH, int inW, int outH, int outW, float mult,
                            int pre_scale, void* stream);
int dfe_resize_bilinear_bwd(const float* gout, float* gin, int planes, int inH, int inW, int outH, int outW, float mult,
                            int pre_scale, void* stream);

/* ---- depth-decoder glue between the MIOpen convolutions (depth_model.py:60-211:
 * Conv3x3 = ReflectionPad2d(1) + conv, ConvBlock = Conv3x3 + ELU, stage = ConvBlock, bilinear x2, cat(skip), ConvBlock).
 * The convolutions are called without their bias; x is such an output and ``bias`` [C] (or NULL) is added on read.
 * dfe_elu_pad:          out [B,C,H+2,W+2] = reflect_pad1(apply_elu ? elu(x + bias) : x + bias),  x [B,C,H,W]
 * dfe_elu_up2_cat_pad:  out [B,C1+C2,2h+2,2w+2] = reflect_pad1(cat(bilinear_x2(elu(x + bias)), skip)),
 *                       x [B,C1,h,w], skip [B,C2,2h,2w] or NULL with C2 = 0.
 * Backward: gout has the padded shape; gx = gradient wrt x (= wrt x + bias); gbias [C] (or NULL) = its sum over
 * (b,h,w), which needs ``partials`` = dfe_glue_partials_floats(B,C,H,W) floats of scratch (H,W of x);
 * gskip may be NULL (ignored when C2 = 0). */
long dfe_glue_partials_floats(int B, int C, int H, int W);
int dfe_elu_pad_fwd(const float* x, const float* bias, float* out, int B, int C, int H, int W, int apply_elu, void* stream);
int dfe_elu_pad_bwd(const float* x, const float* bias, const float* gout, float* gx, float* gbias, float* partials,
                    int B, int C, int H, int W, int apply_elu, void* stream);
int dfe_elu_up2_cat_pad_fwd(const float* x, const float* bias, const float* skip, float* out, int B, int C1, int C2,
                            int h, int w, void* stream);
int dfe_elu_up2_cat_pad_bwd(const float* x, const float* bias, const float* gout, float* gx, float* gskip, float* gbias,
                            float* partials, int B, int C1, int C2, int h, int w, void* stream);

/* ---- disparity head of the depth decoder (depth_model.py: dispconv = Conv3x3(C -> 1) + Sigmoid per output scale):
 * out [B,1,H,W] = sigmoid(conv3x3(p) + bias[0]) on the reflection-padded activation p [B,C,H+2,W+2], weight [1,C,3,3];
 * C must be a multiple of 16 (DFE_ERR_UNSUPPORTED otherwise).  Backward: gp [B,C,H+2,W+2] (every element written),
 * gweight [C*9] / gbias [1] (may both be NULL); partials: dfe_disp_head_partials_floats floats of scratch. */
long dfe_disp_head_partials_floats(int B, int C, int H, int W);
int dfe_disp_head_fwd(const float* p, const float* weight, const float* bias, float* out, int B, int C, int H, int W,
                      void* stream);
int dfe_disp_head_bwd(const float* p, const float* weight, const float* out, const float* gout, float* gp, float* gweight,
                      float* gbias, float* partials, int B, int C, int H, int W, void* stream);

/* ---- flow head  pwc_tf.py:39-40  predict_flow = Conv2d(C, 2, kernel_size=3, stride=1, padding=1, bias=True) ----------
 * The same rolling-window kernels with two output channels on the unpadded activation x [B,C,H,W] (zero padding is
 * virtual): out [B,2,H,W] = conv3x3(x, weight [2,C,3,3]) + bias [2].  MIOpen has no matrix to feed with two output
 * channels (4-5 TFLOP/s: 69 / 26 / 99 us forward / data / weight gradient at C = 96, 64x208, 8 images).
 * Backward: gx [B,C,H,W] (every element written), gweight [2*C*9] / gbias [2] (may both be NULL), fixed-order sums;
 * partials: dfe_flow_head_partials_floats floats of scratch.  C must be a multiple of 8 (DFE_ERR_UNSUPPORTED otherwise). */
long dfe_flow_head_partials_floats(int B, int C, int H, int W);
int dfe_flow_head_fwd(const float* x, const float* weight, const float* bias, float* out, int B, int C, int H, int W,
                      void* stream);
int dfe_flow_head_bwd(const float* x, const float* weight, const float* gout, float* gx, float* gweight, float* gbias,
                      float* partials, int B, int C, int H, int W, void* stream);

/* ---- weight gradient of the thin, wide 3x3 convolutions of the depth decoder on pre-padded inputs (fp32 MFMA):
 * gweight [Co,Ci,3,3] = sum_{b,y,x} gy[b,co,y,x] * p[b,ci,y+ky,x+kx],  p [B,Ci,H+2,W+2], gy [B,Co,H,W].
 * Requires Ci % 16 == 0, Co % 16 == 0, W % 16 == 0, p 8-byte and gy 16-byte aligned (DFE_ERR_UNSUPPORTED otherwise: the
 * caller then keeps MIOpen's weight gradient).  partials: dfe_wgrad3x3_partials_floats floats of scratch. */
long dfe_wgrad3x3_partials_floats(int B, int Ci, int Co, int H, int W);
int dfe_wgrad3x3_fwd(const float* p, const float* gy, float* gweight, float* partials, int B, int Ci, int Co, int H, int W,
                     void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolutions on small planes (H*W <= 4096 per sample) on the fp32 matrix cores: PWC's decoder
 * levels 6 and 5 (pwc_tf.py:28-47 conv6_0 .. conv5_4 = Conv2d(3x3, pad 1, bias) + LeakyReLU(0.1), called at
 * pwc_tf.py:113-135), where MIOpen's kernels are launch- and layout-bound (10-45 us forward, 26-126 us per weight gradient
 * for 0.1-0.4 GFLOP).  x [B,Ci,H,W], weight [Co,Ci,3,3], all fp32 NCHW contiguous.
 * dfe_planeconv_fwd:   y = act(conv(x, weight) + bias[co]) (bias may be NULL; act(v) = v > 0 ? v : slope * v, slope 1 = none)
 *                      written to dst1 and (optional) dst2: element (b,co,i) at dst + b * batch_stride + co*H*W + i, so the
 *                      output lands in the channel slices of the concatenated buffers that consume it.
 * dfe_planeconv_dgrad: gx [B,Ci,H,W] = the data gradient of conv(x, weight) for the output gradient gy [B,Co,H,W].
 * dfe_planeconv_wgrad: gweight [Co,Ci,3,3] = the weight gradient.
 * ws: dfe_planeconv_ws_floats floats of scratch (partial sums of the split reductions; every sum is added in a fixed order:
 * results are reproducible).  DFE_ERR_UNSUPPORTED (dfe_planeconv_supported == 0) for larger planes: the caller keeps MIOpen. */
int dfe_planeconv_supported(int B, int Ci, int Co, int H, int W);
long dfe_planeconv_ws_floats(int B, int Ci, int Co, int H, int W);
int dfe_planeconv_fwd(const float* x, const float* weight, const float* bias, float slope, float* dst1, long dst1_batch_stride,
                      float* dst2, long dst2_batch_stride, float* ws, int B, int Ci, int Co, int H, int W, void* stream);
int dfe_planeconv_dgrad(const float* gy, const float* weight, float* gx, float* ws, int B, int Ci, int Co, int H, int W,
                        void* stream);
int dfe_planeconv_wgrad(const float* gy, const float* x, float* gweight, float* ws, int B, int Ci, int Co, int H, int W,
                        void* stream);

/* ---- 3x3 / stride 1 convolutions of the networks' large layers as Winograd F(2x2, 3x3) on the fp32 matrix cores, one fused
 * kernel (depth_model.py:60-211 ResNet encoder / decoder, pwc_tf.py:28-95 decoder and context network, feature_pyramid.py:7-36;
 * MIOpen runs the same algorithm on the vector ALU).  x [B,Ci,H,W]; y [B,Co,Ho,Wo], Ho = H + 2P - 2, Wo = W + 2P - 2, P = 1
 * (zero padding), 0 (valid) or 2 (full: the data gradient of a valid convolution); element (b,co,i) of y at
 * y + b * y_batch_stride + co * Ho*Wo + i.
 * transposed_weight = 0: weight [Co,Ci,3,3], y = conv(x, weight) (no bias).
 * transposed_weight = 1: weight [Ci,Co,3,3] is the FORWARD filter of a convolution whose output gradient is x: y is its data
 *                        gradient, i.e. conv(x, w') with w'[co][ci][ky][kx] = weight[ci][co][2-ky][2-kx] (P = 1 for a forward
 *                        P = 1, P = 2 for a forward P = 0).
 * wbuf: scratch, 16-byte aligned, wbuf_floats floats: at least dfe_wino_weight_floats(Ci, Co) (the transformed filters);
 * with dfe_wino_scratch_floats(B, Ci, Co, H, W, P) the planes that cannot fill the chip (few tiles, many channels) split their
 * input channels over several blocks whose partial outputs are added in a fixed order.
 * DFE_ERR_DIMS when B*Ci*H*W >= 2^30 (32-bit offsets). */
long dfe_wino_weight_floats(int Ci, int Co);
long dfe_wino_scratch_floats(int B, int Ci, int Co, int H, int W, int P);
int dfe_wino_conv3x3(const float* x, const float* weight, float* y, long y_batch_stride, float* wbuf, long wbuf_floats, int B, int Ci,
                     int Co, int H, int W, int P, int transposed_weight, void* stream);
/* weight gradient of the same convolutions in the Winograd domain (replaces aten::convolution_backward's weight output for the
 * reference's 3x3 stride-1 layers, depth_model.py:13-58,135-191, pwc_tf.py:28-95): gweight [Co,Ci,3,3] = d/dw of
 * conv(x [B,Ci,H,W], w, padding P in {0, 1}) for the output gradient gy; element (b,ci,i) of x at x + b * x_batch_stride +
 * ci * H*W + i, element (b,co,i) of gy at gy + b * gy_batch_stride + co * Ho*Wo + i.  Straight from NCHW: no layout
 * transposes, no zero fill, no atomics (fixed-order partial sums in ws: dfe_wino_wgrad_floats floats); any size and
 * channel count.  A dilated layer (pwc_tf.py:31-36) is this call on its dilation x dilation phase images (B d^2 samples of
 * H/d x W/d pixels, P = 1), which the caller gathers. */
long dfe_wino_wgrad_floats(int B, int Ci, int Co, int H, int W, int P);
/* tuning hook (process-wide, not for concurrent use): tile = 0 (by shape) / 11 / 12 / 21 forces the wave tile (16 MH x 16 NH
 * channels); blocks1 / blocks2 > 0: the grid-size targets of the 64- / 128-accumulator kernels; chunk = 8 / 12 tiles per
 * barrier (0: keep).  The workspace size follows the setting: query dfe_wino_wgrad_floats after changing it. */
int dfe_wino_wgrad_tune(int tile, int blocks1, int blocks2, int chunk);
int dfe_wino_wgrad3x3(const float* x, long x_batch_stride, const float* gy, long gy_batch_stride, float* gweight, float* ws, int B,
                      int Ci, int Co, int H, int W, int P, void* stream);
/* Weight gradient of a STRIDED convolution straight from NCHW on the fp32 matrix cores (csrc/ops_sconv.hip; replaces MIOpen's
 * transposed NHWC calls for nn.Conv2d(..., stride=2): depth_model.py:60-95 ResNet stem, feature_pyramid.py:7-36, pose_cnn.py:14-36).
 * x [B,Ci,H,W], gy [B,Co,Ho,Wo] (Ho = (H + 2P - K) / stride + 1), both with a batch stride in floats; K x K filters, K <= 7.
 * dfe_sconv_wgrad: gweight [Co,Ci,K,K] = the weight gradient; ws holds dfe_sconv_wgrad_floats(...) floats of split partials,
 * added in split order (no atomics: bit-reproducible).  0 floats / DFE_ERR_UNSUPPORTED: a shape outside the kernel family
 * (3x3 with Ci >= 16; Ci K K <= 160; Ci K K <= 448 with Co <= 32).  dfe_sconv_tune: blocks >= 0 the grid-size target (0: two or
 * three resident blocks per CU, by kernel); rows >= 0 caps the output rows per chunk (0: no cap); negative: keep. */
int dfe_sconv_tune(int blocks, int rows);
long dfe_sconv_wgrad_floats(int B, int Ci, int Co, int H, int W, int K, int stride, int P);
int dfe_sconv_wgrad(const float* x, long x_batch_stride, const float* gy, long gy_batch_stride, float* gweight, float* ws, int B,
                    int Ci, int Co, int H, int W, int K, int stride, int P, void* stream);
/* the same for a DILATED 3x3 convolution with padding = dilation (pwc_tf.py:31-36 context network: dilation 2, 4, 8, 16): the
 * Winograd tiles live on the dilation x dilation phase images; H and W must be multiples of the dilation.  y has x's size. */
int dfe_wino_conv3x3_dilated(const float* x, const float* weight, float* y, long y_batch_stride, float* wbuf, int B, int Ci, int Co,
                             int H, int W, int dilation, int transposed_weight, void* stream);
/* Transformed filters kept across calls (a training step transforms every filter twice -- forward and data gradient -- although
 * it only changes once, in the optimiser step; nn.Conv2d has no such state, depth_model.py:60-211 / pwc_tf.py:28-95):
 * dfe_wino_transform_weights_multi transforms n filters in ONE launch, bit-identically to what dfe_wino_conv3x3 computes for
 * itself.  table (device): 6 longs per filter = {weight pointer, U pointer (dfe_wino_weight_floats(Ci, Co) floats, 16-byte
 * aligned), K, C, transposed_weight, first block}, where (K, C) = (Co, Ci) of the convolution the U is FOR (transposed_weight = 1:
 * weight is [C,K,3,3]); a filter takes dfe_wino_transform_blocks(Ci = C, Co = K) consecutive blocks; blockmap (device): the
 * filter index of each of the n_blocks blocks.
 * dfe_wino_conv3x3_u is dfe_wino_conv3x3 / _dilated (dilation > 1: P is ignored) on such a U; part: part_floats floats of
 * scratch for the channel splits (dfe_wino_scratch_floats - dfe_wino_weight_floats; may be null / 0: no splits). */
long dfe_wino_transform_blocks(int Ci, int Co);
int dfe_wino_transform_weights_multi(const long* table, const int* blockmap, int n_blocks, void* stream);
int dfe_wino_conv3x3_u(const float* x, const float* U, float* y, long y_batch_stride, float* part, long part_floats, int B, int Ci,
                       int Co, int H, int W, int P, int dilation, void* stream);
/* dfe_wino_conv3x3_u with the bias + activation epilogue inside the output transform (net_utils.py:7-11 conv() = Conv2d +
 * LeakyReLU(0.1); pwc_tf.py:113-118): y = act(conv(x) + bias[co]), act(v) = v > 0 ? v : slope * v (slope 1: bias only; bias may
 * be null) -- bit for bit what dfe_bias_act_fwd makes of dfe_wino_conv3x3_u's output -- written to y and, when y2 is not null,
 * to the same (b, co, i) of y2 (batch stride y2_batch_stride): the two concatenated buffers a PWC decoder layer feeds. */
int dfe_wino_conv3x3_u_act(const float* x, const float* U, const float* bias, float slope, float* y, long y_batch_stride, float* y2,
                           long y2_batch_stride, float* part, long part_floats, int B, int Ci, int Co, int H, int W, int P,
                           int dilation, void* stream);

/* ---- 1x1 convolutions on tiny planes (H*W <= 256, B*H*W <= 4096): PoseCNN's pose_conv and refinement head
 * (pose_cnn.py:32,43,48: Conv2d(256 | 24 | 12, 12, 1) on 2x7 planes).  x [B,Ci,H,W], weight [Co,Ci] (= [Co,Ci,1,1]).
 * dfe_conv1x1_small_fwd: y = act(conv1x1(x, weight) + bias[co]) (bias may be NULL; act as in dfe_planeconv_fwd).
 * dfe_conv1x1_small_bwd: gx [B,Ci,H,W] and / or gweight [Co,Ci] (either may be NULL) for the output gradient gy [B,Co,H,W]
 * (of the pre-activation: apply dfe_bias_act_bwd first).  One thread per output, serial sums: reproducible. */
int dfe_conv1x1_small_supported(int B, int Ci, int Co, int H, int W);
int dfe_conv1x1_small_fwd(const float* x, const float* weight, const float* bias, float slope, float* y, int B, int Ci, int Co,
                          int H, int W, void* stream);
int dfe_conv1x1_small_bwd(const float* gy, const float* x, const float* weight, float* gx, float* gweight, int B, int Ci, int Co,
                          int H, int W, void* stream);

/* ---- grouped training-mode BatchNorm2d (+ residual + ReLU) of the depth encoder (SURVEY.md 8(f) rank 1;
 * depth_model.py:60-95 = torchvision BasicBlock conv-bn-relu-conv-bn-(+identity)-relu; model_geometry.py:786-788 calls the
 * depth net once per frame).  x [G*Bg,C,H,W] is G groups of Bg consecutive samples: statistics are per (group, channel)
 * and running_mean / running_var (may be NULL) receive the G momentum updates in group order, i.e. what G sequential
 * nn.BatchNorm2d calls on the groups do.  y = act((x - mean) * invstd * weight + bias [+ residual]), act = ReLU if relu.
 * save_mean / save_invstd: [G*C] outputs for the backward pass; partials: dfe_bn_partials_floats floats of scratch.
 * Backward: g' = gy masked by y > 0 when relu; gx, gres (= g', may be NULL), gweight / gbias [C] (may be NULL);
 * scratch_means: 2*G*C floats. */
long dfe_bn_partials_floats(int G, int Bg, int C, int H, int W);
int dfe_bn_fwd(const float* x, const float* residual, const float* weight, const float* bias, float* running_mean,
               float* running_var, float* y, float* save_mean, float* save_invstd, float* partials, int G, int Bg, int C,
               int H, int W, float eps, float momentum, int relu, void* stream);
int dfe_bn_bwd(const float* x, const float* y, const float* gy, const float* weight, const float* save_mean,
               const float* save_invstd, float* gx, float* gres, float* gweight, float* gbias, float* partials,
               float* scratch_means, int G, int Bg, int C, int H, int W, int relu, void* stream);

/* ---- convolution epilogue of the flow nets (net_utils.py conv() = Conv2d(bias) + LeakyReLU(0.1),
 * feature_pyramid.py:7-36, pwc_tf.py:16-95): the convolution itself runs on MIOpen *without* its bias, then
 * dfe_bias_act_fwd: z [B,C,H,W] <- act(z + bias[c]) in place; act(v) = v > 0 ? v : slope*v (0.1 LeakyReLU, 0 ReLU, 1 none);
 *                   bias may be NULL.
 * dfe_bias_act_bwd: gz = gy * act'(y) (decided on the sign of the output y), gbias[c] = sum_{b,h,w} gz (gbias may be NULL).
 *                   gy may be a channel slice of a wider tensor: element (b,c,i) at gy + b*gy_batch_stride + c*H*W + i.
 *                   partials: dfe_bias_act_partials_floats(B,C,H,W) floats of scratch (needed when gbias != NULL). */
long dfe_bias_act_partials_floats(int B, int C, int H, int W);
int dfe_bias_act_fwd(float* z, const float* bias, int B, int C, int H, int W, float slope, void* stream);
int dfe_bias_act_bwd(const float* y, const float* gy, long gy_batch_stride, float* gz, float* gbias, float* partials,
                     int B, int C, int H, int W, float slope, void* stream);

/* The same epilogue inside a DenseNet-style block (PWC_tf's decoder levels, pwc_tf.py:113-117 and the same five lines
 * per level: x2 = conv(cat(x0, x1)), x3 = conv(cat(x1, x2)) ...): act(z + bias) is written straight into the channel
 * slices of the concatenated buffers that consume it (dst2 optional; dst1 may be z), and the backward pass sums the
 * matching slices of the consumers' input gradients (g2 optional) and reads y from its slice.  Every slice is a base
 * pointer + a batch stride in floats; channel planes are contiguous inside a sample. */
int dfe_bias_act_fwd2(const float* z, const float* bias, float* dst1, long dst1_batch_stride, float* dst2,
                      long dst2_batch_stride, int B, int C, int H, int W, float slope, void* stream);
int dfe_bias_act_bwd2(const float* y, long y_batch_stride, const float* g1, long g1_batch_stride, const float* g2,
                      long g2_batch_stride, float* gz, float* gbias, float* partials, int B, int C, int H, int W,
                      float slope, void* stream);
/* With gbias == NULL and partials != NULL dfe_bias_act_bwd / dfe_bias_act_bwd2 only write the per-block partial sums; the
 * bias gradients of up to 8 such layers of one plane size (a PWC decoder level: pwc_tf.py:113-117) are then finished by
 * ONE launch: host arrays of n device pointers / channel counts; same summation order as the single-layer finish. */
int dfe_bias_grad_final_multi(const float* const* partials, float* const* gbias, const int* C, int n, int B, int H, int W,
                              void* stream);

/* ---- fused loss stack: everything from model_geometry.py:797 to :951 given the nets' outputs ---
 * One call computes the active loss_pack vectors of Model_geometry.forward (mode 0) for a batch:
 * pyramids (:65-72,:91), rigid view synthesis (:80-103), texture / occlusion / validity / dynamic
 * masks (:105-140,:685-713), flow warps (:74-78), masked L1 (:143-153), SSIM (:212-223),
 * smoothness (:225-279), flow consistency (:195-210), depth-flow consistency (:716-732) and the
 * epipolar distance (:355-418).  The backward call consumes d(total)/d(loss vectors) and writes
 * gradients wrt disparities, flows and pose (recompute-in-backward; only the 1-byte mask pack and
 * the masked warped images are kept in the workspace).
 *
 * loss rows of `losses` / `grad_losses` ([DFE_NUM_LOSSES][B], fp32): */
#define DFE_LOSS_DEPTH_PIXEL 0
#define DFE_LOSS_DEPTH_SMOOTH 1
#define DFE_LOSS_FLOW_PIXEL 2
#define DFE_LOSS_FLOW_SSIM 3
#define DFE_LOSS_FLOW_SMOOTH 4
#define DFE_LOSS_FLOW_CONSIS 5
#define DFE_LOSS_DEPTH_FLOW_CONSIS 6
#define DFE_LOSS_EPIPOLAR 7
/* the two depth terms the reference ships commented out (model_geometry.py:889-891,897-899):
 * rows 8 / 9 are written (and differentiated) only when the matching bit of dfe_geom_args.depth_terms is set, 0 otherwise */
#define DFE_LOSS_DEPTH_SSIM 8
#define DFE_LOSS_DEPTH_CONSIS 9
#define DFE_NUM_LOSSES 10
#define DFE_DEPTH_TERM_SSIM 1    /* compute_ssim_loss(img_list, reconstructed_imgs_from_{l,r}, {bwd,fwd}_mask_texture) */
#define DFE_DEPTH_TERM_CONSIS 2  /* compute_consis_loss(predicted_depths_to_{l,r}, computed_depths_to_{l,r}, ..._mask_texture), model_geometry.py:182-193 */

/* mask pack: one byte per pixel of every scale, [scale][B][Hs*Ws] */
#define DFE_MASK_VALID_BWD 0x01
#define DFE_MASK_VALID_FWD 0x02
#define DFE_MASK_OCC_BWD 0x04
#define DFE_MASK_OCC_FWD 0x08
#define DFE_MASK_DYNA_BWD 0x10
#define DFE_MASK_DYNA_FWD 0x20
#define DFE_MASK_TEX_BWD 0x40
#define DFE_MASK_TEX_FWD 0x80

typedef struct dfe_geom_args {
  int B, H, W;               /* batch, full-resolution frame size */
  int num_scales;            /* S <= DFE_MAX_SCALES; scale s is int(H/2^s) x int(W/2^s) */
  int align_corners;
  int mode;                  /* 0 = Model_geometry loss stack; 1 = Model_depth loss stack (model_depth.py:272-337:
                                depth pixel + smoothness terms; flow / K_inv pointers are ignored); 2 = Model_flow
                                loss stack (model_flow.py:209-261: box-mean pyramids, soft occlusion weights;
                                disp / pose / K / K_inv pointers are ignored) */
  float alpha, beta;         /* flow_consist_alpha / flow_consist_beta (model_geometry.py:25-26) */
  const float* img[3];       /* left, target, right frames [B,3,H,W] */
  const float* disp[3][DFE_MAX_SCALES]; /* depth_net outputs per frame (l,t,r) and scale [B,1,Hs,Ws] */
  const float* flow[2][DFE_MAX_SCALES]; /* pwc flows, dir 0 = target->left (bwd), 1 = target->right (fwd), [B,2,Hs,Ws] */
  const float* pose;         /* [B,2,6]; index 0 = bwd, 1 = fwd (model_geometry.py:789-790) */
  const float* K;            /* [B,3,3] scale-0 intrinsics */
  const float* K_inv;        /* [B,3,3] */
  float* workspace;          /* dfe_geom_workspace_floats(args) floats, kept from forward to backward */
  long workspace_floats;
  float* losses;             /* forward out */
  const float* grad_losses;  /* backward in */
  float* grad_disp[3][DFE_MAX_SCALES]; /* backward out, same shapes as disp (NULL = skip) */
  float* grad_flow[2][DFE_MAX_SCALES]; /* backward out, same shapes as flow (NULL = skip) */
  float* grad_pose;          /* backward out [B,2,6] (NULL = skip) */
  int depth_terms;           /* modes 0 and 1: DFE_DEPTH_TERM_* bits; 0 = the reference as shipped (mode 1 = the same two
                                lines of Model_depth, model_depth.py:326-327,332-333: mask = validity x texture, the
                                consistency term unmasked).  With
                                DFE_DEPTH_TERM_CONSIS the SOURCE disparities grad_disp[0], grad_disp[2] also receive the
                                gradient of the projected depth (a bilinear scatter: float atomics, reproducible to
                                rounding only); everything else stays bitwise reproducible. */
} dfe_geom_args;

long dfe_geom_workspace_floats(const dfe_geom_args* args);
/* byte offset of the mask pack inside the workspace, and its size in bytes per scale start */
long dfe_geom_maskpack_offset_bytes(const dfe_geom_args* args, int scale);
int dfe_geom_loss_fwd(const dfe_geom_args* args, void* stream);
int dfe_geom_loss_bwd(const dfe_geom_args* args, void* stream);

/* Diagnostic variants used by bench.py: identical launches with a hipEvent recorded on `stream`
 * between them; they synchronise on the last event and return the per-segment durations in
 * milliseconds in `ms_host` (host array).  Forward segments: 0 cameras+epipolar prep, 1 pyramids,
 * 2 k_geom_point_fwd (warp stage), 3 k_geom_ssim_fwd, 4 flow smoothness, 5 disparity smoothness,
 * 6 finalize.  Backward segments: 0 k_geom_ssim_bwd, 1 k_geom_point_bwd, 2 flow smoothness,
 * 3 disparity smoothness stage 1, 4 stage 2, 5 pose finalize. */
#define DFE_GEOM_FWD_SEGMENTS 7
#define DFE_GEOM_BWD_SEGMENTS 6
int dfe_geom_loss_fwd_profiled(const dfe_geom_args* args, void* stream, float* ms_host);
int dfe_geom_loss_bwd_profiled(const dfe_geom_args* args, void* stream, float* ms_host);

/* Deferred read-out of the same events: the launches and the event records are enqueued on `stream` with no host
 * synchronisation (usable inside a running training step); *handle owns the events.  dfe_geom_timed_collect waits
 * for the last event of that call, writes the segment durations (7 forward / 6 backward floats) and frees the
 * handle; every handle must be collected exactly once. */
int dfe_geom_loss_fwd_timed(const dfe_geom_args* args, void* stream, void** handle);
int dfe_geom_loss_bwd_timed(const dfe_geom_args* args, void* stream, void** handle);
int dfe_geom_timed_collect(void* handle, float* ms_host);

#ifdef __cplusplus
}
#endif
#endif /* DFE_HIP_H */
