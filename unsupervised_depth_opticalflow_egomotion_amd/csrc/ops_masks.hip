// Mask decisions of the reference's per-method API as stand-alone element-wise kernels:
//   compute_occ_weight      core/networks/model_geometry.py:105-132
//   compute_texture_mask    core/networks/model_geometry.py:134-140
//   compute_dynamic_mask    core/networks/model_geometry.py:685-713  (decision + score; |rigid - flow| stays autograd)
// They call the SAME device functions as the fused stack (dfe_device.h: mean3_abs_diff, occ_weights,
// dyna_decision), so a mask decided through the per-method API is bit-identical to the fused stack's and to the
// reference's.  Pure streaming: 9-13 floats read, 1-4 written per pixel -> HBM-bound, one pixel
// per thread, coalesced planes.
#include "dfe_device.h"
#include "dfe_internal.h"

namespace dfe {

__global__ void k_occ_masks(const float* __restrict__ from_l, const float* __restrict__ tgt,
                            const float* __restrict__ from_r, float* __restrict__ occ_bwd,
                            float* __restrict__ occ_fwd, float* __restrict__ valid_bwd,
                            float* __restrict__ valid_fwd, int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const long o3 = static_cast<long>(b) * 3 * HW + p, o1 = static_cast<long>(b) * HW + p;
  const float l0 = from_l[o3], l1 = from_l[o3 + HW], l2 = from_l[o3 + 2L * HW];
  const float r0 = from_r[o3], r1 = from_r[o3 + HW], r2 = from_r[o3 + 2L * HW];
  const float t0 = tgt[o3], t1 = tgt[o3 + HW], t2 = tgt[o3 + 2L * HW];
  float wb, wf;
  occ_weights(mean3_abs_diff(t0, t1, t2, l0, l1, l2), mean3_abs_diff(t0, t1, t2, r0, r1, r2), wb, wf);
  occ_bwd[o1] = wb > 0.48f ? 1.0f : 0.0f;
  occ_fwd[o1] = wf > 0.48f ? 1.0f : 0.0f;
  valid_bwd[o1] = (l0 == 0.0f && l1 == 0.0f && l2 == 0.0f) ? 0.0f : 1.0f;
  valid_fwd[o1] = (r0 == 0.0f && r1 == 0.0f && r2 == 0.0f) ? 0.0f : 1.0f;
}

__global__ void k_texture_mask(const float* __restrict__ img, const float* __restrict__ warped,
                               const float* __restrict__ source, float* __restrict__ out, int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const long o3 = static_cast<long>(b) * 3 * HW + p;
  const float t0 = img[o3], t1 = img[o3 + HW], t2 = img[o3 + 2L * HW];
  const float e_rec = mean3_abs_diff(t0, t1, t2, warped[o3], warped[o3 + HW], warped[o3 + 2L * HW]);
  const float e_src = mean3_abs_diff(t0, t1, t2, source[o3], source[o3 + HW], source[o3 + 2L * HW]);
  out[static_cast<long>(b) * HW + p] = e_rec < e_src ? 1.0f : 0.0f;
}

__global__ void k_dynamic_mask(const float* __restrict__ flow, const float* __restrict__ rigid,
                               float* __restrict__ mask, float* __restrict__ score, float alpha, float beta, int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const long o2 = static_cast<long>(b) * 2 * HW + p;
  const float fu = flow[o2], fv = flow[o2 + HW], ru = rigid[o2], rv = rigid[o2 + HW];
  const float du = fabsf(ru - fu), dv = fabsf(rv - fv);
  mask[static_cast<long>(b) * HW + p] = dyna_decision(fu, fv, ru, rv, du, dv, alpha, beta) ? 1.0f : 0.0f;
  if (score) score[static_cast<long>(b) * HW + p] = 1.0f / (1e-4f + l2norm2(du, dv));
}

}  // namespace dfe

using namespace dfe;

#define DFE_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)

extern "C" {

int dfe_occ_masks(const float* from_l, const float* tgt, const float* from_r, float* occ_bwd, float* occ_fwd,
                  float* valid_bwd, float* valid_fwd, int B, int H, int W, void* stream) {
  DFE_REQUIRE(from_l && tgt && from_r && occ_bwd && occ_fwd && valid_bwd && valid_fwd, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, DFE_ERR_DIMS);
  const int HW = H * W;
  k_occ_masks<<<dim3((HW + 255) / 256, B), 256, 0, static_cast<hipStream_t>(stream)>>>(from_l, tgt, from_r, occ_bwd, occ_fwd, valid_bwd, valid_fwd, HW);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_texture_mask(const float* img, const float* warped, const float* source, float* out, int B, int H, int W,
                     void* stream) {
  DFE_REQUIRE(img && warped && source && out, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, DFE_ERR_DIMS);
  const int HW = H * W;
  k_texture_mask<<<dim3((HW + 255) / 256, B), 256, 0, static_cast<hipStream_t>(stream)>>>(img, warped, source, out, HW);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_dynamic_mask(const float* flow, const float* rigid, float* mask, float* score, float alpha, float beta, int B,
                     int H, int W, void* stream) {
  DFE_REQUIRE(flow && rigid && mask, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, DFE_ERR_DIMS);
  const int HW = H * W;
  k_dynamic_mask<<<dim3((HW + 255) / 256, B), 256, 0, static_cast<hipStream_t>(stream)>>>(flow, rigid, mask, score, alpha, beta, HW);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

}  // extern "C"
