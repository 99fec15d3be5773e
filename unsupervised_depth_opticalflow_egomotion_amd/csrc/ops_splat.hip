// Forward-splat occlusion map: Model_flow.get_occlusion_mask_from_flow (core/networks/model_flow.py:33-39) calls
// `transformerFwd`, a symbol the reference never defines (TrianFlow / UnFlow heritage: bilinear forward warp of a ones
// image by the flow, clamped to [0,1]); the method is dead code there.  This makes it functional: every source pixel
// (x, y) deposits its four bilinear weights at (x + u, y + v); a target pixel that receives less than 1 is
// (partially) occluded.  The scatter adds 64-bit fixed-point integers (dfe_scatter.h: order-independent, bitwise
// reproducible; the weights are <= 1, so the scale is the constant 2^35); no oracle exists, property tests only.
// One thread per source pixel: 8 B read, up to 4 atomics -> atomic-rate bound, tiny.
#include "dfe_device.h"
#include "dfe_internal.h"
#include "dfe_scatter.h"

namespace dfe {

__global__ void __launch_bounds__(256) k_forward_splat_ones(const float* __restrict__ flow, void* __restrict__ ws, int H, int W) {
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const float* f = flow + static_cast<long>(b) * 2 * HW;
  const float tx = static_cast<float>(x) + f[p], ty = static_cast<float>(y) + f[HW + p];
  if (!(tx > -1.0f && tx < static_cast<float>(W) && ty > -1.0f && ty < static_cast<float>(H))) return;   // NaN / fully outside
  const float xf = floorf(tx), yf = floorf(ty);
  const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
  const float wx = tx - xf, wy = ty - yf;
  const ScatterScale sc = scatter_scale(*static_cast<const unsigned*>(ws));
  long long* o = scatter_acc(ws) + static_cast<long>(b) * HW;
  const bool xa = x0 >= 0, xb = x0 + 1 < W, ya = y0 >= 0, yb = y0 + 1 < H;
  if (xa && ya) fixed_add(o + y0 * W + x0, to_fixed((1.0f - wx) * (1.0f - wy), sc.to_fixed));
  if (xb && ya) fixed_add(o + y0 * W + x0 + 1, to_fixed(wx * (1.0f - wy), sc.to_fixed));
  if (xa && yb) fixed_add(o + (y0 + 1) * W + x0, to_fixed((1.0f - wx) * wy, sc.to_fixed));
  if (xb && yb) fixed_add(o + (y0 + 1) * W + x0 + 1, to_fixed(wx * wy, sc.to_fixed));
}

__global__ void __launch_bounds__(256) k_clamp01(float* __restrict__ v, long n) {
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) v[i] = fminf(fmaxf(v[i], 0.0f), 1.0f);
}

}  // namespace dfe

extern "C" int dfe_forward_splat_ones(const float* flow, float* out, void* ws, int B, int H, int W, int clamp01, void* stream) {
  if (!flow || !out || !ws) return DFE_ERR_NULL;
  if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long n = static_cast<long>(B) * H * W;
  int rc = dfe::scatter_begin_const(ws, n, 1.0f, st);
  if (rc != DFE_OK) return rc;
  dfe::k_forward_splat_ones<<<dim3((H * W + 255) / 256, B), 256, 0, st>>>(flow, ws, H, W);
  rc = dfe::scatter_finish(ws, out, n, st);
  if (rc != DFE_OK) return rc;
  if (clamp01) dfe::k_clamp01<<<static_cast<unsigned>((n + 255) / 256), 256, 0, st>>>(out, n);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}
