// Device-side input pipeline: what KITTI_Prepared.__getitem__ does per sample on CPU workers
// (core/dataset/kitti_prepared.py:63-90,132-152) -- split the stacked triplet, resize each frame to the training
// size, optional horizontal flip, / 255, HWC -> CHW -- as one launch over the whole batch, fed with the raw uint8
// triplets (one H2D copy of 1/4 of the float bytes instead of a blocking float copy, train.py:171).
//   in  : uint8 [B][3*H0][W0][3]  (frames stacked along H, channel order as stored -- cv2.imread's BGR is kept)
//   out : fp32  [B][3][3*H][W]    in [0,1]
// Resize = bilinear with half-pixel centres and edge replication (cv2.INTER_LINEAR's geometry) evaluated in fp32;
// cv2's 8-bit path rounds through 11-bit fixed-point coefficients, which is not reproduced (differences <= 1/255).
// One thread per output pixel, three channels each: 12 source bytes gathered, 12 bytes written -> HBM-bound streaming.
#include "dfe_device.h"
#include "dfe_internal.h"

namespace dfe {

__global__ void __launch_bounds__(256) k_prepare_triplets(const unsigned char* __restrict__ in, const unsigned char* __restrict__ flip,
                                                          float* __restrict__ out, int B, int H0, int W0, int H, int W) {
  const long n = static_cast<long>(B) * 3 * H * W;
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = static_cast<int>(i % W), y = static_cast<int>((i / W) % H);
  const int f = static_cast<int>((i / (static_cast<long>(W) * H)) % 3), b = static_cast<int>(i / (static_cast<long>(W) * H * 3));
  const int xs = (flip && flip[b]) ? W - 1 - x : x;           // cv2.flip(img, 1) after the resize
  const float sy = static_cast<float>(H0) / H, sx = static_cast<float>(W0) / W;
  float fy = (y + 0.5f) * sy - 0.5f, fx = (xs + 0.5f) * sx - 0.5f;
  fy = fminf(fmaxf(fy, 0.0f), static_cast<float>(H0 - 1)); fx = fminf(fmaxf(fx, 0.0f), static_cast<float>(W0 - 1));
  const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
  const int y1 = min(y0 + 1, H0 - 1), x1 = min(x0 + 1, W0 - 1);
  const float wy = fy - y0, wx = fx - x0;
  const unsigned char* src = in + (static_cast<long>(b) * 3 + f) * H0 * W0 * 3;
  const unsigned char* p00 = src + (static_cast<long>(y0) * W0 + x0) * 3;
  const unsigned char* p01 = src + (static_cast<long>(y0) * W0 + x1) * 3;
  const unsigned char* p10 = src + (static_cast<long>(y1) * W0 + x0) * 3;
  const unsigned char* p11 = src + (static_cast<long>(y1) * W0 + x1) * 3;
  float* o = out + (static_cast<long>(b) * 3 * 3 * H + static_cast<long>(f) * H + y) * W + x;   // [b][c][f*H + y][x]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float top = p00[c] + (static_cast<float>(p01[c]) - p00[c]) * wx;
    const float bot = p10[c] + (static_cast<float>(p11[c]) - p10[c]) * wx;
    o[static_cast<long>(c) * 3 * H * W] = (top + (bot - top) * wy) * (1.0f / 255.0f);
  }
}

}  // namespace dfe

extern "C" int dfe_prepare_triplets(const unsigned char* in_u8, const unsigned char* flip, float* out, int B, int H0, int W0,
                                    int H, int W, void* stream) {
  if (!in_u8 || !out) return DFE_ERR_NULL;
  if (B <= 0 || H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  const long n = static_cast<long>(B) * 3 * H * W;
  dfe::k_prepare_triplets<<<static_cast<unsigned>((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(in_u8, flip, out, B, H0, W0, H, W);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}
