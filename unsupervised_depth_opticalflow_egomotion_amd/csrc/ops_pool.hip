// 3x3 / stride 2 / pad 1 max pooling of the ResNet stem (torchvision resnet.maxpool as the reference's ResnetEncoder
// uses it, core/networks/structures/depth_model.py:60-95) -- net glue, SURVEY.md 8(f) rank 1.
// ATen's kernels for this layer take 0.12 ms (forward) + 0.29 ms (backward, a gather over int64 indices) per training
// step at 12 x 64 x 128 x 416; both are pure HBM streams: 163 MB in / 41 MB out.
//   forward : rolling-window wave kernel.  A wave owns 64 output columns and marches down PL_ROWS output rows; per
//             input row one 8-byte load per lane (columns 2ox, 2ox+1), column 2ox-1 comes from the left lane by a
//             DPP wave shift, the odd input row shared by two output rows stays in registers: every input element
//             is loaded once.  Writes the maximum and a 1-byte window position (0..8) instead of an int64 index.
//   backward: a wave marches down the same rows and writes each INPUT row once as 8-byte stores: an input pixel
//             collects dL/dy of the <= 2x2 windows that cover it and whose recorded position is this pixel.
// Tie-breaking and accumulation order are ATen's (first maximum in row-major window order, NaN wins; window rows then
// columns ascending), so values and gradients are bit-identical to F.max_pool2d's.
#include "dfe_device.h"
#include "dfe_internal.h"
#include <cmath>

namespace dfe {

constexpr int PL_ROWS = 8;

struct Row3 { float v0, v1, v2; };   // input columns 2ox-1, 2ox, 2ox+1 of one input row (-inf outside the image)

__device__ __forceinline__ Row3 pool_row(const float* __restrict__ plane, int r, int H, int W, int ox, bool lane_live) {
  const float ninf = -INFINITY;
  Row3 o{ninf, ninf, ninf};
  const bool rok = lane_live && r >= 0 && r < H;
  const int c1 = 2 * ox;
  if (rok) {
    const float* row = plane + static_cast<long>(r) * W;
    if (c1 + 1 < W) { const PairF p = *reinterpret_cast<const PairF*>(row + c1); o.v1 = p.a; o.v2 = p.b; }
    else o.v1 = row[c1];
  }
  // column 2ox-1 is the left lane's column 2(ox-1)+1; the first lane of a wave fetches it itself
  const float left = wave_shr1(o.v2);
  o.v0 = left;
  if ((threadIdx.x & 63) == 0) o.v0 = (rok && c1 - 1 >= 0) ? plane[static_cast<long>(r) * W + c1 - 1] : ninf;
  if (ox == 0) o.v0 = ninf;
  return o;
}

__device__ __forceinline__ void pool_scan(const Row3& r, int base, float& best, int& bi, bool& seeded, int rr, int H, int ox, int W, int oy) {
  // ATen: maxidx starts at the first in-bounds element, maxval at -inf; (val > maxval) || isnan(val) replaces
  const int row = 2 * oy - 1 + rr;
  if (row < 0 || row >= H) return;
  const float v[3] = {r.v0, r.v1, r.v2};
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    const int col = 2 * ox - 1 + cc;
    if (col < 0 || col >= W) continue;
    if (!seeded) { bi = base + cc; seeded = true; }
    if (v[cc] > best || v[cc] != v[cc]) { best = v[cc]; bi = base + cc; }
  }
}

__global__ void __launch_bounds__(64) k_maxpool3x3s2_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                         unsigned char* __restrict__ idx, int H, int W, int Ho, int Wo,
                                                         int strips) {
  const int strip = blockIdx.x % strips, rb = blockIdx.x / strips;
  const long pl = blockIdx.y;
  const int ox = strip * 64 + static_cast<int>(threadIdx.x);
  const bool live = ox < Wo;
  const float* plane = x + pl * H * W;
  const int oy0 = rb * PL_ROWS, oy1 = min(oy0 + PL_ROWS, Ho);
  Row3 top = pool_row(plane, 2 * oy0 - 1, H, W, ox, live);
  for (int oy = oy0; oy < oy1; ++oy) {
    const Row3 mid = pool_row(plane, 2 * oy, H, W, ox, live);
    const Row3 bot = pool_row(plane, 2 * oy + 1, H, W, ox, live);
    float best = -INFINITY;
    int bi = 0;
    bool seeded = false;
    pool_scan(top, 0, best, bi, seeded, 0, H, ox, W, oy);
    pool_scan(mid, 3, best, bi, seeded, 1, H, ox, W, oy);
    pool_scan(bot, 6, best, bi, seeded, 2, H, ox, W, oy);
    if (live) {
      const long o = (pl * Ho + oy) * Wo + ox;
      y[o] = best;
      idx[o] = static_cast<unsigned char>(bi);
    }
    top = bot;
  }
}

struct GI { float g; int k; };   // dL/dy and recorded window position of one output element (k = -1: none)

__device__ __forceinline__ GI pool_gi(const float* __restrict__ gy, const unsigned char* __restrict__ idx, long plane_off,
                                      int oy, int ox, int Ho, int Wo) {
  GI r{0.0f, -1};
  if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
    const long o = plane_off + static_cast<long>(oy) * Wo + ox;
    r.g = gy[o]; r.k = idx[o];
  }
  return r;
}

__global__ void __launch_bounds__(64) k_maxpool3x3s2_bwd(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                         float* __restrict__ gx, int H, int W, int Ho, int Wo, int strips) {
  const int strip = blockIdx.x % strips, rb = blockIdx.x / strips;
  const long pl = blockIdx.y;
  const int ox = strip * 64 + static_cast<int>(threadIdx.x);
  const int c1 = 2 * ox, c2 = 2 * ox + 1;
  if (c1 >= W) return;
  const long po = pl * Ho * Wo;
  float* plane = gx + pl * H * W;
  const int oy0 = rb * PL_ROWS, oy1 = min(oy0 + PL_ROWS, Ho);
  GI pa = pool_gi(gy, idx, po, oy0 - 1, ox, Ho, Wo), pb = pool_gi(gy, idx, po, oy0 - 1, ox + 1, Ho, Wo);
  for (int oy = oy0; oy <= oy1; ++oy) {
    // this step writes input rows 2oy-1 (windows oy-1 at window row 2, oy at window row 0) and 2oy (window oy, row 1);
    // the step oy == oy1 only finishes the odd row below the block's last window row when no later block owns it
    const bool tail = oy == oy1;
    if (tail && oy1 < Ho) break;
    const GI ca = tail ? GI{0.0f, -1} : pool_gi(gy, idx, po, oy, ox, Ho, Wo);
    const GI cb = tail ? GI{0.0f, -1} : pool_gi(gy, idx, po, oy, ox + 1, Ho, Wo);
    const int ro = 2 * oy - 1;
    if (ro >= 0 && ro < H) {
      // even column c1: window column ox at window col 1; odd column c2: ox at col 2, ox+1 at col 0
      float e = 0.0f, o = 0.0f;
      if (pa.k == 2 * 3 + 1) e += pa.g;
      if (ca.k == 0 * 3 + 1) e += ca.g;
      if (pa.k == 2 * 3 + 2) o += pa.g;
      if (pb.k == 2 * 3 + 0) o += pb.g;
      if (ca.k == 0 * 3 + 2) o += ca.g;
      if (cb.k == 0 * 3 + 0) o += cb.g;
      float* row = plane + static_cast<long>(ro) * W;
      if (c2 < W) *reinterpret_cast<PairF*>(row + c1) = PairF{e, o}; else row[c1] = e;
    }
    const int re = 2 * oy;
    if (!tail && re < H) {
      float e = 0.0f, o = 0.0f;
      if (ca.k == 1 * 3 + 1) e += ca.g;
      if (ca.k == 1 * 3 + 2) o += ca.g;
      if (cb.k == 1 * 3 + 0) o += cb.g;
      float* row = plane + static_cast<long>(re) * W;
      if (c2 < W) *reinterpret_cast<PairF*>(row + c1) = PairF{e, o}; else row[c1] = e;
    }
    pa = ca; pb = cb;
  }
}

}  // namespace dfe

extern "C" int dfe_maxpool3x3s2_out(int n) { return (n - 1) / 2 + 1; }

extern "C" int dfe_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int planes, int H, int W, void* stream) {
  if (!x || !y || !idx) return DFE_ERR_NULL;
  if (planes <= 0 || planes > 65535 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  const int Ho = dfe_maxpool3x3s2_out(H), Wo = dfe_maxpool3x3s2_out(W);
  const int strips = (Wo + 63) / 64, rbs = (Ho + dfe::PL_ROWS - 1) / dfe::PL_ROWS;
  dfe::k_maxpool3x3s2_fwd<<<dim3(strips * rbs, planes), 64, 0, static_cast<hipStream_t>(stream)>>>(x, y, idx, H, W, Ho, Wo, strips);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}

extern "C" int dfe_maxpool3x3s2_bwd(const float* gy, const unsigned char* idx, float* gx, int planes, int H, int W, void* stream) {
  if (!gy || !idx || !gx) return DFE_ERR_NULL;
  if (planes <= 0 || planes > 65535 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  const int Ho = dfe_maxpool3x3s2_out(H), Wo = dfe_maxpool3x3s2_out(W);
  const int strips = (Wo + 63) / 64, rbs = (Ho + dfe::PL_ROWS - 1) / dfe::PL_ROWS;
  dfe::k_maxpool3x3s2_bwd<<<dim3(strips * rbs, planes), 64, 0, static_cast<hipStream_t>(stream)>>>(gy, idx, gx, H, W, Ho, Wo, strips);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}
