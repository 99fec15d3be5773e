// Disparity head of the depth decoder (reference depth_model.py: dispconv = Conv3x3(C -> 1) + Sigmoid on every output
// scale): disp = sigmoid(conv3x3(p) + bias) on the reflection-padded activation p [B,C,H+2,W+2].
//
// With a single output channel there is no matrix to feed the matrix cores with: MIOpen spends 0.28 / 0.13 / 0.43 ms
// (forward / data gradient / weight gradient at C = 16, 256x832, 12 images) on 44 us worth of HBM traffic.  Here the head
// is a rolling-window wave kernel like the loss stack's stencils: a wave owns a strip of 62 columns and marches down
// the rows; the three horizontal taps come from DPP wave shifts of ONE coalesced load per (channel, row), the three
// vertical taps from rotating accumulators (forward) or a 3-row register window of the 1-channel gradient (backward).
//   forward : out = sigmoid(sum_{c,ky,kx} w[c][ky][kx] * p[c][y+ky][x+kx] + bias)                    reads p once
//   backward: g = gout * out * (1 - out);  gp[c][r][q] = sum_{ky,kx} w[c][ky][kx] * g[r-ky][q-kx];
//             gw[c][ky][kx] = sum_{r,q} p[c][r][q] * g[r-ky][q-kx];  gb = sum g       reads p once, writes gp once
// The weight / bias gradients are per-wave partial sums finished in a fixed order (no atomics).  Bound: HBM.
#include "dfe_internal.h"
#include "dfe_device.h"
#include <hip/hip_runtime.h>

namespace dfe {

constexpr int DH_COLS = 62;       // columns owned by a wave (64 lanes minus the 2-column halo of the 3 taps)
constexpr int DH_ROWS = 16;       // rows marched by a wave
constexpr int DH_CI = 16;         // channels per register chunk (forward) / per block (backward, grid.z)
constexpr int DH_NACC = DH_CI * 9 + 1;

// grid: x = strip + nstrips * rowblock, y = b; block = one wave
__global__ void __launch_bounds__(64) k_disp_head_fwd(const float* __restrict__ p, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int C, int H,
                                                      int W, int nstrips, int R) {
  const int lane = threadIdx.x, b = blockIdx.y;
  const int strip = blockIdx.x % nstrips, rb = blockIdx.x / nstrips;
  const int Hp = H + 2, Wp = W + 2;
  const int xx = strip * DH_COLS + lane;              // padded column loaded by this lane = output column it produces
  const bool ld = xx < Wp, st = lane < DH_COLS && xx < W;
  const int y0 = rb * R;
  const float bv = bias ? bias[0] : 0.0f;
  const long plane = static_cast<long>(Hp) * Wp;
  const float* pb = p + static_cast<long>(b) * C * plane + (ld ? xx : 0);
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;      // output rows r, r-1, r-2 while padded row r is consumed
  const int rend = min(y0 + R + 2, Hp);
  for (int r = y0; r < rend; ++r) {
    const float* pr = pb + static_cast<long>(r) * Wp;
    for (int c0 = 0; c0 < C; c0 += DH_CI) {
      float v[DH_CI];
#pragma unroll
      for (int k = 0; k < DH_CI; ++k) v[k] = ld ? pr[(c0 + k) * plane] : 0.0f;
#pragma unroll
      for (int k = 0; k < DH_CI; ++k) {
        const float v1 = wave_shl1(v[k]), v2 = wave_shl1(v1);
        const float* wp = w + (c0 + k) * 9;
        acc0 = fmaf(wp[0], v[k], acc0); acc0 = fmaf(wp[1], v1, acc0); acc0 = fmaf(wp[2], v2, acc0);
        acc1 = fmaf(wp[3], v[k], acc1); acc1 = fmaf(wp[4], v1, acc1); acc1 = fmaf(wp[5], v2, acc1);
        acc2 = fmaf(wp[6], v[k], acc2); acc2 = fmaf(wp[7], v1, acc2); acc2 = fmaf(wp[8], v2, acc2);
      }
    }
    const int yo = r - 2;
    if (st && yo >= y0 && yo < H) out[(static_cast<long>(b) * H + yo) * W + xx] = 1.0f / (1.0f + __expf(-(acc2 + bv)));
    acc2 = acc1; acc1 = acc0; acc0 = 0.0f;
  }
}

// 1-channel gradient before the sigmoid at (y, x), zero outside the image
__device__ __forceinline__ float dh_grad(const float* __restrict__ gout, const float* __restrict__ out, int y, int x, int H, int W) {
  if (y < 0 || y >= H || x < 0 || x >= W) return 0.0f;
  const long o = static_cast<long>(y) * W + x;
  const float d = out[o];
  return gout[o] * (d * (1.0f - d));
}

// grid: x = strip + nstrips * rowblock (rows of the PADDED plane), y = b, z = channel chunk; block = one wave
__global__ void __launch_bounds__(64) k_disp_head_bwd(const float* __restrict__ p, const float* __restrict__ w,
                                                      const float* __restrict__ out, const float* __restrict__ gout,
                                                      float* __restrict__ gp, float* __restrict__ part, int C, int H, int W,
                                                      int nstrips) {
  const int lane = threadIdx.x, b = blockIdx.y, c0 = blockIdx.z * DH_CI;
  const int strip = blockIdx.x % nstrips, rb = blockIdx.x / nstrips;
  const int Hp = H + 2, Wp = W + 2;
  const int xx = strip * DH_COLS + lane - 2;          // padded column owned by lanes 2..63 (lanes 0, 1: left halo)
  const bool own = lane >= 2 && xx < Wp;
  const int r0 = rb * DH_ROWS, rend = min(r0 + DH_ROWS, Hp);
  const float* go = gout + static_cast<long>(b) * H * W;
  const float* oo = out + static_cast<long>(b) * H * W;
  const long plane = static_cast<long>(Hp) * Wp;
  const long base = (static_cast<long>(b) * C + c0) * plane + (own ? xx : 0);
  float acc[DH_CI][9];
#pragma unroll
  for (int k = 0; k < DH_CI; ++k)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[k][t] = 0.0f;
  float accb = 0.0f;
  // G[ky][kx] = g[r - ky][xx - kx]: rows r (A), r-1 (B), r-2 (C); columns by wave_shr shifts
  float gB0 = dh_grad(go, oo, r0 - 1, xx, H, W), gC0 = dh_grad(go, oo, r0 - 2, xx, H, W);
  float gB1 = wave_shr1(gB0), gB2 = wave_shr1(gB1), gC1 = wave_shr1(gC0), gC2 = wave_shr1(gC1);
  for (int r = r0; r < rend; ++r) {
    const float gA0 = dh_grad(go, oo, r, xx, H, W);
    const float gA1 = wave_shr1(gA0), gA2 = wave_shr1(gA1);
    float pv[DH_CI];
#pragma unroll
    for (int k = 0; k < DH_CI; ++k) pv[k] = own ? p[base + k * plane + static_cast<long>(r) * Wp] : 0.0f;
#pragma unroll
    for (int k = 0; k < DH_CI; ++k) {
      const float* wp = w + (c0 + k) * 9;
      float s = wp[0] * gA0;
      s = fmaf(wp[1], gA1, s); s = fmaf(wp[2], gA2, s);
      s = fmaf(wp[3], gB0, s); s = fmaf(wp[4], gB1, s); s = fmaf(wp[5], gB2, s);
      s = fmaf(wp[6], gC0, s); s = fmaf(wp[7], gC1, s); s = fmaf(wp[8], gC2, s);
      if (own) gp[base + k * plane + static_cast<long>(r) * Wp] = s;
      acc[k][0] = fmaf(pv[k], gA0, acc[k][0]); acc[k][1] = fmaf(pv[k], gA1, acc[k][1]); acc[k][2] = fmaf(pv[k], gA2, acc[k][2]);
      acc[k][3] = fmaf(pv[k], gB0, acc[k][3]); acc[k][4] = fmaf(pv[k], gB1, acc[k][4]); acc[k][5] = fmaf(pv[k], gB2, acc[k][5]);
      acc[k][6] = fmaf(pv[k], gC0, acc[k][6]); acc[k][7] = fmaf(pv[k], gC1, acc[k][7]); acc[k][8] = fmaf(pv[k], gC2, acc[k][8]);
    }
    if (own) accb += gA0;
    gC0 = gB0; gC1 = gB1; gC2 = gB2; gB0 = gA0; gB1 = gA1; gB2 = gA2;
  }
  // wave sums of the 144 weight-gradient products + the bias term, in a fixed order
  float* po = part + ((static_cast<long>(b) * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * DH_NACC;
#pragma unroll
  for (int k = 0; k < DH_CI; ++k) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[k][t];
      v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
      const float r0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
      const float r1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
      const float r2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
      const float r3s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
      if (lane == 0) po[k * 9 + t] = (r0s + r1s) + (r2s + r3s);
    }
  }
  {
    float v = accb;
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    const float r0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    if (lane == 0) po[DH_CI * 9] = (r0s + r1s) + (r2s + r3s);
  }
}

// gw[c*9 + t] / gb: one wave per output value, lanes stride over the (b, block) partials, fixed butterfly.
// grid: x = C*9 + 1 (the last block finishes the bias gradient from the z = 0 chunks)
__global__ void __launch_bounds__(64) k_disp_head_final(const float* __restrict__ part, float* __restrict__ gw,
                                                        float* __restrict__ gb, int C, int nunits, int nz) {
  const int o = blockIdx.x, lane = threadIdx.x;
  const bool isb = o == C * 9;
  const int z = isb ? 0 : (o / 9) / DH_CI, idx = isb ? DH_CI * 9 : (o - z * DH_CI * 9);
  float s = 0.0f;
  for (int u = lane; u < nunits; u += 64) s += part[(static_cast<long>(u) * nz + z) * DH_NACC + idx];
  s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x141>(s); s = dpp_add<0x140>(s);
  const float r0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
  const float r1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16));
  const float r2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32));
  const float r3s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
  if (lane == 0) {
    const float t = (r0s + r1s) + (r2s + r3s);
    if (isb) { if (gb) gb[0] = t; } else if (gw) gw[o] = t;
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static int dh_dims(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H < 1 || W < 1 || B > 65535) return DFE_ERR_DIMS;
  if (C % DH_CI != 0 || C / DH_CI > 65535) return DFE_ERR_UNSUPPORTED;
  if ((static_cast<long>(H) + 2) * (W + 2) * C >= (1L << 31)) return DFE_ERR_DIMS;
  return DFE_OK;
}
static inline int dh_strips(int cols) { return (cols + DH_COLS - 1) / DH_COLS; }
static inline int dh_rowblocks(int rows) { return (rows + DH_ROWS - 1) / DH_ROWS; }

extern "C" long dfe_disp_head_partials_floats(int B, int C, int H, int W) {
  if (dh_dims(B, C, H, W) != DFE_OK) return 0;
  return static_cast<long>(B) * dh_strips(W + 2) * dh_rowblocks(H + 2) * (C / DH_CI) * DH_NACC;
}

extern "C" int dfe_disp_head_fwd(const float* p, const float* weight, const float* bias, float* out, int B, int C, int H, int W,
                                 void* stream) {
  if (!p || !weight || !out) return DFE_ERR_NULL;
  const int rc = dh_dims(B, C, H, W);
  if (rc != DFE_OK) return rc;
  const int ns = dh_strips(W);
  int R = DH_ROWS;                       // fewer rows per wave on small images: at least ~2048 waves (2-row halo per wave)
  while (R > 2 && static_cast<long>(B) * ns * ((H + R - 1) / R) < 2048) R /= 2;
  k_disp_head_fwd<<<dim3(ns * ((H + R - 1) / R), B), 64, 0, static_cast<hipStream_t>(stream)>>>(p, weight, bias, out, C, H, W, ns, R);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_disp_head_bwd(const float* p, const float* weight, const float* out, const float* gout, float* gp,
                                 float* gweight, float* gbias, float* partials, int B, int C, int H, int W, void* stream) {
  if (!p || !weight || !out || !gout || !gp || !partials) return DFE_ERR_NULL;
  const int rc = dh_dims(B, C, H, W);
  if (rc != DFE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ns = dh_strips(W + 2), nrb = dh_rowblocks(H + 2), nz = C / DH_CI;
  k_disp_head_bwd<<<dim3(ns * nrb, B, nz), 64, 0, st>>>(p, weight, out, gout, gp, partials, C, H, W, ns);
  DFE_LAUNCH_CHECK();
  if (gweight || gbias) {
    k_disp_head_final<<<C * 9 + 1, 64, 0, st>>>(partials, gweight, gbias, C, B * ns * nrb, nz);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
