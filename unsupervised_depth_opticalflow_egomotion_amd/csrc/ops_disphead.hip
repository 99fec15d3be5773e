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
constexpr int DH_CI = 16;         // channels per register chunk (forward) / per block (backward, grid.z): disparity head
constexpr int FH_CI = 8;          // the same for the 2-channel flow head (twice the accumulators per channel)

// The kernels are shared by two heads (round 3):
//   disparity head  CO = 1, SIG = true,  VPAD = false: p is the reflection-padded activation [B,C,H+2,W+2]
//   flow head       CO = 2, SIG = false, VPAD = true : p is the activation itself [B,C,H,W]; the padded plane is virtual
//                   (zeros outside the image: Conv2d(padding=1)), pwc_tf.py:39-40 predict_flow = Conv2d(C, 2, 3, 1, 1)
// weights [CO][C][3][3] (the Conv2d layout).

template <bool VPAD>
__device__ __forceinline__ float head_load(const float* __restrict__ pc, int r, int xx, int H, int W, bool ld) {
  // pc: channel plane; (r, xx): padded coordinates
  if (VPAD) return (ld && r >= 1 && r <= H && xx >= 1 && xx <= W) ? pc[static_cast<long>(r - 1) * W + (xx - 1)] : 0.0f;
  return ld ? pc[static_cast<long>(r) * (W + 2) + xx] : 0.0f;
}

// grid: x = strip + nstrips * rowblock, y = b; block = one wave
template <int CO, bool SIG, bool VPAD, int CI>
__global__ void __launch_bounds__(64) k_head_fwd(const float* __restrict__ p, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ out, int C, int H,
                                                 int W, int nstrips, int R) {
  const int lane = threadIdx.x, b = blockIdx.y;
  const int strip = blockIdx.x % nstrips, rb = blockIdx.x / nstrips;
  const int Hp = H + 2, Wp = W + 2;
  const int xx = strip * DH_COLS + lane;              // padded column loaded by this lane = output column it produces
  const bool ld = xx < Wp, st = lane < DH_COLS && xx < W;
  const int y0 = rb * R;
  const long plane = VPAD ? static_cast<long>(H) * W : static_cast<long>(Hp) * Wp;
  const float* pb = p + static_cast<long>(b) * C * plane;
  float acc[CO][3];                                   // output rows r, r-1, r-2 while padded row r is consumed
#pragma unroll
  for (int co = 0; co < CO; ++co) { acc[co][0] = 0.0f; acc[co][1] = 0.0f; acc[co][2] = 0.0f; }
  const int rend = min(y0 + R + 2, Hp);
  for (int r = y0; r < rend; ++r) {
    for (int c0 = 0; c0 < C; c0 += CI) {
      float v[CI];
#pragma unroll
      for (int k = 0; k < CI; ++k) v[k] = head_load<VPAD>(pb + (c0 + k) * plane, r, xx, H, W, ld);
#pragma unroll
      for (int k = 0; k < CI; ++k) {
        const float v1 = wave_shl1(v[k]), v2 = wave_shl1(v1);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float* wp = w + (static_cast<long>(co) * C + c0 + k) * 9;
          acc[co][0] = fmaf(wp[0], v[k], acc[co][0]); acc[co][0] = fmaf(wp[1], v1, acc[co][0]); acc[co][0] = fmaf(wp[2], v2, acc[co][0]);
          acc[co][1] = fmaf(wp[3], v[k], acc[co][1]); acc[co][1] = fmaf(wp[4], v1, acc[co][1]); acc[co][1] = fmaf(wp[5], v2, acc[co][1]);
          acc[co][2] = fmaf(wp[6], v[k], acc[co][2]); acc[co][2] = fmaf(wp[7], v1, acc[co][2]); acc[co][2] = fmaf(wp[8], v2, acc[co][2]);
        }
      }
    }
    const int yo = r - 2;
    if (st && yo >= y0 && yo < H) {
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float z = acc[co][2] + (bias ? bias[co] : 0.0f);
        out[((static_cast<long>(b) * CO + co) * H + yo) * W + xx] = SIG ? 1.0f / (1.0f + __expf(-z)) : z;
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) { acc[co][2] = acc[co][1]; acc[co][1] = acc[co][0]; acc[co][0] = 0.0f; }
  }
}

// Forward with the channels spread over the waves of a block (flow head: its planes are small -- 64x208 down to 4x13 --
// so a wave that walks all C channels of its strip is a long serial chain on a mostly idle chip: 35-58 us).  Wave k of
// the block owns channels [k*CI, (k+1)*CI), marches the FP_R + 2 padded rows of the block's FP_R output rows, keeps its
// partial outputs in registers and the partials meet in LDS, added in wave order (fixed).
// grid: x = strip + nstrips * rowblock, y = b; block = C / CI waves (<= 16).
constexpr int FP_R = 4;

template <int CO, bool SIG, bool VPAD, int CI>
__global__ void __launch_bounds__(1024) k_head_fwd_par(const float* __restrict__ p, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int C, int H,
                                                       int W, int nstrips) {
  extern __shared__ float red[];                      // [nwaves][FP_R][CO][64]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int b = blockIdx.y;
  const int strip = blockIdx.x % nstrips, rb = blockIdx.x / nstrips;
  const int Hp = H + 2, Wp = W + 2;
  const int xx = strip * DH_COLS + lane;
  const bool ld = xx < Wp, st = lane < DH_COLS && xx < W;
  const int y0 = rb * FP_R, c0 = wv * CI;
  const long plane = VPAD ? static_cast<long>(H) * W : static_cast<long>(Hp) * Wp;
  const float* pb = p + (static_cast<long>(b) * C + c0) * plane;
  float acc[CO][3], res[FP_R][CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) { acc[co][0] = 0.0f; acc[co][1] = 0.0f; acc[co][2] = 0.0f; }
#pragma unroll
  for (int q = 0; q < FP_R; ++q)
#pragma unroll
    for (int co = 0; co < CO; ++co) res[q][co] = 0.0f;
#pragma unroll
  for (int rr = 0; rr < FP_R + 2; ++rr) {
    const int r = y0 + rr;
    if (r < Hp) {                                     // block-uniform
      float v[CI];
#pragma unroll
      for (int k = 0; k < CI; ++k) v[k] = head_load<VPAD>(pb + k * plane, r, xx, H, W, ld);
#pragma unroll
      for (int k = 0; k < CI; ++k) {
        const float v1 = wave_shl1(v[k]), v2 = wave_shl1(v1);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float* wp = w + (static_cast<long>(co) * C + c0 + k) * 9;
          acc[co][0] = fmaf(wp[0], v[k], acc[co][0]); acc[co][0] = fmaf(wp[1], v1, acc[co][0]); acc[co][0] = fmaf(wp[2], v2, acc[co][0]);
          acc[co][1] = fmaf(wp[3], v[k], acc[co][1]); acc[co][1] = fmaf(wp[4], v1, acc[co][1]); acc[co][1] = fmaf(wp[5], v2, acc[co][1]);
          acc[co][2] = fmaf(wp[6], v[k], acc[co][2]); acc[co][2] = fmaf(wp[7], v1, acc[co][2]); acc[co][2] = fmaf(wp[8], v2, acc[co][2]);
        }
      }
    }
    if (rr >= 2) {
#pragma unroll
      for (int co = 0; co < CO; ++co) res[rr - 2][co] = acc[co][2];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) { acc[co][2] = acc[co][1]; acc[co][1] = acc[co][0]; acc[co][0] = 0.0f; }
  }
#pragma unroll
  for (int q = 0; q < FP_R; ++q)
#pragma unroll
    for (int co = 0; co < CO; ++co) red[((wv * FP_R + q) * CO + co) * 64 + lane] = res[q][co];
  __syncthreads();
  // wave q of the first FP_R waves finishes output row y0 + q
  if (wv < FP_R) {
    const int yo = y0 + wv;
    if (st && yo < H) {
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        float z = red[((0 * FP_R + wv) * CO + co) * 64 + lane];
        for (int k = 1; k < nwv; ++k) z += red[((k * FP_R + wv) * CO + co) * 64 + lane];
        z += bias ? bias[co] : 0.0f;
        out[((static_cast<long>(b) * CO + co) * H + yo) * W + xx] = SIG ? 1.0f / (1.0f + __expf(-z)) : z;
      }
    }
  }
}

// gradient before the activation at (y, x) of one output plane, zero outside the image
template <bool SIG>
__device__ __forceinline__ float dh_grad(const float* __restrict__ gout, const float* __restrict__ out, int y, int x, int H, int W) {
  if (y < 0 || y >= H || x < 0 || x >= W) return 0.0f;
  const long o = static_cast<long>(y) * W + x;
  if (!SIG) return gout[o];
  const float d = out[o];
  return gout[o] * (d * (1.0f - d));
}

__device__ __forceinline__ float wave_total(float v) {      // fixed butterfly, valid in lane 0
  v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
  const float r0s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0s + r1s) + (r2s + r3s);
}

// grid: x = strip + nstrips * rowblock (rows of the PADDED plane), y = b, z = channel chunk; block = one wave
// partials [(b, block)][z][CI][CO][9] followed by [CO] bias terms
template <int CO, bool SIG, bool VPAD, int CI>
__global__ void __launch_bounds__(64) k_head_bwd(const float* __restrict__ p, const float* __restrict__ w,
                                                 const float* __restrict__ out, const float* __restrict__ gout,
                                                 float* __restrict__ gp, float* __restrict__ part, int C, int H, int W,
                                                 int nstrips) {
  constexpr int NACC = CI * CO * 9 + CO;
  const int lane = threadIdx.x, b = blockIdx.y, c0 = blockIdx.z * CI;
  const int strip = blockIdx.x % nstrips, rb = blockIdx.x / nstrips;
  const int Hp = H + 2, Wp = W + 2;
  const int xx = strip * DH_COLS + lane - 2;          // padded column owned by lanes 2..63 (lanes 0, 1: left halo)
  const bool own = lane >= 2 && xx < Wp;
  const int r0 = rb * DH_ROWS, rend = min(r0 + DH_ROWS, Hp);
  const long oplane = static_cast<long>(H) * W;
  const long plane = VPAD ? oplane : static_cast<long>(Hp) * Wp;
  const float* pb = p + (static_cast<long>(b) * C + c0) * plane;
  float* gb_ = gp + (static_cast<long>(b) * C + c0) * plane;
  float acc[CI][CO][9];
#pragma unroll
  for (int k = 0; k < CI; ++k)
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[k][co][t] = 0.0f;
  float accb[CO];
  // G[co][ky*3 + kx] = g[co][r - ky][xx - kx]: rows r (A), r-1 (B), r-2 (C); columns by wave_shr shifts
  float G[CO][9];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const float* go = gout + (static_cast<long>(b) * CO + co) * oplane;
    const float* oo = SIG ? out + (static_cast<long>(b) * CO + co) * oplane : nullptr;
    accb[co] = 0.0f;
    G[co][3] = dh_grad<SIG>(go, oo, r0 - 1, xx, H, W); G[co][6] = dh_grad<SIG>(go, oo, r0 - 2, xx, H, W);
    G[co][4] = wave_shr1(G[co][3]); G[co][5] = wave_shr1(G[co][4]); G[co][7] = wave_shr1(G[co][6]); G[co][8] = wave_shr1(G[co][7]);
  }
  // the input rows are loaded two rows ahead of their use (a wave is alone on its SIMD most of the time -- 190 VGPRs --
  // and a row's 288 FMAs are shorter than a round trip to memory: without the prefetch every row paid one)
  float pn1[CI], pn2[CI];
#pragma unroll
  for (int k = 0; k < CI; ++k) {
    pn1[k] = head_load<VPAD>(pb + k * plane, r0, xx, H, W, own);
    pn2[k] = head_load<VPAD>(pb + k * plane, r0 + 1 < rend ? r0 + 1 : r0, xx, H, W, own);
  }
  for (int r = r0; r < rend; ++r) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float* go = gout + (static_cast<long>(b) * CO + co) * oplane;
      const float* oo = SIG ? out + (static_cast<long>(b) * CO + co) * oplane : nullptr;
      G[co][0] = dh_grad<SIG>(go, oo, r, xx, H, W);
      G[co][1] = wave_shr1(G[co][0]); G[co][2] = wave_shr1(G[co][1]);
    }
    float pv[CI];
    const int rn = r + 2 < rend ? r + 2 : r0;
#pragma unroll
    for (int k = 0; k < CI; ++k) { pv[k] = pn1[k]; pn1[k] = pn2[k]; pn2[k] = head_load<VPAD>(pb + k * plane, rn, xx, H, W, own); }
    const bool wr = VPAD ? (own && r >= 1 && r <= H && xx >= 1 && xx <= W) : own;
#pragma unroll
    for (int k = 0; k < CI; ++k) {
      float s = 0.0f;
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float* wp = w + (static_cast<long>(co) * C + c0 + k) * 9;
        if (co == 0) s = wp[0] * G[co][0]; else s = fmaf(wp[0], G[co][0], s);
#pragma unroll
        for (int t = 1; t < 9; ++t) s = fmaf(wp[t], G[co][t], s);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][co][t] = fmaf(pv[k], G[co][t], acc[k][co][t]);
      }
      if (wr) {
        if (VPAD) gb_[k * plane + static_cast<long>(r - 1) * W + (xx - 1)] = s;
        else gb_[k * plane + static_cast<long>(r) * Wp + xx] = s;
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      if (own) accb[co] += G[co][0];
      G[co][6] = G[co][3]; G[co][7] = G[co][4]; G[co][8] = G[co][5]; G[co][3] = G[co][0]; G[co][4] = G[co][1]; G[co][5] = G[co][2];
    }
  }
  // wave sums of the weight-gradient products + the bias terms, in a fixed order
  float* po = part + ((static_cast<long>(b) * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * NACC;
#pragma unroll
  for (int k = 0; k < CI; ++k)
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = wave_total(acc[k][co][t]);
        if (lane == 0) po[(k * CO + co) * 9 + t] = v;
      }
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const float v = wave_total(accb[co]);
    if (lane == 0) po[CI * CO * 9 + co] = v;
  }
}

// gw[(co*C + c)*9 + t] / gb[co]: one wave per output value, lanes stride over the (b, block) partials, fixed butterfly.
// grid: x = CO*C*9 + CO (the last CO blocks finish the bias gradient from the z = 0 chunks)
template <int CO, int CI>
__global__ void __launch_bounds__(64) k_head_final(const float* __restrict__ part, float* __restrict__ gw,
                                                   float* __restrict__ gb, int C, int nunits, int nz) {
  constexpr int NACC = CI * CO * 9 + CO;
  const int o = blockIdx.x, lane = threadIdx.x;
  const bool isb = o >= CO * C * 9;
  int z = 0, idx;
  if (isb) idx = CI * CO * 9 + (o - CO * C * 9);
  else {
    const int co = o / (C * 9), c = (o - co * C * 9) / 9, t = o % 9;
    z = c / CI;
    idx = ((c - z * CI) * CO + co) * 9 + t;
  }
  float s = 0.0f;
  for (int u = lane; u < nunits; u += 64) s += part[(static_cast<long>(u) * nz + z) * NACC + idx];
  const float tot = wave_total(s);
  if (lane == 0) {
    if (isb) { if (gb) gb[o - CO * C * 9] = tot; } else if (gw) gw[o] = tot;
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static int dh_dims(int B, int C, int H, int W, int ci = DH_CI) {
  if (B <= 0 || C <= 0 || H < 1 || W < 1 || B > 65535) return DFE_ERR_DIMS;
  if (C % ci != 0 || C / ci > 65535) return DFE_ERR_UNSUPPORTED;
  if ((static_cast<long>(H) + 2) * (W + 2) * C >= (1L << 31)) return DFE_ERR_DIMS;
  return DFE_OK;
}
static inline int dh_strips(int cols) { return (cols + DH_COLS - 1) / DH_COLS; }
static inline int dh_rowblocks(int rows) { return (rows + DH_ROWS - 1) / DH_ROWS; }

extern "C" long dfe_disp_head_partials_floats(int B, int C, int H, int W) {
  if (dh_dims(B, C, H, W) != DFE_OK) return 0;
  return static_cast<long>(B) * dh_strips(W + 2) * dh_rowblocks(H + 2) * (C / DH_CI) * (DH_CI * 9 + 1);
}

extern "C" int dfe_disp_head_fwd(const float* p, const float* weight, const float* bias, float* out, int B, int C, int H, int W,
                                 void* stream) {
  if (!p || !weight || !out) return DFE_ERR_NULL;
  const int rc = dh_dims(B, C, H, W);
  if (rc != DFE_OK) return rc;
  const int ns = dh_strips(W);
  int R = DH_ROWS;                       // fewer rows per wave on small images: at least ~2048 waves (2-row halo per wave)
  while (R > 2 && static_cast<long>(B) * ns * ((H + R - 1) / R) < 2048) R /= 2;
  k_head_fwd<1, true, false, DH_CI><<<dim3(ns * ((H + R - 1) / R), B), 64, 0, static_cast<hipStream_t>(stream)>>>(p, weight, bias, out, C, H, W, ns, R);
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
  k_head_bwd<1, true, false, DH_CI><<<dim3(ns * nrb, B, nz), 64, 0, st>>>(p, weight, out, gout, gp, partials, C, H, W, ns);
  DFE_LAUNCH_CHECK();
  if (gweight || gbias) {
    k_head_final<1, DH_CI><<<C * 9 + 1, 64, 0, st>>>(partials, gweight, gbias, C, B * ns * nrb, nz);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

// ---- flow head: Conv2d(C, 2, 3, 1, 1) with bias on x [B,C,H,W] (pwc_tf.py:39-40) -> out [B,2,H,W]
extern "C" long dfe_flow_head_partials_floats(int B, int C, int H, int W) {
  if (dh_dims(B, C, H, W, FH_CI) != DFE_OK) return 0;
  return static_cast<long>(B) * dh_strips(W + 2) * dh_rowblocks(H + 2) * (C / FH_CI) * (FH_CI * 2 * 9 + 2);
}

extern "C" int dfe_flow_head_fwd(const float* x, const float* weight, const float* bias, float* out, int B, int C, int H, int W,
                                 void* stream) {
  if (!x || !weight || !out) return DFE_ERR_NULL;
  const int rc = dh_dims(B, C, H, W, FH_CI);
  if (rc != DFE_OK) return rc;
  const int ns = dh_strips(W), nwv = C / FH_CI;
  if (nwv >= FP_R && nwv <= 16) {      // channels across the waves of a block (the finishing step needs FP_R waves)
    const size_t lds = sizeof(float) * nwv * FP_R * 2 * 64;
    k_head_fwd_par<2, false, true, FH_CI><<<dim3(ns * ((H + FP_R - 1) / FP_R), B), 64 * nwv, lds, static_cast<hipStream_t>(stream)>>>(x, weight, bias, out, C, H, W, ns);
  } else {
    int R = DH_ROWS;
    while (R > 2 && static_cast<long>(B) * ns * ((H + R - 1) / R) < 2048) R /= 2;
    k_head_fwd<2, false, true, FH_CI><<<dim3(ns * ((H + R - 1) / R), B), 64, 0, static_cast<hipStream_t>(stream)>>>(x, weight, bias, out, C, H, W, ns, R);
  }
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_flow_head_bwd(const float* x, const float* weight, const float* gout, float* gx, float* gweight, float* gbias,
                                 float* partials, int B, int C, int H, int W, void* stream) {
  if (!x || !weight || !gout || !gx || !partials) return DFE_ERR_NULL;
  const int rc = dh_dims(B, C, H, W, FH_CI);
  if (rc != DFE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ns = dh_strips(W + 2), nrb = dh_rowblocks(H + 2), nz = C / FH_CI;
  k_head_bwd<2, false, true, FH_CI><<<dim3(ns * nrb, B, nz), 64, 0, st>>>(x, weight, nullptr, gout, gx, partials, C, H, W, ns);
  DFE_LAUNCH_CHECK();
  if (gweight || gbias) {
    k_head_final<2, FH_CI><<<2 * C * 9 + 2, 64, 0, st>>>(partials, gweight, gbias, C, B * ns * nrb, nz);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
