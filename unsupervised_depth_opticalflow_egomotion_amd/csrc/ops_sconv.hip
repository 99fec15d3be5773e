// The networks' STRIDED convolutions on the fp32 matrix cores, straight from NCHW (reference layers: the ResNet18 encoder's 7x7/2
// stem and 3x3/2 down-sampling convolutions depth_model.py:60-95, FeaturePyramid's six 3x3/2 layers feature_pyramid.py:7-36,
// PoseCNN's 7x7/2, 5x5/2 and 3x3/2 layers pose_cnn.py:14-36).  MIOpen ran each of them as an NHWC implicit GEMM wrapped in two or
// three layout transposes and (weight gradients) a zero fill: 13 launches per layer and step for forward + both gradients, a
// quarter of the step's launches for 3.8 of its 24.7 ms of kernel time.  Here every pass is ONE launch (+ a fixed-order sum of
// the split partials) that stages raw NCHW rows in LDS and reads the matrix-core operands from there:
//   * weight gradient (k_sconv_wgrad): dW[co][ci][ky][kx] = sum_pixels gy[co][oy][ox] x[ci][S oy + D ky - P][S ox + D kx - P] is a
//     GEMM (co) x (ci, ky, kx) whose reduction runs over the output pixels.  v_mfma_f32_16x16x4_f32 takes "one row / column per
//     lane & 15, one of four reduction slots per lane >> 4": the A operand is gy[co = lane & 15][pixel 4 g + (lane >> 4)], the B
//     operand x at the lane's OWN (ci, ky, kx) column for the same pixel -- a per-lane LDS offset fixed for the whole kernel plus
//     the step's uniform pixel offset.  For stride 2 the staged x rows are split into even and odd columns so that the four
//     pixels of a step are four consecutive words.  Columns are either "one filter tap per 16-lane tile, 16 input channels
//     across the lanes" (Ci >= 16) or the flattened (ci, ky, kx) index (the 3- and 9-channel stems, 5x5).  The pixel range is
//     split over blocks; partials [split][co][ci][ky][kx] are added in split order by k_wgrad_sum: no atomics, bit-reproducible.
// Bound: MFMA (157 TFLOP/s fp32 dense) for the wide layers, HBM for the 3-channel stems.
#include "dfe_internal.h"
#include "dfe_device.h"
#include "dfe_wgrad_sum.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

namespace dfe {

typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
typedef float sc_f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) ScQuadU { float a, b, c, d; };     // dword-aligned 16 bytes

struct ScWg {
  long xbs, gbs;
  int Ci, Co, H, W, Ho, Wo, KH, KW, S, D, P, flat;
  int TH, TW;              // a chunk: TH output rows x TW output columns (TW % 4 == 0)
  int RI, NSLOT, QW;       // staged input rows, 16-byte slots per staged row, words per column phase (stride 2)
  int XRS, XCS, GCS;       // LDS strides: x row, x channel, gy channel (a gy row is TW words)
  int DX;                  // columns the staged window starts left of the first tap's: the window starts on a multiple of 4
  int CIB, COB;            // channels of a block: input (the column slab) / output
  int XCH, GCH;            // ... rounded up to whole thread groups: the channels the LDS regions hold
  int cpr, rpi, nchunks, cps, ncit, ntb;
};

// MT x NT tiles of 16 x 16 per wave, WM x WN waves per block: COB = 16 MT WM output channels x 16 NT WN columns.  Staging: a thread
// group of 2^XSH threads covers the (row, 16-byte slot) positions of the x window and each thread loads NXL channels 256 >> XSH
// apart (2^GSH, NGL for gy): the counts are compile-time so that the prefetch is straight-line code in registers.
template <int MT, int NT, int WM, int WN, int XSH, int NXL, int GSH, int NGL>
__global__ void __launch_bounds__(256, (NT <= 7 && NXL <= 9) ? 3 : 2)
k_sconv_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, const ScWg g) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wm = wv / WN, wn = wv % WN;
  const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tb = static_cast<int>(lid % static_cast<unsigned>(g.ntb)), split = static_cast<int>(lid / static_cast<unsigned>(g.ntb));
  const int cob0 = (tb / g.ncit) * g.COB, cib0 = (tb % g.ncit) * g.CIB;
  const int c_beg = split * g.cps, c_end = min(g.nchunks, c_beg + g.cps);
  const int KK = g.KH * g.KW, HW = g.H * g.W, gHW = g.Ho * g.Wo;
  const int XTOT = g.XCH * g.XCS;

  // ---- the lane's operand offsets
  int boff[NT], oidx[NT], aoff[MT];
  bool any_col = false;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tile = wn * NT + j;
    int ci_l, t;
    bool ok;
    if (g.flat) { const int col = tile * 16 + i; ci_l = col / KK; t = col - ci_l * KK; ok = ci_l < g.CIB; }
    else { t = tile % KK; ci_l = (tile / KK) * 16 + i; ok = ci_l < g.CIB; }
    ok = ok && cib0 + ci_l < g.Ci;
    const int ky = t / g.KW, kx = t - ky * g.KW, dk = g.D * kx + g.DX;
    const int o = ci_l * g.XCS + g.D * ky * g.XRS + (g.S == 2 ? (dk & 1) * g.QW + (dk >> 1) : dk) + kq;
    boff[j] = ok ? o : kq;
    oidx[j] = ok ? (cib0 + ci_l) * KK + t : -1;
    any_col = any_col || ok;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = XTOT + ((wm * MT + mt) * 16 + i) * g.GCS + kq;
  const bool wave_on = __any(any_col) && cob0 + wm * MT * 16 < g.Co;

  // ---- staging: thread = (position = (row, slot) of the window, channel group)
  constexpr int xncg = 256 >> XSH, gncg = 256 >> GSH;
  const int xpos = tid & ((1 << XSH) - 1), xcg = tid >> XSH;
  const int xr_ = xpos / g.NSLOT, xs_ = xpos - xr_ * g.NSLOT;
  const bool xact = xpos < g.RI * g.NSLOT;
  const int xl0 = xcg * g.XCS + xr_ * g.XRS + (g.S == 2 ? 2 : 4) * xs_;
  const int GSL = g.TW >> 2;
  const int gpos = tid & ((1 << GSH) - 1), gcg = tid >> GSH;
  const int gr_ = gpos / GSL, gs_ = gpos - gr_ * GSL;
  const bool gact = gpos < g.TH * GSL;
  const int gl0 = XTOT + gcg * g.GCS + gr_ * g.TW + 4 * gs_;
  sc_f32x4 xr[NXL], gr[NGL];

  // Loads are straight-line: a 16-byte load from the slot's address -- or from the tensor's first words when the slot is outside
  // the image, another channel than the layer has, or cut by the image's edge -- and a select; the cut slots (two per row) are
  // then patched word by word under one branch that whole waves skip.
  auto load_chunk = [&](int c) {
    const int cx = c % g.cpr, ry = (c / g.cpr) % g.rpi, img = c / g.cpr / g.rpi;
    {
      const int iy = g.S * ry * g.TH - g.P + xr_, ix = g.S * cx * g.TW - g.P - g.DX + 4 * xs_;
      const bool rowok = xact && iy >= 0 && iy < g.H;
      const bool full = rowok && ix >= 0 && ix + 3 < g.W;
      const bool cut = rowok && !full && ix + 3 >= 0 && ix < g.W;
      const float* xb = x + img * g.xbs;
      const int off0 = (cib0 + xcg) * HW + iy * g.W + ix;
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj) {
          const int ch = xcg + jj * xncg;
          const bool ok = full && ch < g.CIB && cib0 + ch < g.Ci;
          const ScQuadU u = *reinterpret_cast<const ScQuadU*>(xb + (ok ? off0 + jj * xncg * HW : 0));
          xr[jj] = sc_f32x4{ok ? u.a : 0.0f, ok ? u.b : 0.0f, ok ? u.c : 0.0f, ok ? u.d : 0.0f};
        }
      if (cut) {
#pragma unroll
        for (int jj = 0; jj < NXL; ++jj) {
            const int ch = xcg + jj * xncg;
            if (ch < g.CIB && cib0 + ch < g.Ci) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (ix + e >= 0 && ix + e < g.W) xr[jj][e] = xb[off0 + jj * xncg * HW + e];
            }
          }
      }
    }
    {
      const int oy = ry * g.TH + gr_, ox = cx * g.TW + 4 * gs_;
      const bool rowok = gact && oy < g.Ho;
      const bool full = rowok && ox + 3 < g.Wo;
      const bool cut = rowok && !full && ox < g.Wo;
      const float* gb = gy + img * g.gbs;
      const int off0 = (cob0 + gcg) * gHW + oy * g.Wo + ox;
#pragma unroll
      for (int jj = 0; jj < NGL; ++jj) {
          const int ch = gcg + jj * gncg;
          const bool ok = full && ch < g.COB && cob0 + ch < g.Co;
          const ScQuadU u = *reinterpret_cast<const ScQuadU*>(gb + (ok ? off0 + jj * gncg * gHW : 0));
          gr[jj] = sc_f32x4{ok ? u.a : 0.0f, ok ? u.b : 0.0f, ok ? u.c : 0.0f, ok ? u.d : 0.0f};
        }
      if (cut) {
#pragma unroll
        for (int jj = 0; jj < NGL; ++jj) {
            const int ch = gcg + jj * gncg;
            if (ch < g.COB && cob0 + ch < g.Co) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (ox + e < g.Wo) gr[jj][e] = gb[off0 + jj * gncg * gHW + e];
            }
          }
      }
    }
  };
  // the LDS regions hold whole thread groups of channels (XCH / GCH >= CIB / COB): every active thread stores
  auto store_chunk = [&]() {
    if (xact) {
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj) {
          float* d = lds + xl0 + jj * xncg * g.XCS;
          if (g.S == 2) {
            *reinterpret_cast<sc_f32x2*>(d) = sc_f32x2{xr[jj][0], xr[jj][2]};
            *reinterpret_cast<sc_f32x2*>(d + g.QW) = sc_f32x2{xr[jj][1], xr[jj][3]};
          } else {
            *reinterpret_cast<sc_f32x2*>(d) = sc_f32x2{xr[jj][0], xr[jj][1]};
            *reinterpret_cast<sc_f32x2*>(d + 2) = sc_f32x2{xr[jj][2], xr[jj][3]};
          }
        }
    }
    if (gact) {
#pragma unroll
      for (int jj = 0; jj < NGL; ++jj) {      // two 8-byte stores: the channel stride is even, not a multiple of four
          float* d = lds + gl0 + jj * gncg * g.GCS;
          *reinterpret_cast<sc_f32x2*>(d) = sc_f32x2{gr[jj][0], gr[jj][1]};
          *reinterpret_cast<sc_f32x2*>(d + 2) = sc_f32x2{gr[jj][2], gr[jj][3]};
        }
    }
  };

  sc_f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[mt][j] = sc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  if (c_beg < c_end) {
    load_chunk(c_beg);
    store_chunk();
  }
  __syncthreads();
  const int xstep_row = g.S * g.XRS;
  for (int c = c_beg; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
    if (wave_on) {
      // two steps per trip, the operands of the next step read while this step's MFMAs run (the compiler, left alone, waits for
      // every LDS word right before the MFMA that takes it: a third of the pipe)
      const int n4 = g.TW >> 2;
      for (int oyl = 0; oyl < g.TH; ++oyl) {
        int pa[MT], pb[NT];      // word offsets (pointers would lose the LDS address space: flat loads)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pa[mt] = aoff[mt] + oyl * g.TW;
#pragma unroll
        for (int j = 0; j < NT; ++j) pb[j] = boff[j] + oyl * xstep_row;
        float a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a0[mt] = lds[pa[mt]];
#pragma unroll
        for (int j = 0; j < NT; ++j) b0[j] = lds[pb[j]];
        int s4 = 0;
        // sched_barrier: left alone, the scheduler merges the reads of the two steps into ds_read2 or sinks them to right before
        // their first use (no prefetch left)
        for (; s4 + 2 <= n4; s4 += 2) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a1[mt] = lds[pa[mt] + 4];
#pragma unroll
          for (int j = 0; j < NT; ++j) b1[j] = lds[pb[j] + 4];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], b0[j], acc[mt][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // unconditional (past the row's end in the last trip: words nobody uses, inside the allocation + its 16-word tail): a
          // branch here makes the compiler wait for ALL outstanding LDS words at the loop's head
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a0[mt] = lds[pa[mt] + 8];
#pragma unroll
          for (int j = 0; j < NT; ++j) b0[j] = lds[pb[j] + 8];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b1[j], acc[mt][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pa[mt] += 8;
#pragma unroll
          for (int j = 0; j < NT; ++j) pb[j] += 8;
        }
        if (s4 < n4) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt], b0[j], acc[mt][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- partial sums of this split: [co][ci][ky][kx]
  float* po = part + static_cast<long>(split) * g.Co * g.Ci * KK;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cob0 + (wm * MT + mt) * 16 + 4 * kq + r;
      if (co >= g.Co) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (oidx[j] >= 0) po[static_cast<long>(co) * g.Ci * KK + oidx[j]] = acc[mt][j][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward: y[co][oy][ox] = act(bias[co] + sum_(ci, ky, kx) w[co][ci][ky][kx] x[ci][2 oy + ky - P][2 ox + kx - P]), K x K, stride 2,
// P = K / 2.  A GEMM (co) x (pixels) whose reduction runs over (ci, ky, kx): a step of v_mfma_f32_16x16x4_f32 takes four INPUT
// CHANNELS at one tap.  The A operand is the filter w[co = lane & 15][ci = 4 c4 + (lane >> 4)][tap] from a slab that k_sconv_pack laid
// out as rows [ci / 4][tap][ci % 4][Co] (16-byte copies into LDS, no index arithmetic in the loop), the B operand the staged x at the
// lane's own output pixel (a per-lane LDS offset) plus the tap's uniform offset; x rows are split into even and odd columns as in the
// weight gradient.  A block = COB output channels x a tile of TH x TW output pixels (16 NT WN pixels, numbered row-major inside the
// tile), the input channels in chunks of CB; thin layers split the chunks over blocks and k_sconv_sum adds the partials in order.
struct ScFw {
  long xbs, ybs;
  int Ci, Co, H, W, Ho, Wo;
  int TH, TW, RI, NSLOT, QW, XRS, XCS;
  int CB, nck, cps, nsplit;      // channels per chunk, chunks, chunks per split
  int cpr, rpi, ntile, ncot;
  int Cop, WRS;                  // padded output channels of the packed filter; LDS row stride of the filter slab
  int wl4;                       // 16-byte words of a chunk's filter slab
  float slope;
};

// wp[g4][t][k][Cop] = w[co][4 g4 + k][t] (zero past Ci / Co); TRANSPOSED (data gradient): wp[g4][t][k][Cip] = w[4 g4 + k][ci][t]
template <bool TRANSPOSED>
__global__ void k_sconv_pack(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int KK, int Mp, int rows) {
  const long idx = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<long>(rows) * Mp) return;
  const int m = static_cast<int>(idx % Mp), row = static_cast<int>(idx / Mp);
  const int k = row & 3, t = (row >> 2) % KK, g4 = (row >> 2) / KK, r = 4 * g4 + k;
  float v = 0.0f;
  if (!TRANSPOSED) { if (m < Co && r < Ci) v = w[(static_cast<long>(m) * Ci + r) * KK + t]; }
  else { if (m < Ci && r < Co) v = w[(static_cast<long>(r) * Ci + m) * KK + t]; }
  wp[idx] = v;
}

__global__ void k_sconv_sum(const float* __restrict__ part, float* __restrict__ y, long ybs, long per_img, long n, int S, int HWo,
                            const float* __restrict__ bias, float slope) {
  const long idx = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = 0.0f;
  for (int k = 0; k < S; ++k) s += part[k * n + idx];
  const long b = idx / per_img, r = idx - b * per_img;
  if (bias) s += bias[r / HWo];
  if (slope != 1.0f) s = s > 0.0f ? s : s * slope;
  y[b * ybs + r] = s;
}

template <int K, int MT, int NT, int WM, int WN, int XSH, int NXL, int NWL>
__global__ void __launch_bounds__(256, 2)
k_sconv_fwd(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
            float* __restrict__ part, const ScFw g) {
  constexpr int KK = K * K, P = K / 2, DX = (4 - P % 4) % 4, COB = 16 * MT * WM;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wm = wv / WN, wn = wv % WN;
  const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
  // block = (co block fastest: the blocks that share an x window are neighbours, pixel tile, split)
  const int cb_i = static_cast<int>(lid % static_cast<unsigned>(g.ncot));
  const int tile = static_cast<int>((lid / static_cast<unsigned>(g.ncot)) % static_cast<unsigned>(g.ntile));
  const int split = static_cast<int>(lid / (static_cast<unsigned>(g.ncot) * static_cast<unsigned>(g.ntile)));
  const int cob0 = cb_i * COB;
  const int cx = tile % g.cpr, ry = (tile / g.cpr) % g.rpi, img = tile / g.cpr / g.rpi;
  const int oy0 = ry * g.TH, ox0 = cx * g.TW;
  const int c_beg = split * g.cps, c_end = min(g.nck, c_beg + g.cps);
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
  const int XTOT = g.CB * g.XCS;

  // ---- the lane's output pixels and operand offsets
  int boff[NT], opix[NT], aoff[MT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int p = (wn * NT + j) * 16 + i, oyl = p / g.TW, oxl = p - oyl * g.TW;
    const bool ok = oyl < g.TH && oy0 + oyl < g.Ho && ox0 + oxl < g.Wo;
    boff[j] = kq * g.XCS + (ok ? 2 * oyl * g.XRS + oxl : 0);
    opix[j] = ok ? (oy0 + oyl) * g.Wo + ox0 + oxl : -1;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = XTOT + kq * g.WRS + (wm * MT + mt) * 16 + i;

  // ---- staging: x as in the weight gradient (thread = (row, slot) position, channels 256 >> XSH apart), the filter slab by 16-byte words
  constexpr int xncg = 256 >> XSH;
  const int xpos = tid & ((1 << XSH) - 1), xcg = tid >> XSH;
  const int xr_ = xpos / g.NSLOT, xs_ = xpos - xr_ * g.NSLOT;
  const bool xact = xpos < g.RI * g.NSLOT;
  const int xl0 = xcg * g.XCS + xr_ * g.XRS + 2 * xs_;
  const int iy = 2 * oy0 - P + xr_, ix = 2 * ox0 - P - DX + 4 * xs_;
  const bool rowok = xact && iy >= 0 && iy < g.H;
  const bool full = rowok && ix >= 0 && ix + 3 < g.W;
  const bool cut = rowok && !full && ix + 3 >= 0 && ix < g.W;
  const float* xb = x + img * g.xbs;
  constexpr int WQ = COB / 4;      // 16-byte words per filter row of this block
  sc_f32x4 xr[NXL], wr[NWL];
  auto load_chunk = [&](int c) {
    const int off0 = (c * g.CB + xcg) * HW + iy * g.W + ix;
#pragma unroll
    for (int jj = 0; jj < NXL; ++jj) {
      const int ch = xcg + jj * xncg;
      const bool ok = full && ch < g.CB && c * g.CB + ch < g.Ci;
      const ScQuadU u = *reinterpret_cast<const ScQuadU*>(xb + (ok ? off0 + jj * xncg * HW : 0));
      xr[jj] = sc_f32x4{ok ? u.a : 0.0f, ok ? u.b : 0.0f, ok ? u.c : 0.0f, ok ? u.d : 0.0f};
    }
    if (cut) {
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj) {
        const int ch = xcg + jj * xncg;
        if (ch < g.CB && c * g.CB + ch < g.Ci) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ix + e >= 0 && ix + e < g.W) xr[jj][e] = xb[off0 + jj * xncg * HW + e];
        }
      }
    }
    const float* wsrc = wp + static_cast<long>(c) * g.CB * KK * g.Cop + cob0;
#pragma unroll
    for (int jj = 0; jj < NWL; ++jj) {
      const int e = tid + 256 * jj, row = e / WQ, q = e - row * WQ;
      wr[jj] = e < g.wl4 ? *reinterpret_cast<const sc_f32x4*>(wsrc + static_cast<long>(row) * g.Cop + 4 * q) : sc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto store_chunk = [&]() {
    if (xact) {
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj)
        if (xcg + jj * xncg < g.CB) {
          float* d = lds + xl0 + jj * xncg * g.XCS;
          *reinterpret_cast<sc_f32x2*>(d) = sc_f32x2{xr[jj][0], xr[jj][2]};
          *reinterpret_cast<sc_f32x2*>(d + g.QW) = sc_f32x2{xr[jj][1], xr[jj][3]};
        }
    }
#pragma unroll
    for (int jj = 0; jj < NWL; ++jj) {
      const int e = tid + 256 * jj, row = e / WQ, q = e - row * WQ;
      if (e < g.wl4) *reinterpret_cast<sc_f32x4*>(lds + XTOT + row * g.WRS + 4 * q) = wr[jj];
    }
  };

  sc_f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[mt][j] = sc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  if (c_beg < c_end) {
    load_chunk(c_beg);
    store_chunk();
  }
  __syncthreads();
  const int nc4 = g.CB >> 2;
  for (int c = c_beg; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
    for (int c4 = 0; c4 < nc4; ++c4)
      for (int ky = 0; ky < K; ++ky) {
        // one filter row: the operands of its K taps are all requested before the first MFMA
        const int ab = (c4 * KK + ky * K) * 4 * g.WRS, bb = c4 * 4 * g.XCS + ky * g.XRS;
        float a[K][MT], b[K][NT];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[kx][mt] = lds[aoff[mt] + ab + kx * 4 * g.WRS];
#pragma unroll
          for (int j = 0; j < NT; ++j) b[kx][j] = lds[boff[j] + bb + ((kx + DX) & 1) * g.QW + ((kx + DX) >> 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kx][mt], b[kx][j], acc[mt][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue: bias + activation and y, or this split's partial sums
  const bool direct = g.nsplit == 1;
  const int nimg = g.ntile / (g.cpr * g.rpi);
  float* ob = direct ? y + img * g.ybs : part + (static_cast<long>(split) * nimg + img) * g.Co * HWo;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cob0 + (wm * MT + mt) * 16 + 4 * kq + r;
      if (co >= g.Co) continue;
      const float bv = (direct && bias) ? bias[co] : 0.0f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (opix[j] >= 0) {
          float v = acc[mt][j][r];
          if (direct) { v += bv; v = v > 0.0f ? v : v * g.slope; }
          ob[static_cast<long>(co) * HWo + opix[j]] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Data gradient of the stride-2 convolution: gx[ci][iy][ix] = sum_(co, ky, kx) w[co][ci][ky][kx] gy[co][(iy + P - ky) / 2][(ix + P - kx) / 2]
// over the taps that divide evenly.  Input pixel (2u + py, 2v + px) only meets the taps with ky = (py + P) mod 2, kx = (px + P) mod 2:
// four stride-1 correlations of gy with sub-filters, one per PHASE (py, px).  A block owns COB input channels x a tile of (u, v)
// positions and keeps the four phases' accumulators; per group of four OUTPUT channels it reads the (2 x 2 for 3x3, 3 x 3 for 5x5)
// shifted gy operands once and every tap's filter operand once: MFMA count = the forward's, no zero-stuffed work.  The filter slab
// is k_sconv_pack<true>'s [co / 4][tap][co % 4][Ci]; gy is staged unsplit (stride 1).  1x1 (the ResNet down-sampling shortcuts,
// depth_model.py:31-36): phase (0, 0) gets w^T gy, the three others zeros.
struct ScDg {
  long gbs, xbs;
  int Ci, Co, H, W, Ho, Wo;        // gx is [Ci][H][W], gy [Co][Ho][Wo]
  int TH, TW, RI, NSLOT, XRS, XCS;
  int CB, nck, cps, nsplit;
  int cpr, rpi, ntile, ncit;
  int Cip, WRS, wl4;
};

template <int K, int MT, int NT, int WM, int WN, int XSH, int NXL, int NWL>
__global__ void __launch_bounds__(256, 2)
k_sconv_dgrad(const float* __restrict__ gy, const float* __restrict__ wp, float* __restrict__ gx, float* __restrict__ part, const ScDg g) {
  constexpr int KK = K * K, P = K / 2, CIB = 16 * MT * WM;
  // row shift of tap ky: gy row = u + (py + P - ky) / 2 with py = (ky + P) & 1; DMIN / ND: the smallest shift, the number of shifts
  constexpr int DMIN = K == 5 ? -1 : 0, ND = K == 1 ? 1 : (K == 3 ? 2 : 3), DXA = ((DMIN % 4) + 4) % 4;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wm = wv / WN, wn = wv % WN;
  const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int cb_i = static_cast<int>(lid % static_cast<unsigned>(g.ncit));
  const int tile = static_cast<int>((lid / static_cast<unsigned>(g.ncit)) % static_cast<unsigned>(g.ntile));
  const int split = static_cast<int>(lid / (static_cast<unsigned>(g.ncit) * static_cast<unsigned>(g.ntile)));
  const int cib0 = cb_i * CIB;
  const int cx = tile % g.cpr, ry = (tile / g.cpr) % g.rpi, img = tile / g.cpr / g.rpi;
  const int u0 = ry * g.TH, v0 = cx * g.TW;
  const int c_beg = split * g.cps, c_end = min(g.nck, c_beg + g.cps);
  const int gHW = g.Ho * g.Wo, HW = g.H * g.W;
  const int XTOT = g.CB * g.XCS;

  int boff[NT], pu[NT], pv[NT], aoff[MT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int p = (wn * NT + j) * 16 + i, ul = p / g.TW, vl = p - ul * g.TW;
    const bool ok = ul < g.TH;
    boff[j] = kq * g.XCS + (ok ? ul * g.XRS + vl : 0) + DXA;
    pu[j] = ok ? u0 + ul : -1; pv[j] = v0 + vl;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = XTOT + kq * g.WRS + (wm * MT + mt) * 16 + i;

  constexpr int xncg = 256 >> XSH;
  const int xpos = tid & ((1 << XSH) - 1), xcg = tid >> XSH;
  const int xr_ = xpos / g.NSLOT, xs_ = xpos - xr_ * g.NSLOT;
  const bool xact = xpos < g.RI * g.NSLOT;
  const int xl0 = xcg * g.XCS + xr_ * g.XRS + 4 * xs_;
  const int oy = u0 + DMIN + xr_, ox = v0 + DMIN - DXA + 4 * xs_;
  const bool rowok = xact && oy >= 0 && oy < g.Ho;
  const bool full = rowok && ox >= 0 && ox + 3 < g.Wo;
  const bool cut = rowok && !full && ox + 3 >= 0 && ox < g.Wo;
  const float* gb = gy + img * g.gbs;
  constexpr int WQ = CIB / 4;
  sc_f32x4 xr[NXL], wr[NWL];
  auto load_chunk = [&](int c) {
    const int off0 = (c * g.CB + xcg) * gHW + oy * g.Wo + ox;
#pragma unroll
    for (int jj = 0; jj < NXL; ++jj) {
      const int ch = xcg + jj * xncg;
      const bool ok = full && ch < g.CB && c * g.CB + ch < g.Co;
      const ScQuadU u = *reinterpret_cast<const ScQuadU*>(gb + (ok ? off0 + jj * xncg * gHW : 0));
      xr[jj] = sc_f32x4{ok ? u.a : 0.0f, ok ? u.b : 0.0f, ok ? u.c : 0.0f, ok ? u.d : 0.0f};
    }
    if (cut) {
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj) {
        const int ch = xcg + jj * xncg;
        if (ch < g.CB && c * g.CB + ch < g.Co) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ox + e >= 0 && ox + e < g.Wo) xr[jj][e] = gb[off0 + jj * xncg * gHW + e];
        }
      }
    }
    const float* wsrc = wp + static_cast<long>(c) * g.CB * KK * g.Cip + cib0;
#pragma unroll
    for (int jj = 0; jj < NWL; ++jj) {
      const int e = tid + 256 * jj, row = e / WQ, q = e - row * WQ;
      wr[jj] = e < g.wl4 ? *reinterpret_cast<const sc_f32x4*>(wsrc + static_cast<long>(row) * g.Cip + 4 * q) : sc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto store_chunk = [&]() {
    if (xact) {
#pragma unroll
      for (int jj = 0; jj < NXL; ++jj)
        if (xcg + jj * xncg < g.CB) *reinterpret_cast<sc_f32x4*>(lds + xl0 + jj * xncg * g.XCS) = xr[jj];
    }
#pragma unroll
    for (int jj = 0; jj < NWL; ++jj) {
      const int e = tid + 256 * jj, row = e / WQ, q = e - row * WQ;
      if (e < g.wl4) *reinterpret_cast<sc_f32x4*>(lds + XTOT + row * g.WRS + 4 * q) = wr[jj];
    }
  };

  sc_f32x4 acc[2][2][MT][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[a][b][mt][j] = sc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  if (c_beg < c_end) {
    load_chunk(c_beg);
    store_chunk();
  }
  __syncthreads();
  const int nc4 = g.CB >> 2;
  for (int c = c_beg; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
    for (int c4 = 0; c4 < nc4; ++c4) {
      // the shifted gy operands of this channel group, then filter row by filter row
      float b[ND][ND][NT];
      const int bb = c4 * 4 * g.XCS;
#pragma unroll
      for (int dy = 0; dy < ND; ++dy)
#pragma unroll
        for (int dx = 0; dx < ND; ++dx)
#pragma unroll
          for (int j = 0; j < NT; ++j) b[dy][dx][j] = lds[boff[j] + bb + dy * g.XRS + dx];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        constexpr int dummy = 0; (void)dummy;
        const int py = (ky + P) & 1, dy = (py + P - ky) / 2 - DMIN;
        float a[K][MT];
        const int ab = (c4 * KK + ky * K) * 4 * g.WRS;
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[kx][mt] = lds[aoff[mt] + ab + kx * 4 * g.WRS];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int px = (kx + P) & 1, dx = (px + P - kx) / 2 - DMIN;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[py][px][mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kx][mt], b[dy][dx][j], acc[py][px][mt][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- gx (or this split's partial sums): the two column phases of a position are neighbours
  const int nimg = g.ntile / (g.cpr * g.rpi);
  float* ob = g.nsplit == 1 ? gx + img * g.xbs : part + (static_cast<long>(split) * nimg + img) * g.Ci * HW;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = cib0 + (wm * MT + mt) * 16 + 4 * kq + r;
      if (ci >= g.Ci) continue;
      float* oc = ob + static_cast<long>(ci) * HW;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (pu[j] < 0) continue;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int iy = 2 * pu[j] + py, ix = 2 * pv[j];
          if (iy >= g.H || ix >= g.W) continue;
          float* o = oc + iy * g.W + ix;
          o[0] = acc[py][0][mt][j][r];
          if (ix + 1 < g.W) o[1] = acc[py][1][mt][j][r];
        }
      }
    }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

namespace {
int sc_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int g_sc_blocks = sc_env("DFE_SCONV_BLOCKS", 0), g_sc_th = sc_env("DFE_SCONV_TH", 0);

int sc_log2_ceil(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
int sc_pad_stride(int v) { return v + ((34 - (v & 31)) & 31); }      // the next stride with stride % 32 == 2 (see the bank note below)

// the kernels by layer.  mt, nt, wm, wn: the wave tiling; flat: columns = the flattened (ci, ky, kx) index instead of one tap per
// tile; xsh / nxl, gsh / ngl: the staging shape (k_sconv_wgrad):
//   0: 3x3, Ci >= 16: one tap per tile, 16 input x 64 output channels      1: flat, 160 columns x 64 output channels (7x7 x 3)
//   2: flat, 448 columns x 16 output channels (7x7 x 9, 3x3 x 3)           3: flat, 448 columns x 32 output channels (5x5 x 16)
struct ScTile { int mt, nt, wm, wn, flat, xsh, nxl, gsh, ngl; };
const ScTile SC_TILES[4] = {{1, 9, 4, 1, 0, 7, 8, 4, 4}, {2, 5, 2, 2, 1, 8, 3, 4, 4}, {1, 7, 1, 4, 1, 8, 9, 4, 1}, {2, 7, 1, 4, 1, 8, 16, 4, 2}};

bool sc_wg_plan(int B, int Ci, int Co, int H, int W, int K, int S, int P, ScWg* out, int* cfg, int* nsplit) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (S != 1 && S != 2) || P < 0 || K < 1 || K > 7) return false;
  if (static_cast<long>(Ci) * H * W < 4) return false;      // the loads' safe address: the tensors' first four words
  ScWg g = {};
  g.Ci = Ci; g.Co = Co; g.H = H; g.W = W; g.KH = K; g.KW = K; g.S = S; g.D = 1; g.P = P;
  g.Ho = (H + 2 * P - K) / S + 1; g.Wo = (W + 2 * P - K) / S + 1;
  if (g.Ho < 1 || g.Wo < 1 || static_cast<long>(Co) * g.Ho * g.Wo < 4) return false;
  const int KK = K * K;
  int c = -1;
  if (K == 3 && Ci >= 16) c = 0;      // (a 128-output-channel block, 18 accumulators per wave, measured no faster: 124 / 118 / 125 us against 108 / 103 / 118)
  else if (Ci * KK <= 160 && Co > 32) c = 1;
  else if (Ci * KK <= 448 && Co <= 16 && Ci <= 9) c = 2;
  else if (Ci * KK <= 448 && Co <= 32 && Ci <= 16) c = 3;
  else if (Ci * KK <= 160) c = 1;
  if (c < 0) return false;
  const ScTile t = SC_TILES[c];
  g.flat = t.flat;
  g.COB = 16 * t.mt * t.wm;
  const int cols = 16 * t.nt * t.wn;
  g.CIB = t.flat ? std::min(Ci, cols / KK) : 16 * (t.nt * t.wn / KK);
  const int xncg = 256 >> t.xsh, gncg = 256 >> t.gsh;
  if (g.CIB < 1 || g.CIB > t.nxl * xncg || g.COB > t.ngl * gncg) return false;
  g.XCH = t.nxl * xncg; g.GCH = t.ngl * gncg;
  // the staged window starts on a multiple of four columns (16-byte loads that the image's edge never cuts when W % 4 == 0):
  // DX columns left of the first tap's.  A chunk's first output column is a multiple of TW, TW of 4: DX is the same for all.
  g.DX = ((-P) % 4 + 4) % 4;
  // the chunk: TW output columns (a multiple of 4, at most 64, the row cut into equal parts), TH rows: the most the thread
  // groups' positions hold
  const int wo4 = (g.Wo + 3) / 4 * 4, parts = (wo4 + 63) / 64;
  g.TW = ((g.Wo + parts - 1) / parts + 3) / 4 * 4;
  bool found = false;
  for (int th = 8; th >= 1 && !found; th >>= 1) {
    if (th > 1 && (th * g.TW > 128 || th > g.Ho)) continue;
    if (g_sc_th > 0 && th > g_sc_th) continue;
    g.TH = th;
    g.RI = S * (th - 1) + (K - 1) + 1;
    const int ciw = S * (g.TW - 1) + (K - 1) + 1 + g.DX;
    g.NSLOT = (ciw + 3) / 4;
    if (g.RI * g.NSLOT > (1 << t.xsh) || th * (g.TW / 4) > (1 << t.gsh)) continue;
    g.QW = 2 * g.NSLOT;
    g.XRS = S == 2 ? 2 * g.QW : 4 * g.NSLOT;
    // channel strides % 32 == 2: the lanes of a 16-lane tile differ in the channel (and the four lane groups by one word each):
    // a 32-lane pass of a dword read covers 32 different banks
    g.XCS = sc_pad_stride(g.RI * g.XRS);
    g.GCS = sc_pad_stride(th * g.TW);
    if (sizeof(float) * (static_cast<size_t>(g.XCH) * g.XCS + static_cast<size_t>(g.GCH) * g.GCS + 16) > 76 * 1024) continue;
    found = true;
  }
  if (!found) return false;
  g.cpr = (g.Wo + g.TW - 1) / g.TW; g.rpi = (g.Ho + g.TH - 1) / g.TH;
  const long nch = static_cast<long>(B) * g.cpr * g.rpi;
  if (nch >= (1L << 30)) return false;
  g.nchunks = static_cast<int>(nch);
  const int ncot = (Co + g.COB - 1) / g.COB;
  g.ncit = (Ci + g.CIB - 1) / g.CIB;
  g.ntb = ncot * g.ncit;
  // one resident round of blocks: three per CU for the kernels that fit three waves per SIMD (configurations 1 and 2), else two
  const int target = g_sc_blocks > 0 ? g_sc_blocks : ((c == 1 || c == 2) ? 768 : 512);
  long Sp = (target + g.ntb / 2) / g.ntb;
  Sp = std::max(1L, std::min<long>(Sp, g.nchunks));
  g.cps = static_cast<int>((g.nchunks + Sp - 1) / Sp);
  *nsplit = (g.nchunks + g.cps - 1) / g.cps;
  *out = g; *cfg = c;
  return true;
}

template <int MT, int NT, int WM, int WN, int XSH, int NXL, int GSH, int NGL>
void sc_wg_launch(const ScWg& g, int nsplit, const float* x, const float* gy, float* ws, hipStream_t st) {
  const size_t lds_bytes = sizeof(float) * (static_cast<size_t>(g.XCH) * g.XCS + static_cast<size_t>(g.GCH) * g.GCS + 16);
  auto kern = k_sconv_wgrad<MT, NT, WM, WN, XSH, NXL, GSH, NGL>;
  static bool attr_set = false;      // > 64 KB of dynamic LDS needs the opt-in attribute once per kernel
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; }
  kern<<<static_cast<unsigned>(g.ntb) * nsplit, 256, lds_bytes, st>>>(x, gy, ws, g);
}
}  // namespace

extern "C" int dfe_sconv_tune(int blocks, int rows) {
  if (blocks >= 0) g_sc_blocks = blocks;
  if (rows >= 0) g_sc_th = rows;
  return DFE_OK;
}

extern "C" long dfe_sconv_wgrad_floats(int B, int Ci, int Co, int H, int W, int K, int stride, int P) {
  ScWg g; int cfg, ns;
  if (!sc_wg_plan(B, Ci, Co, H, W, K, stride, P, &g, &cfg, &ns)) return 0;
  return static_cast<long>(ns) * Co * Ci * K * K;
}

extern "C" int dfe_sconv_wgrad(const float* x, long x_batch_stride, const float* gy, long gy_batch_stride, float* gweight, float* ws, int B,
                               int Ci, int Co, int H, int W, int K, int stride, int P, void* stream) {
  if (!x || !gy || !gweight || !ws) return DFE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  ScWg g; int cfg, ns;
  if (!sc_wg_plan(B, Ci, Co, H, W, K, stride, P, &g, &cfg, &ns)) return DFE_ERR_UNSUPPORTED;
  const long gplane = static_cast<long>(g.Ho) * g.Wo;
  if (x_batch_stride < static_cast<long>(Ci) * H * W || gy_batch_stride < Co * gplane) return DFE_ERR_DIMS;
  if (static_cast<long>(Ci) * H * W >= (1L << 30) || Co * gplane >= (1L << 30) || static_cast<long>(Co) * Ci * K * K >= (1L << 30)) return DFE_ERR_DIMS;
  g.xbs = x_batch_stride; g.gbs = gy_batch_stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (cfg) {
    case 0: sc_wg_launch<1, 9, 4, 1, 7, 8, 4, 4>(g, ns, x, gy, ws, st); break;
    case 1: sc_wg_launch<2, 5, 2, 2, 8, 3, 4, 4>(g, ns, x, gy, ws, st); break;
    case 2: sc_wg_launch<1, 7, 1, 4, 8, 9, 4, 1>(g, ns, x, gy, ws, st); break;
    default: sc_wg_launch<2, 7, 1, 4, 8, 16, 4, 2>(g, ns, x, gy, ws, st); break;
  }
  DFE_LAUNCH_CHECK();
  const long n = static_cast<long>(Co) * Ci * K * K;
  if (ns > 64) k_wgrad_sum<32><<<static_cast<unsigned>((n + 31) / 32), 1024, 0, st>>>(ws, gweight, ns, n);
  else k_wgrad_sum<8><<<static_cast<unsigned>((n + 31) / 32), 256, 0, st>>>(ws, gweight, ns, n);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

// ---- forward, host side
namespace {
int sc_pad16(int v) { return v + ((48 - (v & 31)) & 31); }      // the next stride with stride % 32 == 16

struct ScFwCfg { int k, t32, cb, nwl; };      // the instantiations: filter size, 32- or 64-channel block, channels per chunk, filter words per thread

bool sc_fw_plan(int B, int Ci, int Co, int H, int W, int K, ScFw* out, int* t32) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (K != 3 && K != 5 && K != 7)) return false;
  if (static_cast<long>(Ci) * H * W < 4) return false;
  const int P = K / 2, DX = (4 - P % 4) % 4, KK = K * K;
  ScFw g = {};
  g.Ci = Ci; g.Co = Co; g.H = H; g.W = W;
  g.Ho = (H + 2 * P - K) / 2 + 1; g.Wo = (W + 2 * P - K) / 2 + 1;
  if (g.Ho < 1 || g.Wo < 1) return false;
  *t32 = Co <= 32;
  const int COB = *t32 ? 32 : 64, NPIX = 128;
  const int cip = (Ci + 3) / 4 * 4;
  g.CB = K == 3 ? std::min(8, cip) : 4;
  g.nck = (cip + g.CB - 1) / g.CB;
  g.Cop = (Co + 63) / 64 * 64;
  g.WRS = COB + 16;
  g.wl4 = g.CB * KK * COB / 4;
  g.ncot = (Co + COB - 1) / COB;
  // the pixel tile: TW (even: the staged window starts on a multiple of four columns) x TH <= 128 pixels whose window fits the 256
  // staging positions and the LDS; the most useful pixels per tile
  double best = 0.0;
  for (int tw = std::min(64, (g.Wo + 1) / 2 * 2); tw >= 2; tw -= 2) {
    const int th = std::min(NPIX / tw, g.Ho);
    if (th < 1) continue;
    const int ri = 2 * (th - 1) + K, nslot = (2 * (tw - 1) + K + DX + 3) / 4;
    if (ri * nslot > 256) continue;
    const int qw = 2 * nslot, xrs = 2 * qw, xcs = sc_pad16(ri * xrs);
    if (sizeof(float) * (static_cast<size_t>(g.CB) * xcs + static_cast<size_t>(g.CB) * KK * g.WRS + 64) > 80 * 1024) continue;
    const int cpr = (g.Wo + tw - 1) / tw, rpi = (g.Ho + th - 1) / th;
    const double eff = static_cast<double>(g.Ho) * g.Wo / (static_cast<double>(cpr) * rpi * NPIX);
    if (eff > best + 1e-9) { best = eff; g.TW = tw; g.TH = th; g.RI = ri; g.NSLOT = nslot; g.QW = qw; g.XRS = xrs; g.XCS = xcs; g.cpr = cpr; g.rpi = rpi; }
  }
  if (best == 0.0) return false;
  const long nt = static_cast<long>(B) * g.cpr * g.rpi;
  if (nt * g.ncot >= (1L << 24)) return false;
  g.ntile = static_cast<int>(nt);
  // thin layers: the channel chunks are split over blocks until there are two blocks per CU
  const long blocks = nt * g.ncot;
  long sp = blocks >= 384 ? 1 : std::min<long>(g.nck, (512 + blocks - 1) / blocks);
  g.cps = static_cast<int>((g.nck + sp - 1) / sp);
  g.nsplit = (g.nck + g.cps - 1) / g.cps;
  *out = g;
  return true;
}

long sc_fw_packed_floats(const ScFw& g, int K) { return static_cast<long>(g.nck) * g.CB * K * K * g.Cop + 64; }

template <int K, int MT, int NT, int WM, int WN, int NXL, int NWL>
void sc_fw_launch(const ScFw& g, const float* x, const float* wp, const float* bias, float* y, float* part, hipStream_t st) {
  const size_t lds_bytes = sizeof(float) * (static_cast<size_t>(g.CB) * g.XCS + static_cast<size_t>(g.CB) * K * K * g.WRS + 64);
  auto kern = k_sconv_fwd<K, MT, NT, WM, WN, 8, NXL, NWL>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; }
  kern<<<static_cast<unsigned>(g.ntile) * g.ncot * g.nsplit, 256, lds_bytes, st>>>(x, wp, bias, y, part, g);
}
}  // namespace

extern "C" long dfe_sconv_fwd_floats(int B, int Ci, int Co, int H, int W, int K) {
  ScFw g; int t32;
  if (!sc_fw_plan(B, Ci, Co, H, W, K, &g, &t32)) return 0;
  return sc_fw_packed_floats(g, K) + (g.nsplit > 1 ? static_cast<long>(g.nsplit) * B * Co * g.Ho * g.Wo : 0);
}

extern "C" int dfe_sconv_fwd(const float* x, long x_batch_stride, const float* weight, const float* bias, float slope, float* y,
                             long y_batch_stride, float* ws, int B, int Ci, int Co, int H, int W, int K, void* stream) {
  if (!x || !weight || !y || !ws) return DFE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  ScFw g; int t32;
  if (!sc_fw_plan(B, Ci, Co, H, W, K, &g, &t32)) return DFE_ERR_UNSUPPORTED;
  const long oplane = static_cast<long>(g.Ho) * g.Wo;
  if (x_batch_stride < static_cast<long>(Ci) * H * W || y_batch_stride < Co * oplane) return DFE_ERR_DIMS;
  if (static_cast<long>(Ci) * H * W >= (1L << 30) || Co * oplane >= (1L << 30)) return DFE_ERR_DIMS;
  if (reinterpret_cast<uintptr_t>(ws) % 16) return DFE_ERR_DIMS;
  g.xbs = x_batch_stride; g.ybs = y_batch_stride; g.slope = slope;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = g.nck * g.CB * K * K;
  const long npk = static_cast<long>(rows) * g.Cop;
  k_sconv_pack<false><<<static_cast<unsigned>((npk + 255) / 256), 256, 0, st>>>(weight, ws, Co, Ci, K * K, g.Cop, rows);
  DFE_LAUNCH_CHECK();
  float* part = ws + sc_fw_packed_floats(g, K);
  if (K == 3 && !t32) sc_fw_launch<3, 2, 4, 2, 2, 8, 5>(g, x, ws, bias, y, part, st);
  else if (K == 3) sc_fw_launch<3, 2, 2, 1, 4, 8, 3>(g, x, ws, bias, y, part, st);
  else if (K == 5 && !t32) sc_fw_launch<5, 2, 4, 2, 2, 4, 7>(g, x, ws, bias, y, part, st);
  else if (K == 5) sc_fw_launch<5, 2, 2, 1, 4, 4, 4>(g, x, ws, bias, y, part, st);
  else if (!t32) sc_fw_launch<7, 2, 4, 2, 2, 4, 13>(g, x, ws, bias, y, part, st);
  else sc_fw_launch<7, 2, 2, 1, 4, 4, 7>(g, x, ws, bias, y, part, st);
  DFE_LAUNCH_CHECK();
  if (g.nsplit > 1) {
    const long per_img = Co * oplane, n = per_img * B;
    k_sconv_sum<<<static_cast<unsigned>((n + 255) / 256), 256, 0, st>>>(part, y, y_batch_stride, per_img, n, g.nsplit, static_cast<int>(oplane), bias, slope);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

// ---- data gradient, host side
namespace {
bool sc_dg_plan(int B, int Ci, int Co, int H, int W, int K, ScDg* out, int* t32) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (K != 1 && K != 3 && K != 5)) return false;
  const int P = K / 2, KK = K * K;
  ScDg g = {};
  g.Ci = Ci; g.Co = Co; g.H = H; g.W = W;
  g.Ho = (H + 2 * P - K) / 2 + 1; g.Wo = (W + 2 * P - K) / 2 + 1;
  if (g.Ho < 1 || g.Wo < 1 || static_cast<long>(Co) * g.Ho * g.Wo < 4) return false;
  *t32 = Ci <= 32;
  const int CIB = *t32 ? 32 : 64, NPIX = *t32 ? 128 : 64;
  const int cop = (Co + 3) / 4 * 4;
  g.CB = K == 5 ? std::min(8, cop) : std::min(16, cop);
  g.nck = (cop + g.CB - 1) / g.CB;
  g.Cip = (Ci + 63) / 64 * 64;
  g.WRS = CIB + 16;
  g.wl4 = g.CB * KK * CIB / 4;
  g.ncit = (Ci + CIB - 1) / CIB;
  const int dmin = K == 5 ? -1 : 0, nd = K == 1 ? 1 : (K == 3 ? 2 : 3), dxa = ((dmin % 4) + 4) % 4;
  // positions (u, v) of the half-resolution grid whose phases the block writes: U x V = ceil(H / 2) x ceil(W / 2)
  const int U = (H + 1) / 2, V = (W + 1) / 2;
  double best = 0.0;
  for (int tw = std::min(64, (V + 3) / 4 * 4); tw >= 4; tw -= 4) {
    const int th = std::min(NPIX / tw, U);
    if (th < 1) continue;
    const int ri = th + nd - 1, nslot = (tw + nd - 1 + dxa + 3) / 4;
    if (ri * nslot > 64) continue;
    const int xrs = 4 * nslot, xcs = sc_pad16(ri * xrs);
    if (sizeof(float) * (static_cast<size_t>(g.CB) * xcs + static_cast<size_t>(g.CB) * KK * g.WRS + 64) > 76 * 1024) continue;
    const int cpr = (V + tw - 1) / tw, rpi = (U + th - 1) / th;
    const double eff = static_cast<double>(U) * V / (static_cast<double>(cpr) * rpi * NPIX);
    if (eff > best + 1e-9) { best = eff; g.TW = tw; g.TH = th; g.RI = ri; g.NSLOT = nslot; g.XRS = xrs; g.XCS = xcs; g.cpr = cpr; g.rpi = rpi; }
  }
  if (best == 0.0) return false;
  const long nt = static_cast<long>(B) * g.cpr * g.rpi;
  if (nt * g.ncit >= (1L << 24)) return false;
  g.ntile = static_cast<int>(nt);
  const long blocks = nt * g.ncit;
  long sp = blocks >= 384 ? 1 : std::min<long>(g.nck, (512 + blocks - 1) / blocks);
  g.cps = static_cast<int>((g.nck + sp - 1) / sp);
  g.nsplit = (g.nck + g.cps - 1) / g.cps;
  *out = g;
  return true;
}

long sc_dg_packed_floats(const ScDg& g, int K) { return static_cast<long>(g.nck) * g.CB * K * K * g.Cip + 64; }

template <int K, int MT, int NT, int WM, int WN, int NXL, int NWL>
void sc_dg_launch(const ScDg& g, const float* gy, const float* wp, float* gx, float* part, hipStream_t st) {
  const size_t lds_bytes = sizeof(float) * (static_cast<size_t>(g.CB) * g.XCS + static_cast<size_t>(g.CB) * K * K * g.WRS + 64);
  auto kern = k_sconv_dgrad<K, MT, NT, WM, WN, 6, NXL, NWL>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; }
  kern<<<static_cast<unsigned>(g.ntile) * g.ncit * g.nsplit, 256, lds_bytes, st>>>(gy, wp, gx, part, g);
}
}  // namespace

extern "C" long dfe_sconv_dgrad_floats(int B, int Ci, int Co, int H, int W, int K) {
  ScDg g; int t32;
  if (!sc_dg_plan(B, Ci, Co, H, W, K, &g, &t32)) return 0;
  return sc_dg_packed_floats(g, K) + (g.nsplit > 1 ? static_cast<long>(g.nsplit) * B * Ci * H * W : 0);
}

extern "C" int dfe_sconv_dgrad(const float* gy, long gy_batch_stride, const float* weight, float* gx, long gx_batch_stride, float* ws, int B,
                               int Ci, int Co, int H, int W, int K, void* stream) {
  if (!gy || !weight || !gx || !ws) return DFE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  ScDg g; int t32;
  if (!sc_dg_plan(B, Ci, Co, H, W, K, &g, &t32)) return DFE_ERR_UNSUPPORTED;
  const long oplane = static_cast<long>(g.Ho) * g.Wo, iplane = static_cast<long>(H) * W;
  if (gx_batch_stride < Ci * iplane || gy_batch_stride < Co * oplane) return DFE_ERR_DIMS;
  if (Ci * iplane >= (1L << 30) || Co * oplane >= (1L << 30)) return DFE_ERR_DIMS;
  if (reinterpret_cast<uintptr_t>(ws) % 16) return DFE_ERR_DIMS;
  g.gbs = gy_batch_stride; g.xbs = gx_batch_stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = g.nck * g.CB * K * K;
  const long npk = static_cast<long>(rows) * g.Cip;
  k_sconv_pack<true><<<static_cast<unsigned>((npk + 255) / 256), 256, 0, st>>>(weight, ws, Co, Ci, K * K, g.Cip, rows);
  DFE_LAUNCH_CHECK();
  float* part = ws + sc_dg_packed_floats(g, K);
  if (K == 3 && !t32) sc_dg_launch<3, 2, 2, 2, 2, 4, 9>(g, gy, ws, gx, part, st);
  else if (K == 3) sc_dg_launch<3, 2, 2, 1, 4, 4, 5>(g, gy, ws, gx, part, st);
  else if (K == 5 && !t32) sc_dg_launch<5, 2, 2, 2, 2, 2, 13>(g, gy, ws, gx, part, st);
  else if (K == 5) sc_dg_launch<5, 2, 2, 1, 4, 2, 7>(g, gy, ws, gx, part, st);
  else if (!t32) sc_dg_launch<1, 2, 2, 2, 2, 4, 1>(g, gy, ws, gx, part, st);
  else sc_dg_launch<1, 2, 2, 1, 4, 4, 1>(g, gy, ws, gx, part, st);
  DFE_LAUNCH_CHECK();
  if (g.nsplit > 1) {
    const long per_img = Ci * iplane, n = per_img * B;
    k_sconv_sum<<<static_cast<unsigned>((n + 255) / 256), 256, 0, st>>>(part, gx, gx_batch_stride, per_img, n, g.nsplit, static_cast<int>(iplane), nullptr, 1.0f);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
