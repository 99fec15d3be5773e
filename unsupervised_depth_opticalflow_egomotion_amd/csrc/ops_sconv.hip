// Weight gradients of the networks' STRIDED convolutions on the fp32 matrix cores, straight from NCHW (reference layers: the ResNet18
// encoder's 7x7/2 stem depth_model.py:60-95, FeaturePyramid's first 3x3/2 layers feature_pyramid.py:7-36, PoseCNN's 7x7/2 and 5x5/2
// layers pose_cnn.py:14-36).  MIOpen ran them as NHWC implicit GEMMs wrapped in three layout transposes and a zero fill and reached
// 6 - 40 TFLOP/s on the 3-, 9- and 16-channel inputs (profiles/r05_sconv_bench_all.md); here ONE launch (+ a fixed-order sum of the
// split partials) stages raw NCHW rows in LDS and reads the matrix-core operands from there:
//     dW[co][ci][ky][kx] = sum_pixels gy[co][oy][ox] x[ci][S oy + D ky - P][S ox + D kx - P]
// is a GEMM (co) x (ci, ky, kx) whose reduction runs over the output pixels.  v_mfma_f32_16x16x4_f32 takes "one row / column per
// lane & 15, one of four reduction slots per lane >> 4": the A operand is gy[co = lane & 15][pixel 4 g + (lane >> 4)], the B
// operand x at the lane's OWN (ci, ky, kx) column for the same pixel -- a per-lane LDS offset fixed for the whole kernel plus
// the step's uniform pixel offset.  For stride 2 the staged x rows are split into even and odd columns so that the four
// pixels of a step are four consecutive words.  Columns are either "one filter tap per 16-lane tile, 16 input channels
// across the lanes" (Ci >= 16) or the flattened (ci, ky, kx) index (the 3- and 9-channel stems, 5x5).  The pixel range is
// split over blocks; partials [split][co][ci][ky][kx] are added in split order by k_wgrad_sum: no atomics, bit-reproducible.
// Bound: MFMA (157 TFLOP/s fp32 dense) for the wide layers, HBM for the 3-channel stems.
// (Round 5 also built the forward pass and the data gradient in this style -- filter slab packed [ci / 4][tap][ci % 4][Co], the data
// gradient as four phase correlations -- correct and on a par with MIOpen's whole call per layer, but the step lost 0.05 - 0.15 ms
// with them: removed, EXPERIMENT_LOG.md "strided convolutions".)
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
