// Weight gradient of the networks' 3x3 / stride-1 convolutions in the Winograd F(2x2, 3x3) domain on the fp32 matrix cores
// (reference layers: depth_model.py:13-58,135-191, pwc_tf.py:28-95, feature_pyramid.py:7-36; MIOpen ran them as NHWC implicit
// GEMMs wrapped in three layout transposes and a zero fill):
//     dg[k][c] = G^T [ sum_tiles (A dY A^T)[k][tile] (.) (B^T d B)[c][tile] ] G
// = 16 GEMMs dU[xi] (k x c) = dM[xi] (k x tiles) * V[xi]^T (tiles x c) whose reduction runs over the TILES.  The operand layout of
// v_mfma_f32_16x16x4_f32 is "one (channel, tile) per lane" for both operands, and a lane that holds the raw 2x2 output-gradient
// tile / 4x4 input patch of ITS (channel, tile) produces all 16 positions of that operand in registers.  So:
//   * a block stages RAW rows of gy and x (a chunk = 8 tiles of one tile row: 2 x 16 gy pixels, 4 x 24 x pixels per channel) in LDS
//     with coalesced 16-byte loads, zeros outside the image -- double-buffered, the next chunk's loads in flight under this chunk's
//     MFMAs, one barrier per chunk; nothing is transformed by a producer;
//   * every wave reads the raw tile / patch of its (channel = lane & 15, tile = lane >> 4) from LDS (2 + 8 reads per 16 x 16
//     channel pair and step), transforms them in registers (12 + 32 adds) and feeds 16 MFMAs per (16 k x 16 c) pair: a wave
//     owns (16 MH) x (16 NH) channels = MH NH 64 accumulator registers; per step of 4 tiles 16 MH NH MFMAs (32 cycles each)
//     stand against 12 MH + 32 NH vector adds;
//   * the reduction over the tiles is split over blocks (chunks of tiles); every block applies G^T . G to its partial sums and
//     writes [split][k][c][9]; k_wgrad_sum adds the splits in order.  No atomics: bit-reproducible.
// Dilated layers (pwc_tf.py:31-36) are this kernel on their dilation x dilation phase images, gathered by the caller (convs.py).
// Bound: MFMA (2.25 x 157 TFLOP/s effective at 100 % of the fp32 matrix pipe).
#include "dfe_internal.h"
#include "dfe_device.h"
#include "dfe_wgrad_sum.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

namespace dfe {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) WgQuadU { float a, b, c, d; };     // dword-aligned 16 bytes

// TWC = tiles per chunk (a multiple of 4: TWC / 4 steps per barrier): 8 or 12
template <int MH, int NH, int WM, int WN, int TWC>
struct WgCfg {
  static constexpr int COB = 16 * MH * WM, CIB = 16 * NH * WN;
  static constexpr int XU = TWC == 8 ? 6 : 8;          // 16-byte slots of an x row that are loaded (2 TWC + 2 columns + alignment)
  static constexpr int XP = 4 * XU;                    // x row in LDS, floats
  // floats per input channel: 4 rows + 4.  A wave reads 8-byte pairs: lane group {0-31} = 16 channels x 2 tiles, each lane two
  // of the 64 banks; with the channel stride / 4 ODD the 16 channels start at 16 different multiples of 4 banks and the two tiles
  // take banks {0,1} / {2,3} of that window: conflict-free.  (Round 5's first version -- stride / 8 odd, the odd patch column read
  // as single dwords -- spent 68 % of its LDS cycles in bank conflicts with the LDS array 64 % busy: profiles/r05_wino_pmc.txt.)
  static constexpr int XCS = 4 * XP + 4;
  static constexpr int GU = TWC / 2;                   // 16-byte slots of a gy row
  static constexpr int GSL = TWC == 8 ? 4 : 8;         // ... as dealt to the threads (a power of two)
  static constexpr int GP = 2 * TWC;                   // gy row in LDS
  static constexpr int GCS = 2 * GP + 4;               // floats per output channel (stride / 4 odd, as XCS)
  static constexpr int BUF = CIB * XCS + COB * GCS;    // floats per LDS buffer
  // staging: x rows are dealt as 8 slots (XU used): thread = (slot of the row, row, channel mod 8); a thread's slots differ only
  // in the channel.  gy rows as GSL slots: thread = (slot, row, channel mod 256 / (2 GSL))
  static constexpr int GCH = 256 / (2 * GSL);
  static constexpr int NXT = (CIB + 7) / 8, NGT = (COB + GCH - 1) / GCH;
};

// PP: the forward convolution's padding, 0 or 1
template <int MH, int NH, int WM, int WN, int TWC, int PP>
__global__ void __launch_bounds__(256, 2)
k_wino_wgrad2(const float* __restrict__ x, long xbs, const float* __restrict__ gy, long gbs, float* __restrict__ part, int C, int K,
              int W, int HW, int Hq, int Wq, int Ho, int Wo, int gW, int gHW, int TH, int cpr, int TWn, int nchunks, int cps, int ncit,
              int ntb) {
  typedef WgCfg<MH, NH, WM, WN, TWC> Cfg;
  constexpr int WG_XP = Cfg::XP, WG_XCS = Cfg::XCS, WG_GP = Cfg::GP, WG_GCS = Cfg::GCS, WG_TWC = TWC;
  constexpr int S = PP == 1 ? 3 : 0;       // LDS column of a chunk's first patch column: global column 16 cx - PP sits at S
  constexpr int CS = PP == 1 ? 4 : 0;      // the staged window starts CS columns left of column 16 cx
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wm = wv / WN, wn = wv % WN;
  const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tb = static_cast<int>(lid % static_cast<unsigned>(ntb)), split = static_cast<int>(lid / static_cast<unsigned>(ntb));
  const int cob0 = (tb / ncit) * Cfg::COB, cib0 = (tb % ncit) * Cfg::CIB;
  const int c_beg = split * cps, c_end = min(nchunks, c_beg + cps);

  // ---- staging: thread = (16-byte slot q of the row, row r, channel ch + 8 k [x] / ch + 32 k [gy])
  const int xq_ = tid & 7, xr_ = (tid >> 3) & 3, xc_ = tid >> 5;
  const int gq_ = tid & (Cfg::GSL - 1), gr_ = (tid / Cfg::GSL) & 1, gc_ = tid / (2 * Cfg::GSL);
  const int xl0 = xc_ * WG_XCS + xr_ * WG_XP + 4 * xq_;
  const int gl0 = Cfg::CIB * WG_XCS + gc_ * WG_GCS + gr_ * WG_GP + 4 * gq_;
  const int xg0 = (cib0 + xc_) * HW + (xr_ - PP) * W + (4 * xq_ - CS);
  const int gg0 = (cob0 + gc_) * gHW + gr_ * gW + 4 * gq_;
  wg_f32x4 xr[Cfg::NXT], gr[Cfg::NGT];
  int cx = c_beg % cpr, ty = (c_beg / cpr) % TH, img = c_beg / cpr / TH;       // the chunk load_chunk stages next
  auto load_chunk = [&]() {
    const long xb = img * xbs + static_cast<long>(2 * ty) * W + 2 * TWC * cx;
    const long gb = img * gbs + static_cast<long>(2 * ty) * gW + 2 * TWC * cx;
    {
      const int col = 2 * TWC * cx + 4 * xq_ - CS, yy = 2 * ty - PP + xr_;
      const bool rok = xq_ < Cfg::XU && yy >= 0 && yy < Hq;
      const bool full = rok && col >= 0 && col + 3 < Wq, some = rok && col + 3 >= 0 && col < Wq;
      const float* p0 = x + xb + xg0;
#pragma unroll
      for (int k = 0; k < Cfg::NXT; ++k) {
        const float* p = p0 + static_cast<long>(8 * k) * HW;
        const bool cok = xc_ + 8 * k < Cfg::CIB && cib0 + xc_ + 8 * k < C;
        wg_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        {            // straight-line: one 16-byte load from a safe address, zero unless the slot lies inside
          const WgQuadU u = *reinterpret_cast<const WgQuadU*>(full && cok ? p : x);
          if (full && cok) v = wg_f32x4{u.a, u.b, u.c, u.d};
        }
        if (!full && some && cok) {      // a slot that straddles the image's edge: element by element
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j >= 0 && col + j < Wq) v[j] = p[j];
        }
        xr[k] = v;
      }
    }
    {
      const int col = 2 * TWC * cx + 4 * gq_, oy = 2 * ty + gr_;
      const bool rok = gq_ < Cfg::GU && oy < Ho;
      const bool full = rok && col + 3 < Wo, some = rok && col < Wo;
      const float* p0 = gy + gb + gg0;
#pragma unroll
      for (int k = 0; k < Cfg::NGT; ++k) {
        const float* p = p0 + static_cast<long>(Cfg::GCH * k) * gHW;
        const bool cok = gc_ + Cfg::GCH * k < Cfg::COB && cob0 + gc_ + Cfg::GCH * k < K;
        wg_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        {
          const WgQuadU u = *reinterpret_cast<const WgQuadU*>(full && cok ? p : gy);
          if (full && cok) v = wg_f32x4{u.a, u.b, u.c, u.d};
        }
        if (!full && some && cok) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < Wo) v[j] = p[j];
        }
        gr[k] = v;
      }
    }
    if (++cx == cpr) { cx = 0; if (++ty == TH) { ty = 0; ++img; } }
  };
  auto store_chunk = [&](float* buf) {
    if (xq_ < Cfg::XU) {
#pragma unroll
      for (int k = 0; k < Cfg::NXT; ++k)
        if (xc_ + 8 * k < Cfg::CIB) *reinterpret_cast<wg_f32x4*>(buf + xl0 + 8 * k * WG_XCS) = xr[k];
    }
    if (gq_ < Cfg::GU) {
#pragma unroll
      for (int k = 0; k < Cfg::NGT; ++k)
        if (gc_ + Cfg::GCH * k < Cfg::COB) *reinterpret_cast<wg_f32x4*>(buf + gl0 + Cfg::GCH * k * WG_GCS) = gr[k];
    }
  };

  wg_f32x4 acc[MH][NH][16];
#pragma unroll
  for (int a = 0; a < MH; ++a)
#pragma unroll
    for (int b = 0; b < NH; ++b)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[a][b][s] = wg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // consumer addresses: input channel (wn NH + nh) 16 + i, output channel (wm MH + mh) 16 + i, tile 4 ks + kq
  const int xoff = (wn * NH * 16 + i) * WG_XCS + 2 * kq + S;
  const int goff = Cfg::CIB * WG_XCS + (wm * MH * 16 + i) * WG_GCS + 2 * kq;

  auto step = [&](const float* buf, int ks) {
    float m[MH][16];
#pragma unroll
    for (int mh = 0; mh < MH; ++mh) {
      const float* p = buf + goff + mh * 16 * WG_GCS + 8 * ks;
      const wg_f32x2 r0 = *reinterpret_cast<const wg_f32x2*>(p), r1 = *reinterpret_cast<const wg_f32x2*>(p + WG_GP);
      // dM' = A' dY A'^T with A' = diag(1, 1, 1, -1) A (no negations here; the signs are put back in the epilogue)
      const float a0[4] = {r0[0], r0[0] + r1[0], r0[0] - r1[0], r1[0]};
      const float a1[4] = {r0[1], r0[1] + r1[1], r0[1] - r1[1], r1[1]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        m[mh][4 * q] = a0[q]; m[mh][4 * q + 1] = a0[q] + a1[q]; m[mh][4 * q + 2] = a0[q] - a1[q]; m[mh][4 * q + 3] = a1[q];
      }
    }
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) {
      const float* p = buf + xoff + nh * 16 * WG_XCS + 8 * ks;
      float d[16], t[16], v[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (PP == 1) {         // patch column 0 at an odd LDS column: three aligned pairs (dword reads would meet 4 to a bank)
          const wg_f32x2 lo = *reinterpret_cast<const wg_f32x2*>(p + r * WG_XP - 1), mid = *reinterpret_cast<const wg_f32x2*>(p + r * WG_XP + 1),
                         hi = *reinterpret_cast<const wg_f32x2*>(p + r * WG_XP + 3);
          d[4 * r] = lo[1]; d[4 * r + 1] = mid[0]; d[4 * r + 2] = mid[1]; d[4 * r + 3] = hi[0];
        } else {
          const wg_f32x2 lo = *reinterpret_cast<const wg_f32x2*>(p + r * WG_XP), hi = *reinterpret_cast<const wg_f32x2*>(p + r * WG_XP + 2);
          d[4 * r] = lo[0]; d[4 * r + 1] = lo[1]; d[4 * r + 2] = hi[0]; d[4 * r + 3] = hi[1];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = d[j] - d[8 + j];
        t[4 + j] = d[4 + j] + d[8 + j];
        t[8 + j] = d[8 + j] - d[4 + j];
        t[12 + j] = d[4 + j] - d[12 + j];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q * 4 + 0] = t[q * 4] - t[q * 4 + 2];
        v[q * 4 + 1] = t[q * 4 + 1] + t[q * 4 + 2];
        v[q * 4 + 2] = t[q * 4 + 2] - t[q * 4 + 1];
        v[q * 4 + 3] = t[q * 4 + 1] - t[q * 4 + 3];
      }
#pragma unroll
      for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[mh][nh][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[mh][s], v[s], acc[mh][nh][s], 0, 0, 0);
    }
  };

  if (c_beg < c_end) {
    load_chunk();
    store_chunk(lds);
  }
  __syncthreads();
  int cur = 0, tcx = c_beg % cpr;          // tcx: the tile-row chunk being multiplied
  for (int c = c_beg; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) load_chunk();
    const float* buf = lds + cur * Cfg::BUF;
    const int tiles = min(WG_TWC, TWn - WG_TWC * tcx);
    if (++tcx == cpr) tcx = 0;
    step(buf, 0);
    if (tiles > 4) step(buf, 1);
    if (TWC > 8 && tiles > 8) step(buf, 2);
    if (more) {
      store_chunk(lds + (cur ^ 1) * Cfg::BUF);      // its last readers passed the previous barrier
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: D[m][n] -- lane holds input channel n = i, output channels m = 4 kq + r.  gw = G^T dU G per (k, c).
  float* po = part + static_cast<long>(split) * K * C * 9;
#pragma unroll
  for (int mh = 0; mh < MH; ++mh)
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = cob0 + (wm * MH + mh) * 16 + 4 * kq + r, cc = cib0 + (wn * NH + nh) * 16 + i;
        if (k >= K || cc >= C) continue;
        float u[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const float val = acc[mh][nh][4 * a + b][r];
            u[a][b] = ((a == 3) != (b == 3)) ? -val : val;      // the signs of A' (row / column 3)
          }
        float tm[3][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          tm[0][s] = u[0][s] + 0.5f * (u[1][s] + u[2][s]);
          tm[1][s] = 0.5f * (u[1][s] - u[2][s]);
          tm[2][s] = 0.5f * (u[1][s] + u[2][s]) + u[3][s];
        }
        float* o = po + (static_cast<long>(k) * C + cc) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          o[a * 3 + 0] = tm[a][0] + 0.5f * (tm[a][1] + tm[a][2]);
          o[a * 3 + 1] = 0.5f * (tm[a][1] - tm[a][2]);
          o[a * 3 + 2] = 0.5f * (tm[a][1] + tm[a][2]) + tm[a][3];
        }
      }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

namespace {
struct WgPlan { int mh, nh, twc, ncot, ncit, nchunks, cps, S, cpr, TH, TWn, Ho, Wo; };

int wg_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// tuning knobs (dfe_wino_wgrad_tune): forced wave tile (0 = by shape), block targets of the 64- / 128-accumulator kernels, chunk
int g_wg_tune[4] = {wg_env("DFE_WGRAD_TILE", 0), wg_env("DFE_WGRAD_BLOCKS1", 768), wg_env("DFE_WGRAD_BLOCKS2", 512), wg_env("DFE_WGRAD_CHUNK", 12)};

// H, W: the input's size; P: the forward padding (0 / 1)
bool wg_plan(int B, int Ci, int Co, int H, int W, int P, WgPlan* pl) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || P < 0 || P > 1) return false;
  WgPlan p;
  p.Ho = H + 2 * P - 2; p.Wo = W + 2 * P - 2;
  if (p.Ho < 1 || p.Wo < 1) return false;
  p.TH = (p.Ho + 1) / 2; p.TWn = (p.Wo + 1) / 2;
  const int force = g_wg_tune[0];      // 21 / 12 / 11: force the wave tile (16 MH x 16 NH channels)
  // wave tile (16 MH x 16 NH channels; a block = 2 x 2 waves), from tools/wgrad_bench.py (profiles/r05_wgrad_bench.md): 32 x 32 blocks
  // (64 accumulators per wave, three blocks per CU) win wherever 64-channel tiles would be padded or the layer is thin;
  // 64 x 32 blocks (128 accumulators, two per CU) when there are many input channels to share each staged gy tile between
  // (the 64 x 64 form -- 256 accumulators, one wave per SIMD -- spills in the compiler's hands and was dropped)
  const int co_pad = (Co + 63) / 64 * 64 - Co;
  p.mh = (Co >= 64 && co_pad < 16 && Ci >= 128) ? 2 : 1; p.nh = 1;
  if (force >= 11 && force != 22) { p.mh = force / 10; p.nh = force % 10; }
  // 12-tile chunks (3 steps per barrier) where two blocks still share a CU's LDS: every tile but 32 x 64
  p.twc = (p.nh == 2 || g_wg_tune[3] == 8) ? 8 : 12;
  p.cpr = (p.TWn + p.twc - 1) / p.twc;
  const long nch = static_cast<long>(B) * p.TH * p.cpr;
  if (nch >= (1L << 30)) return false;
  p.nchunks = static_cast<int>(nch);
  p.ncot = (Co + 32 * p.mh - 1) / (32 * p.mh); p.ncit = (Ci + 32 * p.nh - 1) / (32 * p.nh);
  const int tgt1 = g_wg_tune[1], tgt2 = g_wg_tune[2];
  const long ntb = static_cast<long>(p.ncot) * p.ncit;
  long S = ((p.mh * p.nh >= 2 ? tgt2 : tgt1) + ntb / 2) / ntb;
  S = std::max(1L, std::min<long>(S, p.nchunks));
  p.cps = static_cast<int>((p.nchunks + S - 1) / S);
  p.S = (p.nchunks + p.cps - 1) / p.cps;
  *pl = p;
  return true;
}
}  // namespace

extern "C" int dfe_wino_wgrad_tune(int tile, int blocks1, int blocks2, int chunk) {
  if (tile != 0 && tile != 11 && tile != 12 && tile != 21) return DFE_ERR_UNSUPPORTED;
  if (chunk != 0 && chunk != 8 && chunk != 12) return DFE_ERR_UNSUPPORTED;
  g_wg_tune[0] = tile;
  if (blocks1 > 0) g_wg_tune[1] = blocks1;
  if (blocks2 > 0) g_wg_tune[2] = blocks2;
  if (chunk > 0) g_wg_tune[3] = chunk;
  return DFE_OK;
}

extern "C" long dfe_wino_wgrad_floats(int B, int Ci, int Co, int H, int W, int P) {
  WgPlan p;
  if (!wg_plan(B, Ci, Co, H, W, P, &p)) return 0;
  return static_cast<long>(p.S) * Co * Ci * 9;
}

template <int MH, int NH, int TWC, int PP>
static void wg_launch(const WgPlan& p, const float* x, long xbs, const float* gy, long gbs, float* ws, int Ci, int Co, int H, int W,
                      hipStream_t st) {
  typedef WgCfg<MH, NH, 2, 2, TWC> Cfg;
  const int ntb = p.ncot * p.ncit;
  const size_t lds_bytes = sizeof(float) * 2 * Cfg::BUF;
  auto kern = k_wino_wgrad2<MH, NH, 2, 2, TWC, PP>;
  static bool attr_set = false;      // > 64 KB of dynamic LDS needs the opt-in attribute once per kernel
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)); attr_set = true; }
  kern<<<static_cast<unsigned>(ntb) * p.S, 256, lds_bytes, st>>>(x, xbs, gy, gbs, ws, Ci, Co, W, H * W, H, W, p.Ho, p.Wo, p.Wo, p.Ho * p.Wo, p.TH,
                                                                   p.cpr, p.TWn, p.nchunks, p.cps, p.ncit, ntb);
}

extern "C" int dfe_wino_wgrad3x3(const float* x, long x_batch_stride, const float* gy, long gy_batch_stride, float* gweight, float* ws, int B,
                                 int Ci, int Co, int H, int W, int P, void* stream) {
  if (!x || !gy || !gweight || !ws) return DFE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  if (P < 0 || P > 1) return DFE_ERR_UNSUPPORTED;
  WgPlan p;
  if (!wg_plan(B, Ci, Co, H, W, P, &p)) return DFE_ERR_DIMS;
  const long gplane = static_cast<long>(p.Ho) * p.Wo;
  if (x_batch_stride < static_cast<long>(Ci) * H * W || gy_batch_stride < Co * gplane) return DFE_ERR_DIMS;
  // 32-bit offsets inside one sample
  if (static_cast<long>(Ci) * H * W >= (1L << 30) || Co * gplane >= (1L << 30) || static_cast<long>(Co) * Ci * 9 >= (1L << 30)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define WG_GO(MHV, NHV, TWV) do { \
    if (P == 1) wg_launch<MHV, NHV, TWV, 1>(p, x, x_batch_stride, gy, gy_batch_stride, ws, Ci, Co, H, W, st); \
    else wg_launch<MHV, NHV, TWV, 0>(p, x, x_batch_stride, gy, gy_batch_stride, ws, Ci, Co, H, W, st); } while (0)
  if (p.mh == 2 && p.twc == 12) WG_GO(2, 1, 12);
  else if (p.mh == 2) WG_GO(2, 1, 8);
  else if (p.nh == 2) WG_GO(1, 2, 8);
  else if (p.twc == 12) WG_GO(1, 1, 12);
  else WG_GO(1, 1, 8);
#undef WG_GO
  DFE_LAUNCH_CHECK();
  const long n = static_cast<long>(Co) * Ci * 9;
  if (p.S > 64) k_wgrad_sum<32><<<static_cast<unsigned>((n + 31) / 32), 1024, 0, st>>>(ws, gweight, p.S, n);
  else k_wgrad_sum<8><<<static_cast<unsigned>((n + 31) / 32), 256, 0, st>>>(ws, gweight, p.S, n);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
