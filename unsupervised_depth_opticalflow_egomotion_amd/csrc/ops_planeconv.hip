// 3x3 / stride 1 / pad 1 convolutions on SMALL planes (PWC's decoder levels 6 and 5: 4x13 and 8x26 pixels per sample for a
// 256x832 frame; reference pwc_tf.py:28-47 conv6_0..conv5_4 = Conv2d(3x3, pad 1) + LeakyReLU(0.1), called at
// pwc_tf.py:113-135) on the fp32 matrix cores.  These layers are 0.1-0.4 GFLOP each against 0.3-1.2 MB of weights: MIOpen
// sends them to kernels built for large images (10-45 us per call forward, 26-126 us for a weight gradient, each weight
// gradient wrapped in three layout transposes and a zero fill).  Here every pass is an implicit GEMM on
// v_mfma_f32_16x16x4_f32 (an exact fp32 fma chain) straight from NCHW:
//
//   forward / data gradient (k_planeconv):  M = 16 pixels, N = 16 output channels, K = 4 input channels per instruction,
//     the 9 taps are 9 shifted reads of the same zero-bordered plane in LDS.  A block (4 waves = 64 pixels x 32 or 64 output
//     channels) stages a 16-channel chunk of the plane rows it touches and the matching weight slab once; the channel sum is
//     split over KS blocks (there are only 416 / 1664 pixels: without the split 64 blocks would carry the whole launch) and
//     k_planeconv_finish adds the KS partial planes in slot order (reproducible), adds the bias, applies the activation and
//     writes the result to the one or two concatenated buffers that consume it -- the epilogue launch that followed the
//     MIOpen call anyway.  The data gradient is the same kernel with the weight slab staged transposed and tap-reversed.
//   weight gradient (k_planeconv_wgrad):  M = 16 output channels, N = 16 input channels, K = 4 pixels, 9 accumulator tiles
//     (one per tap) per wave.  Both operands sit in LDS in a row-padded layout (gy with zero pad columns, x with a zero
//     border), so a tap is a constant address offset and no lane ever tests a border.  One block per (sample, row band,
//     32 x 32 channel tile); k_planeconv_finish adds the per-block partial planes in unit order.
// Bound: launch latency and the LDS round trip of one chunk (the whole of level 6 is 0.5 GFLOP and 2.5 MB of weights).
#include "dfe_internal.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

namespace dfe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PC_CC = 16;        // channels per staged chunk (4 K-steps per tap)
constexpr int PC_MT = 64;        // pixels per block: 4 waves x one 16-pixel M tile
constexpr int PC_KW = PC_CC * 9; // weights per output channel and chunk
constexpr int PC_XT_MAX = 18;    // activation staging loads per thread and chunk: bounds the plane width
constexpr int PW_U = 8;          // staging loads in flight per thread and batch in the weight-gradient kernel

// idx / d for 0 <= idx < 2^20 (d > 0, inv = 1 / d): the quotient's distance to the next integer is >= 0.5 / d, the float
// error of the product is < idx / d * 2^-22
__device__ __forceinline__ int fdiv(int idx, float inv) { return static_cast<int>((static_cast<float>(idx) + 0.5f) * inv); }

// x [B,Ck,H,W]; part [KS][B][N][H*W].
// DGRAD = false: w [N][Ck][3][3],  out[n][p] = sum_{c,t} x[c][p + d(t)] * w[n][c][t]
// DGRAD = true : w [Ck][N][3][3],  out[n][p] = sum_{c,t} x[c][p + d(t)] * w[c][n][8 - t]      (x = the output gradient)
// LDS: xs [PC_CC][XP] (rows r0-1 .. r0+RB-2 of the plane with a zero border, pitch W+2), wl [16*NSUB][KP] ([n][c*9 + t])
template <int NSUB, bool DGRAD, int XT>
__global__ void __launch_bounds__(256) k_planeconv(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                   int B, int Ck, int N, int H, int W, int KS, int cps, int RB, int XP, int KP,
                                                   float inv_wp) {
  extern __shared__ float lds[];
  constexpr int NT = 16 * NSUB, WT = NT * PC_KW / 256;
  float* xs = lds;
  float* wl = lds + PC_CC * XP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, kq = lane >> 4;
  const int HW = H * W, Wp = W + 2;
  const int m0 = blockIdx.x * PC_MT, n0 = blockIdx.y * NT;
  const int b = blockIdx.z / KS, ks = blockIdx.z - b * KS;
  const int cbeg = ks * cps, cend = min(Ck, cbeg + cps);
  const int r0 = m0 / W;
  const int p = min(m0 + 16 * wv + m, HW - 1);
  const int py = p / W, px = p - py * W;
  const int abase = kq * XP + (py - r0) * Wp + px;
  const int bbase = m * KP + kq * 9;
  f32x4 acc[NSUB][2];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) { acc[s][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
  const int rbwp = RB * Wp;
  // Staging.  Every load is unconditional from a clamped 32-bit offset and masked when it is written to LDS, so a chunk's
  // loads (XT + WT per thread) are all in flight at once -- and they are issued for chunk i+1 before chunk i is multiplied.
  // Activations: 16 threads per channel copy the positions rem = l16 + 16 i of its zero-bordered band (the offsets inside
  // a plane do not depend on the chunk: computed once).  Weights: 16 threads per row of the slab, 16 i + l16 along it.
  const int sc = tid >> 4, l16 = tid & 15;
  int xoff[XT];
#pragma unroll
  for (int i = 0; i < XT; ++i) {
    const int rem = l16 + 16 * i;
    const int r = fdiv(rem, inv_wp), col = rem - r * Wp;
    const int gy = r0 - 1 + r, gx = col - 1;
    xoff[i] = (rem < rbwp && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
  }
  const unsigned xbase = static_cast<unsigned>(b) * Ck * HW;
  float xv[XT], wr[WT];
  auto fetch = [&](int c0) {
    const int nc = min(PC_CC, cend - c0);
    const unsigned cb = xbase + static_cast<unsigned>(c0 + min(sc, nc - 1)) * HW;
#pragma unroll
    for (int i = 0; i < XT; ++i) xv[i] = x[cb + static_cast<unsigned>(max(xoff[i], 0))];
    if (!DGRAD) {
      const int jmax = nc * 9 - 1;
#pragma unroll
      for (int q = 0; q < NSUB; ++q) {
        const unsigned rb = (static_cast<unsigned>(min(n0 + sc + 16 * q, N - 1)) * Ck + c0) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) wr[q * 9 + i] = w[rb + static_cast<unsigned>(min(l16 + 16 * i, jmax))];
      }
    } else {
      const unsigned rb = (static_cast<unsigned>(c0 + min(sc, nc - 1)) * N + n0) * 9;
      const int rmax = (N - n0) * 9 - 1;
#pragma unroll
      for (int i = 0; i < WT; ++i) wr[i] = w[rb + static_cast<unsigned>(min(l16 + 16 * i, rmax))];
    }
  };
  auto stage = [&](int c0) {
    const int nc = min(PC_CC, cend - c0);
#pragma unroll
    for (int i = 0; i < XT; ++i)
      if (l16 + 16 * i < rbwp) xs[sc * XP + l16 + 16 * i] = (xoff[i] >= 0 && sc < nc) ? xv[i] : 0.0f;
    if (!DGRAD) {
#pragma unroll
      for (int q = 0; q < NSUB; ++q) {
        const bool okn = n0 + sc + 16 * q < N;
#pragma unroll
        for (int i = 0; i < 9; ++i)
          wl[(sc + 16 * q) * KP + l16 + 16 * i] = (okn && l16 + 16 * i < nc * 9) ? wr[q * 9 + i] : 0.0f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        const int r = l16 + 16 * i, n = r / 9, t = r - n * 9;
        wl[n * KP + sc * 9 + 8 - t] = (sc < nc && n0 + n < N) ? wr[i] : 0.0f;
      }
    }
  };
  if (cbeg < cend) fetch(cbeg);
  for (int c0 = cbeg; c0 < cend; c0 += PC_CC) {
    const int nc = min(PC_CC, cend - c0);
    stage(c0);
    __syncthreads();
    if (c0 + PC_CC < cend) fetch(c0 + PC_CC);
#pragma unroll
    for (int cq = 0; cq < PC_CC / 4; ++cq) {
      if (cq * 4 < nc) {
        const float* xa = xs + abase + cq * 4 * XP;
        const float* wb = wl + bbase + cq * 36;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float a = xa[(t / 3) * Wp + (t % 3)];
#pragma unroll
          for (int s = 0; s < NSUB; ++s)
            acc[s][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[s * 16 * KP + t], acc[s][t & 1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // D[i][j]: lane holds pixels i = 4 kq + r of its wave's tile, output channel j = m of each N tile
  float* po = part + (static_cast<long>(ks) * B + b) * N * HW;
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int n = n0 + 16 * s + m;
    if (n < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pp = m0 + 16 * wv + 4 * kq + r;
        if (pp < HW) po[static_cast<long>(n) * HW + pp] = acc[s][0][r] + acc[s][1][r];
      }
    }
  }
}

// out[b][n][p] = act(sum_ks part[ks][b][n][p] + bias[n]), written to d1 (batch stride bs1) and d2 (optional)
__global__ void __launch_bounds__(256) k_planeconv_finish(const float* __restrict__ part, const float* __restrict__ bias,
                                                          float* __restrict__ d1, long bs1, float* __restrict__ d2, long bs2,
                                                          int total, int NHW, int HW, int KS, float slope, float inv_nhw,
                                                          float inv_hw) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  float s = part[idx];
#pragma unroll 4
  for (int k = 1; k < KS; ++k) s += part[static_cast<long>(k) * total + idx];
  int b = fdiv(idx, inv_nhw);
  int r = idx - b * NHW;
  if (bias) s += bias[fdiv(r, inv_hw)];
  s = s > 0.0f ? s : s * slope;
  d1[b * bs1 + r] = s;
  if (d2) d2[b * bs2 + r] = s;
}

// gy [B,Co,H,W], x [B,Ci,H,W]; part [unit = b * nband + band][tap][Co][Ci].
// LDS: gs [32][GP]: q = (y - y0) * Wp + x, zero in the two pad columns and past the band; xs [32][XP]: rows y0-1 .. y1 with
// a zero border at pitch Wp, so that tap (ty, tx) of position q is xs[q + ty * Wp + tx].
__global__ void __launch_bounds__(256) k_planeconv_wgrad(const float* __restrict__ gy, const float* __restrict__ x,
                                                         float* __restrict__ part, int Ci, int Co, int H, int W, int RS, int nband,
                                                         int Q4, int GP, int XE, int XP, float inv_q4, float inv_xe, float inv_wp) {
  extern __shared__ float lds[];
  float* gs = lds;
  float* xs = lds + 32 * GP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, kq = lane >> 4;
  const int HW = H * W, Wp = W + 2;
  const int b = blockIdx.x / nband, band = blockIdx.x - b * nband;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
  const int y0 = band * RS, y1 = min(H, y0 + RS), Q = (y1 - y0) * Wp;
  const float* gb = gy + static_cast<long>(b) * Co * HW;
  const float* xb = x + static_cast<long>(b) * Ci * HW;
  auto gsite = [&](int idx, long& src, int& dst) -> bool {
    const int c = fdiv(idx, inv_q4), q = idx - c * Q4;
    const int yy = fdiv(q, inv_wp), xx = q - yy * Wp;
    const bool ok = co0 + c < Co && q < Q && xx < W;
    src = ok ? static_cast<long>(co0 + c) * HW + (y0 + yy) * W + xx : 0L;
    dst = c * GP + q;
    return ok;
  };
  auto xsite = [&](int idx, long& src, int& dst) -> bool {
    const int c = fdiv(idx, inv_xe), e = idx - c * XE;
    const int r = fdiv(e, inv_wp), col = e - r * Wp;
    const int yy = y0 - 1 + r, xx = col - 1;
    const bool ok = ci0 + c < Ci && yy >= 0 && yy < H && xx >= 0 && xx < W;
    src = ok ? static_cast<long>(ci0 + c) * HW + yy * W + xx : 0L;
    dst = c * XP + e;
    return ok;
  };
  const int EG = 32 * Q4, EX = 32 * XE;
  for (int base = tid; base < EG; base += 256 * PW_U) {
    float v[PW_U];
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      long src; int dst;
      gsite(min(base + u * 256, EG - 1), src, dst);
      v[u] = gb[src];
    }
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      long src; int dst;
      const bool ok = gsite(min(base + u * 256, EG - 1), src, dst);
      if (base + u * 256 < EG) gs[dst] = ok ? v[u] : 0.0f;
    }
  }
  for (int base = tid; base < EX; base += 256 * PW_U) {
    float v[PW_U];
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      long src; int dst;
      xsite(min(base + u * 256, EX - 1), src, dst);
      v[u] = xb[src];
    }
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      long src; int dst;
      const bool ok = xsite(min(base + u * 256, EX - 1), src, dst);
      if (base + u * 256 < EX) xs[dst] = ok ? v[u] : 0.0f;
    }
  }
  __syncthreads();
  const int wco = wv & 1, wci = wv >> 1;
  const float* ga = gs + (16 * wco + m) * GP + kq;
  const float* xa = xs + (16 * wci + m) * XP + kq;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int q0 = 0; q0 < Q4; q0 += 4) {
    const float a = ga[q0];
#pragma unroll
    for (int t = 0; t < 9; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xa[q0 + (t / 3) * Wp + (t % 3)], acc[t], 0, 0, 0);
  }
  // D[i][j]: rows i = 4 kq + r = output channel, column j = m = input channel; partial planes are [co][ci][tap] like the
  // weight itself (a lane writes the 9 taps of its four (co, ci) pairs: 36 contiguous bytes each)
  float* po = part + static_cast<long>(blockIdx.x) * Co * Ci * 9;
  const int ci = ci0 + 16 * wci + m;
  if (ci < Ci) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * wco + 4 * kq + r;
      if (co < Co) {
        float* q = po + (static_cast<long>(co) * Ci + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) q[t] = acc[t][r];
      }
    }
  }
}

// ---- 1x1 convolutions on tiny planes (PoseCNN's pose_conv / refinement head, pose_cnn.py:32,43,48: Conv2d(256|24|12, 12, 1)
// on 2x7 planes, B = 4: 672 outputs of <= 256 products each -- MIOpen spends ~45 us per pass on them).  One thread per output,
// serial sums in channel (or sample-pixel) order: reproducible.
// TRANSPOSED = false: y[b][n][p] = act(sum_k x[b][k][p] * w[n][k] + bias[n])      w [N][K]
// TRANSPOSED = true : y[b][n][p] =     sum_k x[b][k][p] * w[k][n]                 w [K][N]   (data gradient)
template <bool TRANSPOSED>
__global__ void __launch_bounds__(256) k_conv1x1_small(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int K, int N, int HW,
                                                       int total, float slope) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = idx % HW, bn = idx / HW, n = bn % N, b = bn / N;
  const float* xp = x + static_cast<long>(b) * K * HW + p;
  float s = 0.0f;
  for (int k = 0; k < K; ++k) s = __fmaf_rn(xp[static_cast<long>(k) * HW], TRANSPOSED ? w[static_cast<long>(k) * N + n] : w[static_cast<long>(n) * K + k], s);
  if (!TRANSPOSED) {
    if (bias) s += bias[n];
    s = s > 0.0f ? s : s * slope;
  }
  y[idx] = s;
}

// gw[n][k] = sum_{b,p} gy[b][n][p] * x[b][k][p]
__global__ void __launch_bounds__(256) k_conv1x1_small_wgrad(const float* __restrict__ gy, const float* __restrict__ x,
                                                             float* __restrict__ gw, int B, int K, int N, int HW) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int k = idx % K, n = idx / K;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* g = gy + (static_cast<long>(b) * N + n) * HW;
    const float* xv = x + (static_cast<long>(b) * K + k) * HW;
    for (int p = 0; p < HW; ++p) s = __fmaf_rn(g[p], xv[p], s);
  }
  gw[idx] = s;
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

namespace {
constexpr long PC_MAX_PIXELS = 4096;      // per sample: "small plane"; larger images are MIOpen's

struct PcCfg { int nsub, KS, cps, RB, XP, KP, ntm, ntn, xt; size_t lds; };

int pc_target_blocks() {
  static const int v = [] { const char* e = getenv("DFE_PLANECONV_BLOCKS"); const int t = e ? atoi(e) : 0; return t > 0 ? t : 512; }();
  return v;
}

// forward / data gradient: Ck reduction channels, N output channels
PcCfg pc_cfg(int B, int Ck, int N, int H, int W) {
  PcCfg g;
  const int HW = H * W, Wp = W + 2;
  g.ntm = (HW + PC_MT - 1) / PC_MT;
  const int rows = std::min(H, (PC_MT - 1) / W + 2);
  g.RB = rows + 2;
  g.XP = g.RB * Wp;
  while (g.XP % 32 != 16) ++g.XP;                   // the four channel groups of a wave land on different banks
  g.KP = PC_KW + 4;                                 // KP / 4 odd: 16 channels x 4 K lanes on 64 different banks
  const long base32 = static_cast<long>(B) * g.ntm * ((N + 31) / 32);
  g.nsub = (N > 32 && base32 * ((Ck + 31) / 32) >= 1024) ? 4 : 2;
  g.ntn = (N + 16 * g.nsub - 1) / (16 * g.nsub);
  const long base = static_cast<long>(B) * g.ntm * g.ntn;
  g.xt = (g.RB * Wp + 15) / 16;                     // activation loads per thread and chunk
  int ks = static_cast<int>((pc_target_blocks() + base - 1) / base);
  ks = std::max(1, std::min(ks, (Ck + PC_CC - 1) / PC_CC));
  g.cps = ((Ck + ks - 1) / ks + 3) / 4 * 4;
  g.KS = (Ck + g.cps - 1) / g.cps;
  g.lds = sizeof(float) * (static_cast<size_t>(PC_CC) * g.XP + 16 * g.nsub * g.KP);
  return g;
}

struct PwCfg { int RS, nband, Q4, GP, XE, XP; size_t lds; long units; };

PwCfg pw_cfg(int B, int H, int W) {
  PwCfg g;
  const int Wp = W + 2;
  g.RS = H;
  auto fill = [&](int rs) {
    g.RS = rs; g.nband = (H + rs - 1) / rs;
    g.Q4 = (rs * Wp + 3) / 4 * 4;
    g.GP = g.Q4; if ((g.GP / 4) % 2 == 0) g.GP += 4;
    g.XE = g.Q4 + 2 * Wp + 2;
    g.XP = (g.XE + 3) / 4 * 4; if ((g.XP / 4) % 2 == 0) g.XP += 4;
    g.lds = sizeof(float) * 32 * (static_cast<size_t>(g.GP) + g.XP);
  };
  fill(H);
  while (g.RS > 1 && (g.lds > 40 * 1024 || g.Q4 > 128)) fill((g.RS + 1) / 2);
  g.units = static_cast<long>(B) * g.nband;
  return g;
}

int pc_dims(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  if (static_cast<long>(H) * W > PC_MAX_PIXELS || B > 8192 || Ci > 8192 || Co > 8192) return DFE_ERR_UNSUPPORTED;
  if (((std::min(H, (PC_MT - 1) / W + 2) + 2) * (W + 2) + 15) / 16 > PC_XT_MAX) return DFE_ERR_UNSUPPORTED;   // very wide rows
  return DFE_OK;
}

template <int NSUB, bool DGRAD, int XT>
int pc_launch(const float* x, const float* w, float* part, int B, int Ck, int N, int H, int W, const PcCfg& g, hipStream_t st) {
  const dim3 grid(g.ntm, g.ntn, B * g.KS);
  const int Wp = W + 2;
  k_planeconv<NSUB, DGRAD, XT><<<grid, 256, g.lds, st>>>(x, w, part, B, Ck, N, H, W, g.KS, g.cps, g.RB, g.XP, g.KP,
                                                         1.0f / static_cast<float>(Wp));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

template <int NSUB, bool DGRAD>
int pc_launch_xt(const float* x, const float* w, float* part, int B, int Ck, int N, int H, int W, const PcCfg& g, hipStream_t st) {
  if (g.xt <= 6) return pc_launch<NSUB, DGRAD, 6>(x, w, part, B, Ck, N, H, W, g, st);
  if (g.xt <= 12) return pc_launch<NSUB, DGRAD, 12>(x, w, part, B, Ck, N, H, W, g, st);
  return pc_launch<NSUB, DGRAD, PC_XT_MAX>(x, w, part, B, Ck, N, H, W, g, st);
}

int pc_run(bool dgrad, const float* x, const float* w, const float* bias, float slope, float* d1, long bs1, float* d2, long bs2,
           float* ws, int B, int Ck, int N, int H, int W, hipStream_t st) {
  const PcCfg g = pc_cfg(B, Ck, N, H, W);
  int rc;
  if (g.nsub == 4) rc = dgrad ? pc_launch_xt<4, true>(x, w, ws, B, Ck, N, H, W, g, st) : pc_launch_xt<4, false>(x, w, ws, B, Ck, N, H, W, g, st);
  else rc = dgrad ? pc_launch_xt<2, true>(x, w, ws, B, Ck, N, H, W, g, st) : pc_launch_xt<2, false>(x, w, ws, B, Ck, N, H, W, g, st);
  if (rc != DFE_OK) return rc;
  const long HW = static_cast<long>(H) * W, total = B * N * HW;
  k_planeconv_finish<<<static_cast<unsigned>((total + 255) / 256), 256, 0, st>>>(
      ws, bias, d1, bs1, d2, bs2, static_cast<int>(total), static_cast<int>(N * HW), static_cast<int>(HW), g.KS, slope,
      1.0f / static_cast<float>(N * HW), 1.0f / static_cast<float>(HW));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
}  // namespace

extern "C" int dfe_planeconv_supported(int B, int Ci, int Co, int H, int W) {
  if (pc_dims(B, Ci, Co, H, W) != DFE_OK) return 0;
  return static_cast<long>(B) * Co * H * W < (1L << 20) && static_cast<long>(B) * Ci * H * W < (1L << 20) &&
         static_cast<long>(Co) * Ci * 9 < (1L << 20);   // fdiv's range
}

extern "C" long dfe_planeconv_ws_floats(int B, int Ci, int Co, int H, int W) {
  if (!dfe_planeconv_supported(B, Ci, Co, H, W)) return 0;
  const long HW = static_cast<long>(H) * W;
  const long f = pc_cfg(B, Ci, Co, H, W).KS * static_cast<long>(B) * Co * HW;
  const long d = pc_cfg(B, Co, Ci, H, W).KS * static_cast<long>(B) * Ci * HW;
  const long wg = pw_cfg(B, H, W).units * Co * Ci * 9;
  return std::max(f, std::max(d, wg));
}

extern "C" int dfe_planeconv_fwd(const float* x, const float* weight, const float* bias, float slope, float* dst1,
                                 long dst1_batch_stride, float* dst2, long dst2_batch_stride, float* ws, int B, int Ci, int Co,
                                 int H, int W, void* stream) {
  if (!x || !weight || !dst1 || !ws) return DFE_ERR_NULL;
  const int rc = pc_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  if (!dfe_planeconv_supported(B, Ci, Co, H, W)) return DFE_ERR_UNSUPPORTED;
  const long chw = static_cast<long>(Co) * H * W;
  if (dst1_batch_stride < chw || (dst2 && dst2_batch_stride < chw)) return DFE_ERR_DIMS;
  return pc_run(false, x, weight, bias, slope, dst1, dst1_batch_stride, dst2, dst2_batch_stride, ws, B, Ci, Co, H, W,
                static_cast<hipStream_t>(stream));
}

extern "C" int dfe_planeconv_dgrad(const float* gy, const float* weight, float* gx, float* ws, int B, int Ci, int Co, int H, int W,
                                   void* stream) {
  if (!gy || !weight || !gx || !ws) return DFE_ERR_NULL;
  const int rc = pc_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  if (!dfe_planeconv_supported(B, Ci, Co, H, W)) return DFE_ERR_UNSUPPORTED;
  return pc_run(true, gy, weight, nullptr, 1.0f, gx, static_cast<long>(Ci) * H * W, nullptr, 0, ws, B, Co, Ci, H, W,
                static_cast<hipStream_t>(stream));
}

extern "C" int dfe_planeconv_wgrad(const float* gy, const float* x, float* gweight, float* ws, int B, int Ci, int Co, int H, int W,
                                   void* stream) {
  if (!gy || !x || !gweight || !ws) return DFE_ERR_NULL;
  const int rc = pc_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  if (!dfe_planeconv_supported(B, Ci, Co, H, W)) return DFE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PwCfg g = pw_cfg(B, H, W);
  const dim3 grid(static_cast<unsigned>(g.units), (Ci + 31) / 32, (Co + 31) / 32);
  k_planeconv_wgrad<<<grid, 256, g.lds, st>>>(gy, x, ws, Ci, Co, H, W, g.RS, g.nband, g.Q4, g.GP, g.XE, g.XP,
                                              1.0f / static_cast<float>(g.Q4), 1.0f / static_cast<float>(g.XE),
                                              1.0f / static_cast<float>(W + 2));
  DFE_LAUNCH_CHECK();
  const int total = Co * Ci * 9;      // the units' partial planes are added in unit order
  k_planeconv_finish<<<(total + 255) / 256, 256, 0, st>>>(ws, nullptr, gweight, 0, nullptr, 0, total, total, total,
                                                          static_cast<int>(g.units), 1.0f, 1.0f / static_cast<float>(total),
                                                          1.0f / static_cast<float>(total));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

// ---- 1x1 convolutions on tiny planes
static int c1_dims(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  if (static_cast<long>(H) * W > 256 || static_cast<long>(B) * H * W > 4096 || Ci > 4096 || Co > 4096) return DFE_ERR_UNSUPPORTED;
  return DFE_OK;
}

extern "C" int dfe_conv1x1_small_supported(int B, int Ci, int Co, int H, int W) { return c1_dims(B, Ci, Co, H, W) == DFE_OK; }

extern "C" int dfe_conv1x1_small_fwd(const float* x, const float* weight, const float* bias, float slope, float* y, int B, int Ci,
                                     int Co, int H, int W, void* stream) {
  if (!x || !weight || !y) return DFE_ERR_NULL;
  const int rc = c1_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  const int total = B * Co * H * W;
  k_conv1x1_small<false><<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(x, weight, bias, y, Ci, Co, H * W, total, slope);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_conv1x1_small_bwd(const float* gy, const float* x, const float* weight, float* gx, float* gweight, int B, int Ci,
                                     int Co, int H, int W, void* stream) {
  if (!gy || !x || !weight) return DFE_ERR_NULL;
  const int rc = c1_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gx) {
    const int total = B * Ci * H * W;
    k_conv1x1_small<true><<<(total + 255) / 256, 256, 0, st>>>(gy, weight, nullptr, gx, Co, Ci, H * W, total, 1.0f);
    DFE_LAUNCH_CHECK();
  }
  if (gweight) {
    k_conv1x1_small_wgrad<<<(Co * Ci + 255) / 256, 256, 0, st>>>(gy, x, gweight, B, Ci, Co, H * W);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
