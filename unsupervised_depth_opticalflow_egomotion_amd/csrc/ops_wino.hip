// 3x3 / stride 1 convolutions as Winograd F(2x2, 3x3) with the 16 transformed-domain products on the fp32 matrix cores,
// in ONE kernel (input transform -> MFMA -> output transform): the networks' large layers (ResNet encoder, depth decoder,
// PWC decoder / context network: reference depth_model.py:60-211, pwc_tf.py:28-95, feature_pyramid.py:7-36).  MIOpen runs
// these layers with the same algorithm on the vector ALU (miopenSp3AsmConv_*_f2x3: ~105 TFLOP/s effective); here
//     Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A
// keeps the per-position sums over the input channels as 16 independent GEMMs  M[xi] = U[xi] (k x c) * V[xi] (c x tiles)
// on v_mfma_f32_16x16x4_f32.  A wave owns 32 output channels x 16 tiles and all 16 positions (128 accumulator registers: two
// blocks per CU); per step of 4 input channels a lane loads the 4x4 input patch of ITS (tile, channel) -- the MFMA's B operand
// layout is exactly "one (channel, tile) per lane", so the transformed patch never leaves the registers -- and reads its
// transformed weights from LDS (8-channel slabs, double-buffered, shared by the block's four waves).  The patch arrives as
// one aligned 8-byte load per row plus the neighbouring lanes' pairs by DPP row shifts.  After the channel loop a lane holds,
// for its tile, the 16 positions of 8 output channels: the output transform is register arithmetic and the 2x2 outputs leave
// as stores that are contiguous across the wave.  The data gradient is the same kernel on the transposed, tap-reversed
// filter; dilated layers run on the phase images; planes too small to fill the chip split their input channels over blocks
// (fixed-order partial outputs).  The weight gradient in the same domain: ops_wino_wgrad.hip.
// Same numerics class as the MIOpen kernels it replaces (fp32 Winograd F(2,3)); sums in a fixed order: reproducible.
// Bound: MFMA (2.25 x 157 TFLOP/s effective at 100 % of the fp32 matrix pipe; measured 51 %).
#include "dfe_internal.h"
#include "dfe_device.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

namespace dfe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Optional epilogue of the output transform (round 5): y = act(conv + bias[k]) with act(v) = v > 0 ? v : slope v -- the arithmetic
// of dfe_bias_act_fwd / _fwd2 (ops_epilogue.hip), bit for bit, written to y and (optionally) to the same channels of a second
// buffer: the PWC decoder's concatenated inputs (pwc_tf.py:113-118) and net_utils.conv's Conv2d + LeakyReLU (net_utils.py:7-11)
// no longer pay a read-modify-write pass over every convolution output.  bias == nullptr and slope == 1: no epilogue.
struct WinoEpi { const float* bias; float slope; float* y2; long y2bs; };
__device__ __forceinline__ float wino_act(float v, float bv, float slope) { v += bv; return v > 0.0f ? v : v * slope; }

#ifndef WINO_SCHED
#define WINO_SCHED 1             // 0: the round-5 loop (compiler-placed operand reads)
#endif
constexpr int WN_CC = 16;        // input channels per staged weight slab
constexpr int WN_XP = 20;        // LDS floats per (channel, k): 16 positions + 4 pad (16-byte reads, 64 banks over 16 lanes)

// U[ktile][c][k32][16] = G g G^T of the 3x3 filter from input channel c to output channel ktile*32 + k32 (zero past K).
// TRANSPOSED (data gradient): the filter is w[c][k][8 - t] (w is [Co][Ci][3][3], "k" runs over Ci, "c" over Co).
template <bool TRANSPOSED>
__device__ __forceinline__ void wino_weight_one(const float* __restrict__ w, float* __restrict__ U, int K, int C, int Kpad, int idx) {
  const int k = idx % Kpad, c = idx / Kpad;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = 0.0f;
    if (k < K) v = TRANSPOSED ? w[(static_cast<long>(c) * K + k) * 9 + 8 - t] : w[(static_cast<long>(k) * C + c) * 9 + t];
    g[t] = v;
  }
  float r[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r[0][j] = g[j];
    r[1][j] = 0.5f * ((g[j] + g[3 + j]) + g[6 + j]);
    r[2][j] = 0.5f * ((g[j] - g[3 + j]) + g[6 + j]);
    r[3][j] = g[6 + j];
  }
  float* o = U + ((static_cast<long>(k >> 5) * C + c) * 32 + (k & 31)) * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i * 4 + 0] = r[i][0];
    o[i * 4 + 1] = 0.5f * ((r[i][0] + r[i][1]) + r[i][2]);
    o[i * 4 + 2] = 0.5f * ((r[i][0] - r[i][1]) + r[i][2]);
    o[i * 4 + 3] = r[i][2];
  }
}

template <bool TRANSPOSED>
__global__ void __launch_bounds__(256) k_wino_weights(const float* __restrict__ w, float* __restrict__ U, int K, int C, int Kpad) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Kpad * C) return;
  wino_weight_one<TRANSPOSED>(w, U, K, C, Kpad, idx);
}

// The filters of a whole model in one launch (after the optimiser step): table[e] = {weight, U, K, C, transposed, first block},
// blockmap[block] = e.  Same arithmetic as k_wino_weights: the cached U is bit-identical to the one a call computes for itself.
__global__ void __launch_bounds__(256) k_wino_weights_multi(const long* __restrict__ table, const int* __restrict__ blockmap) {
  const long* e = table + 6L * blockmap[blockIdx.x];
  const float* w = reinterpret_cast<const float*>(e[0]);
  float* U = reinterpret_cast<float*>(e[1]);
  const int K = static_cast<int>(e[2]), C = static_cast<int>(e[3]), Kpad = (K + 31) / 32 * 32;
  const int idx = (static_cast<int>(blockIdx.x) - static_cast<int>(e[5])) * 256 + threadIdx.x;
  if (idx >= Kpad * C) return;
  if (e[4]) wino_weight_one<true>(w, U, K, C, Kpad, idx);
  else wino_weight_one<false>(w, U, K, C, Kpad, idx);
}

// y[b][k][i] = act(sum over the channel splits of part[sp][b][k][i], in split order, + bias[k])
__global__ void __launch_bounds__(256) k_wino_sum(const float* __restrict__ part, float* __restrict__ y, long ybs, int nsp, int B, long KHW,
                                                  long HWo, WinoEpi epi) {
  const long idx = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= B * KHW) return;
  float s = part[idx];
  for (int k = 1; k < nsp; ++k) s += part[static_cast<long>(k) * B * KHW + idx];
  const long b = idx / KHW, r = idx - b * KHW;
  if (epi.bias || epi.slope != 1.0f) s = wino_act(s, epi.bias ? epi.bias[r / HWo] : 0.0f, epi.slope);
  y[b * ybs + r] = s;
  if (epi.y2) epi.y2[b * epi.y2bs + r] = s;
}

// x [B,C,H,W]; y [B,K,Ho,Wo], Ho = H + 2P - 2, Wo = W + 2P - 2 (P = 1: zero padding, 0: valid, 2: full); element (b,k,i) of y at
// y + b * ybs + k * Ho*Wo + i.  Tiles are numbered (b, ty, tx) row-major.  A wave owns 16 tiles x 32 output channels (128
// accumulator registers), so that two blocks share a CU and the hardware hides one wave's loads and LDS reads under the
// other's MFMAs: v_mfma_f32_16x16x4_f32, a lane owns (tile = lane & 15, channel 4 g + (lane >> 4)) of step g; block = 4 waves
// = 64 tiles x 32 output channels.  (The first version -- 32 tiles x 32 channels per wave on v_mfma_f32_32x32x2_f32, 256
// accumulators, one wave per SIMD with software-pipelined loads -- only tied MIOpen and was removed.)
template <int PP, int NW, int NH>   // PP = 0 / 1 / 2: the padding, pair loads (W even); -1: any padding and width, 16 single loads
                                    // per patch.  NW = waves per block.  NH = 16-channel halves of the 32-channel output tile that
                                    // exist (1 when Co <= 16: half the MFMAs, half the accumulators)
__global__ void __launch_bounds__(64 * NW, 2) k_wino_fwd16(const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ y,
                                                       long ybs, int B, int C, int K, int H, int W, int P, int TH, int TW, int ntiles,
                                                       unsigned nkt, int dil, unsigned nsp, int cps, float* __restrict__ part, WinoEpi epi,
                                                       unsigned nitems) {
  extern __shared__ float lds[];      // [WN_CC][32][WN_XP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
  // Round 6: a block CAN be persistent (DFE_WINO_PERSIST) -- the launch then has one round of blocks (two per CU) and a block
  // walks the work items blockIdx.x, blockIdx.x + gridDim.x, ...: no block launch, kernel-argument load and cold first patch per
  // item, and the stores of one item drain under the loop of the next (1-2.5 % per layer on an idle GPU; not the default, see
  // wino_run).  gridDim.x == nitems: one item per block.
  for (unsigned item = blockIdx.x; item < nitems; item += gridDim.x) {
  if (item != blockIdx.x) __syncthreads();      // the previous item's last slab is still being read by slower waves
  // logical item id = tile block * nkt + kt, dealt so that the nkt items of one tile range (they load the same patches)
  // run on the same XCD at about the same time and share its L2 (gridDim.x is a multiple of 8: a block's items stay on its XCD)
  const unsigned lid = xcd_swizzle(item, nitems);
  // nsp > 1: the input channels are split over nsp blocks (planes too small to fill the chip otherwise); each writes its partial
  // output to part[sp] (the output transform is linear) and k_wino_sum adds them in split order
  const int kt = static_cast<int>(lid % nkt), sp = static_cast<int>((lid / nkt) % nsp), tb = static_cast<int>(lid / (nkt * nsp));
  const int cbeg = sp * cps, Cend = min(C, cbeg + cps);
  const int Ho = H + 2 * P - 2, Wo = W + 2 * P - 2, HW = H * W;
  const int tile = min(tb * (16 * NW) + wv * 16 + n, ntiles - 1);
  // dil > 1 (PP = -1 only): the convolution acts on the dil x dil phase images (pixels y = py + dil qy, x = px + dil qx) with
  // padding 1 in phase coordinates; tiles are numbered (b, py, ty, tx, px) so that consecutive lanes read consecutive pixels
  int b, ty, tx, py = 0, px = 0;
  if (PP >= 0 || dil == 1) {
    b = tile / (TH * TW);
    const int tr = tile - b * TH * TW;
    ty = tr / TW; tx = tr - ty * TW;
  } else {
    const int per_row = TW * dil, per_phase = TH * per_row, per_img = per_phase * dil;
    b = tile / per_img;
    int tr = tile - b * per_img;
    py = tr / per_phase; tr -= py * per_phase;
    ty = tr / per_row; tr -= ty * per_row;
    tx = tr / dil; px = tr - tx * dil;
  }
  const int iy0 = 2 * ty - P, ix0 = 2 * tx - P;
  // Pair loads (PP >= 0, W even).  With L / own / R the aligned column pairs (2tx-2, 2tx-1) / (2tx, 2tx+1) / (2tx+2, 2tx+3):
  //   P = 1: a patch row = [L.b, own.a, own.b, R.a]     P = 0: [own.a, own.b, R.a, R.b]     P = 2: [L.a, L.b, own.a, own.b]
  // A lane loads only its own pair (one 8-byte load per row); L and R are the own pairs of the neighbouring lanes of its
  // 16-lane row (consecutive lanes = consecutive tiles) and arrive by DPP row shifts.  Pairs outside the image are the
  // zero border, which also covers the row wraps; the lanes without a usable neighbour (lane 0 / 15 of a row; the last tile
  // of an image row when P = 0) load that pair themselves -- a handful of lanes per load instruction.
  constexpr bool PAIR = PP >= 0;
  constexpr bool USE_L = PP == 1 || PP == 2, USE_R = PP == 1 || PP == 0;
  unsigned off[16], inb = 0;          // BYTE offsets in channel 0 of this lane's sample (0 when outside).  PAIR: off[i] = own pair of
                                      // row i, off[4 + i] = the pair an edge lane loads; inb bits 0-3 rows, 4 = L, 5 = R, 6 = own inside
  bool edge_l = false, edge_r = false;
  if (PAIR) {
    const bool okL = tx >= 1, okR = 2 * tx + 3 < W, okO = 2 * tx + 1 < W;
    edge_l = USE_L && okL && n == 0;
    edge_r = USE_R && okR && (n == 15 || tx == TW - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = iy0 + i;
      const bool ok = yy >= 0 && yy < H;
      const unsigned rowb = 4u * (static_cast<unsigned>(b) * C * HW + (ok ? yy * W : 0));
      off[i] = rowb + (okO ? 8u * static_cast<unsigned>(tx) : 0u);
      off[4 + i] = rowb + (edge_l ? 8u * static_cast<unsigned>(tx - 1) : edge_r ? 8u * static_cast<unsigned>(tx + 1) : 0u);
      off[8 + i] = 0; off[12 + i] = 0;
      inb |= ok ? (1u << i) : 0u;
    }
    inb |= (okL ? 16u : 0u) | (okR ? 32u : 0u) | (okO ? 64u : 0u);
  } else {
    const int Hq = H / dil, Wq = W / dil;           // phase-image size (H, W themselves when dil = 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = iy0 + i, xx = ix0 + j;
        const bool ok = yy >= 0 && yy < Hq && xx >= 0 && xx < Wq;
        off[i * 4 + j] = 4u * (static_cast<unsigned>(b) * C * HW + (ok ? (py + dil * yy) * W + px + dil * xx : 0));
        inb |= ok ? (1u << (i * 4 + j)) : 0u;
      }
  }
  f32x4 acc[NH][16];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[h][s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* Ut = U + static_cast<long>(kt) * C * 512;
  const char* xc = reinterpret_cast<const char*>(x);
  float dn[16];                       // PAIR: dn[2 i], dn[2 i + 1] = the own pair of row i; dn[8 + 2 i], dn[9 + 2 i] = an edge lane's L / R pair
  unsigned mn;

  // issue(c) is called with c = cbeg + kq, cbeg + kq + 4, ...: the channel's byte offset is kept incrementally (a per-lane 32-bit
  // multiply is a quarter-rate instruction, and every vector instruction beside the fp32 MFMAs costs its full issue time:
  // profiles/r05_mfma_feed.txt)
  unsigned cbl = 4u * static_cast<unsigned>(cbeg + kq) * HW;
  const unsigned cstep = 16u * static_cast<unsigned>(HW);
  auto issue = [&](int c) {
    const bool cok = c < Cend;
    mn = cok ? inb : 0u;
    const unsigned cb = cok ? cbl : 0u;
    cbl += cstep;
    if (PAIR) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 p = *reinterpret_cast<const f32x2*>(xc + (off[i] + cb));
        dn[2 * i] = p[0]; dn[2 * i + 1] = p[1];
      }
      if (edge_l || edge_r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 p = *reinterpret_cast<const f32x2*>(xc + (off[4 + i] + cb));
          dn[8 + 2 * i] = p[0]; dn[9 + 2 * i] = p[1];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) dn[q] = *reinterpret_cast<const float*>(xc + (off[q] + cb));
    }
  };
  auto shr1 = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true)); };   // row_shr:1
  auto shl1 = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true)); };   // row_shl:1
  // a wave whose 64 patches lie inside the image (most waves) skips the border selects: 121 -> ~80 vector instructions per step
  constexpr unsigned NEED = 0xFu | (USE_L ? 16u : 0u) | (USE_R ? 32u : 0u) | 64u;
  const bool plain_wave = PAIR && __all(static_cast<int>((inb & NEED) == NEED)) != 0;
  auto gather = [&](float (&d)[16]) {                // the masked 4x4 patch of the step whose loads were issued last
    if (PAIR && plain_wave && __all(static_cast<int>(mn != 0u)) != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float o1 = dn[2 * i], o2 = dn[2 * i + 1], e1 = dn[8 + 2 * i], e2 = dn[9 + 2 * i];
        float l1 = 0.0f, l2 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        if (USE_L) {
          if (PP == 2) { l1 = shr1(o1); l1 = edge_l ? e1 : l1; }
          l2 = shr1(o2); l2 = edge_l ? e2 : l2;
        }
        if (USE_R) {
          r1 = shl1(o1); r1 = edge_r ? e1 : r1;
          if (PP == 0) { r2 = shl1(o2); r2 = edge_r ? e2 : r2; }
        }
        if (PP == 1) { d[4 * i] = l2; d[4 * i + 1] = o1; d[4 * i + 2] = o2; d[4 * i + 3] = r1; }
        else if (PP == 0) { d[4 * i] = o1; d[4 * i + 1] = o2; d[4 * i + 2] = r1; d[4 * i + 3] = r2; }
        else { d[4 * i] = l1; d[4 * i + 1] = l2; d[4 * i + 2] = o1; d[4 * i + 3] = o2; }
      }
    } else if (PAIR) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool rok = (mn >> i) & 1u;
        const bool oko = rok && (mn & 64u);
        const float o1 = oko ? dn[2 * i] : 0.0f, o2 = oko ? dn[2 * i + 1] : 0.0f;
        const float e1 = rok ? dn[8 + 2 * i] : 0.0f, e2 = rok ? dn[9 + 2 * i] : 0.0f;
        float l1 = 0.0f, l2 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        if (USE_L) {
          if (PP == 2) l1 = shr1(o1);
          l2 = shr1(o2);
          if (PP == 2) l1 = (mn & 16u) ? (edge_l ? e1 : l1) : 0.0f;
          l2 = (mn & 16u) ? (edge_l ? e2 : l2) : 0.0f;
        }
        if (USE_R) {
          r1 = shl1(o1);
          if (PP == 0) r2 = shl1(o2);
          r1 = (mn & 32u) ? (edge_r ? e1 : r1) : 0.0f;
          if (PP == 0) r2 = (mn & 32u) ? (edge_r ? e2 : r2) : 0.0f;
        }
        if (PP == 1) { d[4 * i] = l2; d[4 * i + 1] = o1; d[4 * i + 2] = o2; d[4 * i + 3] = r1; }
        else if (PP == 0) { d[4 * i] = o1; d[4 * i + 1] = o2; d[4 * i + 2] = r1; d[4 * i + 3] = r2; }
        else { d[4 * i] = l1; d[4 * i + 1] = l2; d[4 * i + 2] = o1; d[4 * i + 3] = o2; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) d[q] = ((mn >> q) & 1u) ? dn[q] : 0.0f;
    }
  };
#pragma unroll
  for (int i = 8; i < 16; ++i) dn[i] = 0.0f;      // the edge pairs: defined on every lane (the selects must not see an undefined value),
                                                  // once -- only the edge lanes ever write them again
  issue(cbeg + kq);
  // weight slabs of 8 input channels, double-buffered: slab i + 1 travels global -> registers under the two steps of slab i
  // and is written to the other LDS buffer before the chunk's single barrier
  constexpr int SC = 8, SLAB = SC * 32 * WN_XP;
  constexpr int NTH = 64 * NW;
  f32x4 wreg[SC * 128 / NTH];
  auto wfetch = [&](int c0) {
    const int nc = min(SC, Cend - c0);
#pragma unroll
    for (int i = 0; i < SC * 128 / NTH; ++i) {
      const int e = tid + i * NTH;
      wreg[i] = *reinterpret_cast<const f32x4*>(Ut + static_cast<long>(c0) * 512 + (e < nc * 128 ? e : 0) * 4);
    }
  };
  auto wstore = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < SC * 128 / NTH; ++i) {
      const int e = tid + i * NTH;
      *reinterpret_cast<f32x4*>(buf + (e >> 2) * WN_XP + (e & 3) * 4) = wreg[i];
    }
  };
  wfetch(cbeg);
  wstore(lds);
  __syncthreads();
  int cur = 0;
  for (int c0 = cbeg; c0 < Cend; c0 += SC) {
    const int nc = min(SC, Cend - c0);
    const bool more = c0 + SC < Cend;
    if (more) wfetch(c0 + SC);
    const float* slab = lds + cur * SLAB;
    for (int cs = 0; cs < nc; cs += 4) {
      float d[16], t[16], v[16];
      // Round 6: the step's filter reads are software-pipelined by hand.  hipcc put each half's four ds_read_b128 right in front of
      // the 16 MFMAs that consume them: two exposed LDS latencies per step.  Here the first two quads are read at the top of the
      // step (their latency passes under the gather and the transform), the others two MFMA groups ahead of their use, through
      // a ring of three quad registers; the order is pinned with sched_group_barrier.
      const float* up = slab + (min(cs + kq, nc - 1) * 32 + n) * WN_XP;    // a lane past the last channel holds a zero patch
      auto ldU = [&](int h, int q) { return *reinterpret_cast<const f32x4*>(up + h * 16 * WN_XP + 4 * q); };
      f32x4 ua, ub, uc;
      if (NH == 2 && WINO_SCHED) { ua = ldU(0, 0); ub = ldU(0, 1); __builtin_amdgcn_sched_barrier(0); }
      gather(d);
      issue(c0 + cs + 4 + kq);                      // the next step's patch
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = d[j] - d[8 + j];
        t[4 + j] = d[4 + j] + d[8 + j];
        t[8 + j] = d[8 + j] - d[4 + j];
        t[12 + j] = d[4 + j] - d[12 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i * 4 + 0] = t[i * 4] - t[i * 4 + 2];
        v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
        v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
        v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
      }
      if (NH == 2 && WINO_SCHED) {
        auto mfma4 = [&](int h, int q, const f32x4& u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[h][4 * q + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j], v[4 * q + j], acc[h][4 * q + j], 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);
        uc = ldU(0, 2);
        mfma4(0, 0, ua); ua = ldU(0, 3);
        mfma4(0, 1, ub); ub = ldU(1, 0);
        mfma4(0, 2, uc); uc = ldU(1, 1);
        mfma4(0, 3, ua); ua = ldU(1, 2);
        mfma4(1, 0, ub); ub = ldU(1, 3);
        mfma4(1, 1, uc);
        mfma4(1, 2, ua);
        mfma4(1, 3, ub);
#define SGB_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
        SGB_DS(1); SGB_MFMA(4); SGB_DS(1); SGB_MFMA(4); SGB_DS(1); SGB_MFMA(4); SGB_DS(1); SGB_MFMA(4); SGB_DS(1); SGB_MFMA(4); SGB_DS(1); SGB_MFMA(12);
#undef SGB_DS
#undef SGB_MFMA
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        float u[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(up + h * 16 * WN_XP + 4 * q);
          u[4 * q] = w4[0]; u[4 * q + 1] = w4[1]; u[4 * q + 2] = w4[2]; u[4 * q + 3] = w4[3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[h][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[s], v[s], acc[h][s], 0, 0, 0);
      }
    }
    if (more) {
      wstore(lds + (cur ^ 1) * SLAB);               // nobody reads that buffer: its last readers passed the previous barrier
      __syncthreads();
      cur ^= 1;
    }
  }
  // D[i][j]: lane holds tile j = n and the output channels i = 4 kq + r of each 16-channel half
  if (tb * (16 * NW) + wv * 16 + n >= ntiles) continue;
  const int Hoq = (PP >= 0 || dil == 1) ? Ho : H / dil, Woq = (PP >= 0 || dil == 1) ? Wo : W / dil;   // outputs per phase image
  const int oy = 2 * ty, ox = 2 * tx;
  const int sy = dil * Wo, sx = dil;                // strides of the 2x2 outputs in y
  const long kplane = static_cast<long>(Ho) * Wo;
  const long ooff = static_cast<long>(py + dil * oy) * Wo + px + dil * ox;
  float* yb = (nsp > 1 ? part + (static_cast<long>(sp) * B + b) * K * kplane : y + b * ybs) + ooff;
  // with channel splits the epilogue belongs to k_wino_sum (the partial outputs are sums of a part of the channels)
  const bool fused = nsp == 1 && (epi.bias != nullptr || epi.slope != 1.0f);
  float* y2b = (nsp == 1 && epi.y2) ? epi.y2 + b * epi.y2bs + ooff : nullptr;
  // 8-byte stores of a tile's row pairs: dense layers with an even output width and 8-byte aligned planes (every plane starts
  // at an even float offset: batch strides and Ho * Wo even)
  const bool even_w = dil == 1 && (Wo & 1) == 0;
  const bool pair_st = even_w && (reinterpret_cast<uintptr_t>(nsp > 1 ? part : y) & 7) == 0 && ((nsp > 1 ? 0 : ybs) & 1) == 0;
  const bool pair_st2 = even_w && y2b && (reinterpret_cast<uintptr_t>(epi.y2) & 7) == 0 && (epi.y2bs & 1) == 0;
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 32 + 16 * h + 4 * kq + r;
      if (k >= K) continue;
      float t0[4], t1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        t0[s] = (acc[h][s][r] + acc[h][4 + s][r]) + acc[h][8 + s][r];
        t1[s] = (acc[h][4 + s][r] - acc[h][8 + s][r]) - acc[h][12 + s][r];
      }
      float y00 = (t0[0] + t0[1]) + t0[2], y01 = (t0[1] - t0[2]) - t0[3];
      float y10 = (t1[0] + t1[1]) + t1[2], y11 = (t1[1] - t1[2]) - t1[3];
      if (fused) {
        const float bv = epi.bias ? epi.bias[k] : 0.0f;
        y00 = wino_act(y00, bv, epi.slope); y01 = wino_act(y01, bv, epi.slope);
        y10 = wino_act(y10, bv, epi.slope); y11 = wino_act(y11, bv, epi.slope);
      }
      float* o = yb + static_cast<long>(k) * Ho * Wo;
      if (pair_st) {        // Wo even, 8-byte aligned rows: a tile's two outputs of a row leave as one 8-byte store (16 lanes = 128 bytes)
        if (ox < Woq) {
          if (oy < Hoq) *reinterpret_cast<f32x2*>(o) = f32x2{y00, y01};
          if (oy + 1 < Hoq) *reinterpret_cast<f32x2*>(o + sy) = f32x2{y10, y11};
        }
      } else {
        if (oy < Hoq) {
          if (ox < Woq) o[0] = y00;
          if (ox + 1 < Woq) o[sx] = y01;
        }
        if (oy + 1 < Hoq) {
          if (ox < Woq) o[sy] = y10;
          if (ox + 1 < Woq) o[sy + sx] = y11;
        }
      }
      if (y2b) {
        float* o2 = y2b + static_cast<long>(k) * Ho * Wo;
        if (pair_st2) {
          if (ox < Woq) {
            if (oy < Hoq) *reinterpret_cast<f32x2*>(o2) = f32x2{y00, y01};
            if (oy + 1 < Hoq) *reinterpret_cast<f32x2*>(o2 + sy) = f32x2{y10, y11};
          }
        } else {
          if (oy < Hoq) {
            if (ox < Woq) o2[0] = y00;
            if (ox + 1 < Woq) o2[sx] = y01;
          }
          if (oy + 1 < Hoq) {
            if (ox < Woq) o2[sy] = y10;
            if (ox + 1 < Woq) o2[sy + sx] = y11;
          }
        }
      }
    }
  }   // items
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static int wn_dims(int B, int Ci, int Co, int H, int W, int P) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  if (P < 0 || P > 2) return DFE_ERR_UNSUPPORTED;
  if (H + 2 * P < 3 || W + 2 * P < 3) return DFE_ERR_DIMS;
  if (static_cast<long>(B) * Ci * H * W >= (1L << 30) || static_cast<long>(Co) * H * W >= (1L << 31) || Co > 65535 * 32) return DFE_ERR_DIMS;   // 32-bit offsets
  return DFE_OK;
}

extern "C" long dfe_wino_weight_floats(int Ci, int Co) { return (Ci <= 0 || Co <= 0) ? 0 : static_cast<long>((Co + 31) / 32) * 32 * Ci * 16; }

// transposed_weight: see include/dfe_hip.h
extern "C" long dfe_wino_weight_floats(int Ci, int Co);

struct WinoSplit { int nsp, cps; long part_floats; };

// channel splits for the planes that cannot fill the chip: ntb tile blocks x nkt k-tiles < 384 blocks
static WinoSplit wino_split(long ntiles, int Kpad, int Ci, int B, int Co, int Ho, int Wo) {
  WinoSplit w{1, Ci, 0};
  static const bool enabled = [] { const char* e = getenv("DFE_WINO_SPLIT"); return !e || atoi(e) != 0; }();
  const long nblk = (ntiles + 63) / 64 * (Kpad / 32);
  if (!enabled || nblk >= 384 || Ci < 64) return w;
  int nsp = static_cast<int>((512 + nblk - 1) / nblk);
  nsp = std::min(nsp, std::min(Ci / 32, 8));
  if (nsp < 2) return w;
  w.cps = ((Ci + nsp - 1) / nsp + 7) / 8 * 8;
  w.nsp = (Ci + w.cps - 1) / w.cps;
  if (w.nsp < 2) { w.nsp = 1; w.cps = Ci; return w; }
  w.part_floats = static_cast<long>(w.nsp) * B * Co * Ho * Wo;
  return w;
}

// weight == nullptr: wbuf already holds the transformed filters (dfe_wino_conv3x3_u) and is only read; part / part_floats: room
// for the channel splits' partial outputs (may be null / 0)
static int wino_run(const float* x, const float* weight, float* y, long y_batch_stride, float* wbuf, float* part, long part_floats, int B,
                    int Ci, int Co, int H, int W, int P, int dil, int transposed_weight, void* stream, WinoEpi epi = WinoEpi{nullptr, 1.0f, nullptr, 0}) {
  if (!x || !y || !wbuf) return DFE_ERR_NULL;
  const int rc = wn_dims(B, Ci, Co, H, W, P);
  if (rc != DFE_OK) return rc;
  if (dil < 1 || (dil > 1 && (P != 1 || H % dil != 0 || W % dil != 0 || H / dil < 2 || W / dil < 2))) return DFE_ERR_UNSUPPORTED;
  // dil > 1: padding = dil in pixels = 1 in phase coordinates: the output has the input's size
  const int Ho = dil > 1 ? H : H + 2 * P - 2, Wo = dil > 1 ? W : W + 2 * P - 2;
  if (y_batch_stride < static_cast<long>(Co) * Ho * Wo || (epi.y2 && epi.y2bs < static_cast<long>(Co) * Ho * Wo)) return DFE_ERR_DIMS;
  if ((reinterpret_cast<uintptr_t>(wbuf) & 15) != 0) return DFE_ERR_UNSUPPORTED;
  if (part && (reinterpret_cast<uintptr_t>(part) & 15) != 0) return DFE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int Kpad = (Co + 31) / 32 * 32;
  const int nw = Kpad * Ci;
  if (weight) {
    if (transposed_weight) k_wino_weights<true><<<(nw + 255) / 256, 256, 0, st>>>(weight, wbuf, Co, Ci, Kpad);
    else k_wino_weights<false><<<(nw + 255) / 256, 256, 0, st>>>(weight, wbuf, Co, Ci, Kpad);
    DFE_LAUNCH_CHECK();
  }
  const int TH = (Ho / dil + 1) / 2, TW = (Wo / dil + 1) / 2;
  const long ntiles = static_cast<long>(B) * TH * TW * dil * dil;
  if (ntiles >= (1L << 31)) return DFE_ERR_DIMS;
  const size_t lds_bytes = sizeof(float) * WN_CC * 32 * WN_XP;
  {
    const unsigned nkt = Kpad / 32;
    WinoSplit sp = wino_split(ntiles, Kpad, Ci, B, Co, Ho, Wo);
    if (dil > 1 || !part || part_floats < sp.part_floats) sp = WinoSplit{1, Ci, 0};      // no room for the partial outputs: one block per tile range
    // (one-wave blocks of 16 tiles for the planes that cannot fill the chip with 64-tile blocks were measured and lose: every
    // wave then stages the weight slabs for itself -- 12 x 256 -> 256 @ 16x52: 222 us against 112)
    const int tpb = 64;
    const long nblk = (ntiles + tpb - 1) / tpb * nkt * sp.nsp;
    if (nblk >= (1L << 31)) return DFE_ERR_DIMS;
    static const bool pair_ok = [] { const char* e = getenv("DFE_WINO_PAIR"); return !e || atoi(e) != 0; }();
    const bool pair = pair_ok && dil == 1 && W % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;
    const unsigned nitems = static_cast<unsigned>(nblk);
    // DFE_WINO_PERSIST=512: one resident round of persistent blocks (two per CU x 256 CUs).  Default 0 = one block per item: the
    // persistent form is 1-2.5 % faster per layer on an idle GPU and 0.08 ms SLOWER in the training step (18.93 against 18.85 ms,
    // three alternating pairs on one box) -- resident blocks leave the other streams' kernels no slot to slip into
    static const int persist = [] { const char* e = getenv("DFE_WINO_PERSIST"); return e ? atoi(e) : 0; }();
    const unsigned g = persist > 0 ? std::min(nitems, static_cast<unsigned>(persist)) : nitems;
    const int nt = static_cast<int>(ntiles);
    const int pp = pair ? P : -1;
#define WN_LAUNCH(PPV, NHV) k_wino_fwd16<PPV, 4, NHV><<<g, 256, lds_bytes, st>>>(x, wbuf, y, y_batch_stride, B, Ci, Co, H, W, P, TH, TW, nt, nkt, dil, static_cast<unsigned>(sp.nsp), sp.cps, part, epi, nitems)
    if (Co <= 16) {
      if (pp == 1) WN_LAUNCH(1, 1); else if (pp == 0) WN_LAUNCH(0, 1); else if (pp == 2) WN_LAUNCH(2, 1); else WN_LAUNCH(-1, 1);
    } else {
      if (pp == 1) WN_LAUNCH(1, 2); else if (pp == 0) WN_LAUNCH(0, 2); else if (pp == 2) WN_LAUNCH(2, 2); else WN_LAUNCH(-1, 2);
    }
#undef WN_LAUNCH
    DFE_LAUNCH_CHECK();
    if (sp.nsp > 1) {
      const long khw = static_cast<long>(Co) * Ho * Wo, n = B * khw;
      k_wino_sum<<<static_cast<unsigned>((n + 255) / 256), 256, 0, st>>>(part, y, y_batch_stride, sp.nsp, B, khw, static_cast<long>(Ho) * Wo, epi);
    }
  }
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" long dfe_wino_scratch_floats(int B, int Ci, int Co, int H, int W, int P) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || P < 0 || P > 2) return 0;
  const int Ho = H + 2 * P - 2, Wo = W + 2 * P - 2;
  if (Ho < 1 || Wo < 1) return 0;
  const int Kpad = (Co + 31) / 32 * 32;
  const long ntiles = static_cast<long>(B) * ((Ho + 1) / 2) * ((Wo + 1) / 2);
  return static_cast<long>(Kpad) * Ci * 16 + wino_split(ntiles, Kpad, Ci, B, Co, Ho, Wo).part_floats;
}

extern "C" int dfe_wino_conv3x3(const float* x, const float* weight, float* y, long y_batch_stride, float* wbuf, long wbuf_floats, int B,
                                int Ci, int Co, int H, int W, int P, int transposed_weight, void* stream) {
  if (!weight) return DFE_ERR_NULL;
  const long ufl = dfe_wino_weight_floats(Ci, Co);
  if (wbuf_floats < ufl) return DFE_ERR_WORKSPACE;
  return wino_run(x, weight, y, y_batch_stride, wbuf, wbuf ? wbuf + ufl : nullptr, wbuf_floats - ufl, B, Ci, Co, H, W, P, 1, transposed_weight, stream);
}

extern "C" int dfe_wino_conv3x3_dilated(const float* x, const float* weight, float* y, long y_batch_stride, float* wbuf, int B,
                                        int Ci, int Co, int H, int W, int dilation, int transposed_weight, void* stream) {
  if (!weight) return DFE_ERR_NULL;
  return wino_run(x, weight, y, y_batch_stride, wbuf, nullptr, 0, B, Ci, Co, H, W, 1, dilation, transposed_weight, stream);
}

extern "C" int dfe_wino_conv3x3_u(const float* x, const float* U, float* y, long y_batch_stride, float* part, long part_floats, int B, int Ci,
                                  int Co, int H, int W, int P, int dilation, void* stream) {
  return wino_run(x, nullptr, y, y_batch_stride, const_cast<float*>(U), part, part_floats, B, Ci, Co, H, W, dilation > 1 ? 1 : P, dilation, 0,
                  stream);
}

extern "C" int dfe_wino_conv3x3_u_act(const float* x, const float* U, const float* bias, float slope, float* y, long y_batch_stride, float* y2,
                                      long y2_batch_stride, float* part, long part_floats, int B, int Ci, int Co, int H, int W, int P,
                                      int dilation, void* stream) {
  return wino_run(x, nullptr, y, y_batch_stride, const_cast<float*>(U), part, part_floats, B, Ci, Co, H, W, dilation > 1 ? 1 : P, dilation, 0,
                  stream, WinoEpi{bias, slope, y2, y2_batch_stride});
}

extern "C" long dfe_wino_transform_blocks(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  return (static_cast<long>((Co + 31) / 32 * 32) * Ci + 255) / 256;
}

extern "C" int dfe_wino_transform_weights_multi(const long* table, const int* blockmap, int n_blocks, void* stream) {
  if (!table || !blockmap) return DFE_ERR_NULL;
  if (n_blocks <= 0) return DFE_ERR_DIMS;
  k_wino_weights_multi<<<n_blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(table, blockmap);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
