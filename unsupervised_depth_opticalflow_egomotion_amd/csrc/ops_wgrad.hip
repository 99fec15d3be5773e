// Weight gradient of the thin, wide 3x3 convolutions of the depth decoder (reference depth_model.py: the last two
// decoder stages, 16..96 channels at 128x416 / 256x832, on reflection-padded inputs):
//     dW[co][ci][ky][kx] = sum_{b,y,x} gy[b][co][y][x] * p[b][ci][y+ky][x+kx]        p [B,Ci,H+2,W+2], gy [B,Co,H,W]
// This is a true dense contraction with a tiny output (Co x 9Ci) and an enormous reduction length (B*H*W, 0.6-2.6 M):
// MIOpen's implicit-GEMM weight-gradient kernels run it at 23-56 TFLOP/s after transposing both operands to NHWC
// (0.50 ms at 16->16 / 256x832 for 0.08 ms of HBM traffic).  Here it goes to the matrix cores directly from NCHW:
// v_mfma_f32_16x16x4_f32 with M = 16 output channels, N = 16 input channels, K = 4 pixels per instruction; a lane's
// A operand is one float4 of a gy row (4 consecutive pixels of its channel, 4 K-steps), its B operand three float2 of
// the matching p row (the 4 pixels shifted by kx = 0, 1, 2).  A wave owns a slab of R image rows x a column segment,
// one 16-channel tile of Co and one of Ci, and keeps the 9 accumulator tiles in registers for its whole slab; per-wave
// partial sums are added in a fixed order by a second kernel (no atomics: reproducible).  fp32 MFMA is an exact fma chain.
// Bound: MFMA (157 TFLOP/s fp32) for Ci >= 32, HBM below.
#include "dfe_internal.h"
#include <hip/hip_runtime.h>

namespace dfe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(8))) F2 { float a, b; };

// grid: x = unit = (b * nrg + row group) * nseg + column segment, y = group of NCI ci tiles, z = group of NCO co tiles;
// block = one wave holding NCO x NCI tile pairs (9 accumulator tiles each).  (1,1) keeps a wave light (36 accumulator
// registers, many waves per SIMD) for the one- or two-tile layers; for many tiles (96->32 = 2 x 6) a wave takes 2 x 3
// pairs so that every gy / p operand it loads feeds 3 / 2 tiles -- the L2 -> L1 traffic, not the MFMA rate, is what
// bounds the one-pair-per-wave layout there.  The operands of step i+1 are in flight while the MFMAs of step i run.
template <int NCO, int NCI>
__global__ void __launch_bounds__(64) k_wgrad3x3_thin(const float* __restrict__ p, const float* __restrict__ gy,
                                                      float* __restrict__ part, int Ci, int Co, int H, int W, int R, int nrg,
                                                      int nseg) {
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  const int seg = blockIdx.x % nseg, br = blockIdx.x / nseg;
  const int b = br / nrg, rg = br - b * nrg;
  const int ci0 = blockIdx.y * 16 * NCI, co0 = blockIdx.z * 16 * NCO;
  const int Hp = H + 2, Wp = W + 2;
  const int y0 = rg * R, y1 = min(y0 + R, H);
  const int nch = W / 16, cps = (nch + nseg - 1) / nseg;
  const int xa = seg * cps * 16, xb = min((seg + 1) * cps, nch) * 16;
  const float* ga = gy + (static_cast<long>(b) * Co + co0 + m) * H * W + 4 * kq;
  const float* pa = p + (static_cast<long>(b) * Ci + ci0 + m) * Hp * Wp + 4 * kq;
  const long gtile = 16L * H * W, ptile = 16L * Hp * Wp;
  f32x4 acc[NCO][NCI][9];
#pragma unroll
  for (int o = 0; o < NCO; ++o)
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o][t][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int per_row = (xb - xa) / 16, nsteps = max(y1 - y0, 0) * per_row;
  f32x4 a4[NCO];
  F2 e[NCI][3][3];
  auto fetch = [&](int step, f32x4 (&av)[NCO], F2 (&ev)[NCI][3][3]) {
    const int yy = y0 + step / per_row, x0 = xa + (step % per_row) * 16;
#pragma unroll
    for (int o = 0; o < NCO; ++o) av[o] = *reinterpret_cast<const f32x4*>(ga + o * gtile + static_cast<long>(yy) * W + x0);
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* q = pa + t * ptile + static_cast<long>(yy + ky) * Wp + x0;
        ev[t][ky][0] = *reinterpret_cast<const F2*>(q); ev[t][ky][1] = *reinterpret_cast<const F2*>(q + 2);
        ev[t][ky][2] = *reinterpret_cast<const F2*>(q + 4);
      }
  };
#pragma unroll
  for (int o = 0; o < NCO; ++o) a4[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (nsteps > 0) fetch(0, a4, e);
  for (int step = 0; step < nsteps; ++step) {
    f32x4 an[NCO];
    F2 en[NCI][3][3];
#pragma unroll
    for (int o = 0; o < NCO; ++o) an[o] = a4[o];
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < 3; ++j) en[t][ky][j] = e[t][ky][j];
    if (step + 1 < nsteps) fetch(step + 1, an, en);
    // K-step s outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < NCI; ++t)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float v[6] = {e[t][ky][0].a, e[t][ky][0].b, e[t][ky][1].a, e[t][ky][1].b, e[t][ky][2].a, e[t][ky][2].b};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < NCO; ++o)
              acc[o][t][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[o][s], v[s + kx], acc[o][t][ky * 3 + kx], 0, 0, 0);
        }
#pragma unroll
    for (int o = 0; o < NCO; ++o) a4[o] = an[o];
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < 3; ++j) e[t][ky][j] = en[t][ky][j];
  }
  // D[i][j]: lane holds rows i = 4*kq + r (output channel), column j = m (input channel).  The partial plane is laid
  // out [tap][co][ci] so that the 16 lanes of a row write 64 contiguous bytes; k_wgrad_final permutes to [co][ci][tap].
  float* po = part + static_cast<long>(blockIdx.x) * Co * Ci * 9;
#pragma unroll
  for (int o = 0; o < NCO; ++o)
#pragma unroll
    for (int t = 0; t < NCI; ++t)
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          po[(static_cast<long>(k) * Co + co0 + 16 * o + 4 * kq + r) * Ci + ci0 + 16 * t + m] = acc[o][t][k][r];
}

// dW[idx] = sum over units of part[u][idx].  Block = 16 consecutive idx x 64 unit lanes: every thread adds the units
// u = lane, lane + 64, ... (four independent partial sums keep four loads in flight), then the 64 lane sums are added
// in lane order (fixed order: reproducible).
constexpr int WF_LANES = 64;
__global__ void __launch_bounds__(16 * WF_LANES) k_wgrad_final(const float* __restrict__ part, float* __restrict__ gw, int n,
                                                               int nunits, int coci) {
  __shared__ float sm[WF_LANES][17];
  const int ii = threadIdx.x & 15, ul = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ii;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (i < n) {
    const float* q = part + i;
    int u = ul;
    for (; u + 3 * WF_LANES < nunits; u += 4 * WF_LANES) {
      s0 += q[static_cast<long>(u) * n]; s1 += q[static_cast<long>(u + WF_LANES) * n];
      s2 += q[static_cast<long>(u + 2 * WF_LANES) * n]; s3 += q[static_cast<long>(u + 3 * WF_LANES) * n];
    }
    for (; u < nunits; u += WF_LANES) s0 += q[static_cast<long>(u) * n];
  }
  sm[ul][ii] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ul == 0 && i < n) {
    float t = 0.0f;
    for (int k = 0; k < WF_LANES; ++k) t += sm[k][ii];
    const int tap = i / coci, cc = i - tap * coci;          // partial planes are [tap][co][ci]
    gw[static_cast<long>(cc) * 9 + tap] = t;
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

// Work split: units = B * ceil(H / R) * nseg waves per tile pair.  Measured on MI355X (12 images): with one or two tile
// wave groups (16->16, 32->16, and 96->32 as 2 groups of 2 x 3 pairs) ~1024 waves in total are fastest -- more waves cost more in partial-plane traffic
// (units * Co*Ci*9 floats written and re-read) than they gain in occupancy; with many tile pairs (96->32) ~4096.
// Columns are split into up to 4 segments when there are few rows, rows are grouped when there are many.
struct WgSplit { int R, nrg, nseg; long units; };
static WgSplit wg_split(int B, int H, int W, int tiles) {
  WgSplit s{1, H, 1, 0};
  const long total = tiles <= 2 ? 1024 : 4096;
  const long target = total / tiles > 0 ? total / tiles : 1;
  while (s.R < H && static_cast<long>(B) * ((H + s.R - 1) / s.R) > target) s.R *= 2;
  s.nrg = (H + s.R - 1) / s.R;
  while (s.nseg < 4 && s.nseg * 2 <= W / 16 && static_cast<long>(B) * s.nrg * s.nseg * 2 <= target) s.nseg *= 2;
  s.units = static_cast<long>(B) * s.nrg * s.nseg;
  return s;
}

static int wg_dims(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  if (Ci % 16 != 0 || Co % 16 != 0 || W % 16 != 0 || Ci / 16 > 65535 || Co / 16 > 65535) return DFE_ERR_UNSUPPORTED;
  if ((static_cast<long>(H) + 2) * (W + 2) * Ci >= (1L << 31) || static_cast<long>(H) * W * Co >= (1L << 31)) return DFE_ERR_DIMS;
  return DFE_OK;
}

// tile pairs per wave: one for the one- / two-tile layers; 2 x 3 when the tile grid divides (96 -> 32)
static void wg_group(int Ci, int Co, int& nco, int& nci) {
  const int tci = Ci / 16, tco = Co / 16;
  nco = 1; nci = 1;
  if (tci * tco >= 6 && tco % 2 == 0 && tci % 3 == 0) { nco = 2; nci = 3; }
  else if (tci * tco >= 4 && tco % 2 == 0 && tci % 2 == 0) { nco = 2; nci = 2; }
}

extern "C" long dfe_wgrad3x3_partials_floats(int B, int Ci, int Co, int H, int W) {
  if (wg_dims(B, Ci, Co, H, W) != DFE_OK) return 0;
  int nco, nci;
  wg_group(Ci, Co, nco, nci);
  return wg_split(B, H, W, (Ci / (16 * nci)) * (Co / (16 * nco))).units * Co * Ci * 9;
}

extern "C" int dfe_wgrad3x3_fwd(const float* p, const float* gy, float* gweight, float* partials, int B, int Ci, int Co, int H,
                                int W, void* stream) {
  if (!p || !gy || !gweight || !partials) return DFE_ERR_NULL;
  const int rc = wg_dims(B, Ci, Co, H, W);
  if (rc != DFE_OK) return rc;
  if ((reinterpret_cast<uintptr_t>(p) & 7) || (reinterpret_cast<uintptr_t>(gy) & 15)) return DFE_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int nco, nci;
  wg_group(Ci, Co, nco, nci);
  const WgSplit sp = wg_split(B, H, W, (Ci / (16 * nci)) * (Co / (16 * nco)));
  const dim3 grid(static_cast<unsigned>(sp.units), Ci / (16 * nci), Co / (16 * nco));
  if (nco == 2 && nci == 3) k_wgrad3x3_thin<2, 3><<<grid, 64, 0, st>>>(p, gy, partials, Ci, Co, H, W, sp.R, sp.nrg, sp.nseg);
  else if (nco == 2) k_wgrad3x3_thin<2, 2><<<grid, 64, 0, st>>>(p, gy, partials, Ci, Co, H, W, sp.R, sp.nrg, sp.nseg);
  else k_wgrad3x3_thin<1, 1><<<grid, 64, 0, st>>>(p, gy, partials, Ci, Co, H, W, sp.R, sp.nrg, sp.nseg);
  DFE_LAUNCH_CHECK();
  const int n = Co * Ci * 9;
  k_wgrad_final<<<(n + 15) / 16, 16 * WF_LANES, 0, st>>>(partials, gweight, n, static_cast<int>(sp.units), Co * Ci);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
