// Forward / data-gradient pass of the decoder's thinnest full-resolution 3x3 convolutions (16 or 32 -> 16 channels at
// 256x832 / 128x416, 12 images) on the matrix cores.  MIOpen's implicit-GEMM kernels run these at 40 TFLOP/s (0.30 ms
// for 0.08 ms of HBM traffic at 16 -> 16); with M = 16 pixels, N = 16 output channels and K = 4 input channels per
// v_mfma_f32_16x16x4_f32 the whole weight tensor (9 * Ci/4 B operands) lives in registers and a wave only streams
// pixels:   out[b][co][y][x] = sum_{ci,ky,kx} w[co][ci][ky][kx] * in[b][ci][y + ky - P][x + kx - P]
// P = 0: the forward pass on a reflection-padded input [Hi,Wi] = [Ho+2,Wo+2];  P = 2: the data gradient, a "full"
// correlation of the zero-extended output gradient with the flipped, transposed weights (prepared by the caller).
// A wave computes a 16-pixel x 4-row block per step and walks along x: the 6 input rows it needs are loaded once for
// 4 output rows (L2 -> L1 traffic, not the MFMA rate, is what bounds these layers); the accumulators of a block are
// 4 registers per row.  Lane (m, kq): A operand = in[ci = 4c + kq][row][x0 + m + kx], D = out[co = m][y][x0 + 4kq + r].
#include "dfe_internal.h"
#include <hip/hip_runtime.h>

namespace dfe {

typedef float tc_f32x4 __attribute__((ext_vector_type(4)));
constexpr int TC_ROWS = 4;
struct __attribute__((aligned(8))) TcF2 { float a, b; };

// grid: x = (b * nrb + row block) * nseg + segment; block = one wave.  NC4 = Ci / 4.
template <int NC4>
__global__ void __launch_bounds__(64) k_thin_conv3x3(const float* __restrict__ in, const float* __restrict__ wgt,
                                                     float* __restrict__ out, int Hi, int Wi, int Ho, int Wo, int P, int nrb,
                                                     int nseg, int tiles_per_seg, int flip) {
  constexpr int Ci = NC4 * 4, Co = 16;
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  const int seg = blockIdx.x % nseg, br = blockIdx.x / nseg;
  const int b = br / nrb, rb = br - b * nrb;
  const int y0 = rb * TC_ROWS;
  // B operands: w[co = m][ci = 4c + kq][tap]
  float w[NC4][9];
#pragma unroll
  for (int c = 0; c < NC4; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      w[c][t] = flip ? wgt[(static_cast<long>(4 * c + kq) * Co + m) * 9 + (8 - t)]      // w'[co'=m][ci'=4c+kq][t] of the data gradient
                     : wgt[(static_cast<long>(m) * Ci + 4 * c + kq) * 9 + t];
  // wave-uniform sample bases + 32-bit per-lane offsets: the loads are plain `global_load v, voff, s[base]` with no
  // 64-bit address arithmetic, and they are unconditional (clamped address, value zeroed afterwards) -- predicating
  // each load with an exec-mask branch (66 s_and_saveexec / s_cbranch per tile) serialised their issue and cost
  // 168 VGPRs (3 waves per SIMD): MFMA pipe 46 % busy, 74 % of wave-cycles waiting for issue (PMC, round 2).
  // Measured and rejected in round 2: a 64-pixel-tile variant (a lane loads six consecutive pixels with three 8-byte
  // loads, MFMA #j takes the pixel set {x0 + 4m + j}, kx operands are registers j + kx of the same lane: no DPP, wider
  // loads, interior / border tiles in separate instantiations) -- 158-169 us against 157 us here at 16 -> 16, 256x832:
  // 156-224 VGPRs leave 2-3 waves per SIMD, and rows of the padded activation are only 8-byte aligned, so the load
  // count per output row does not drop.  Ablation of this kernel: no loads 134 us, no MFMA 146 us, neither 44 us.
  const float* sb = in + static_cast<long>(b) * Ci * Hi * Wi;
  const unsigned plane = static_cast<unsigned>(Hi) * static_cast<unsigned>(Wi);
  float* ob = out + static_cast<long>(b) * Co * Ho * Wo;
  const int ntx = (Wo + 15) / 16;
  const int t0 = seg * tiles_per_seg, t1 = min(t0 + tiles_per_seg, ntx);
  // One load per (channel chunk, input row) and 16-pixel tile: lane m holds in[..][x0 + m - P]; the kx = 1, 2 operands
  // are that row shifted left inside the 16-lane group (DPP row_shl) with the first lanes of the NEXT tile's load
  // shifted in at the end (row_shr 15 / 14) -- the texture unit, one per CU for four MFMA pipes, would otherwise be
  // as busy as the matrix cores.  The next tile's loads are issued before the current tile's MFMAs.
  auto fetch = [&](int tx, float (&v)[NC4][TC_ROWS + 2]) {
    const int xx = tx * 16 + m - P;
    const bool colok = xx >= 0 && xx < Wi;
    const unsigned xo = static_cast<unsigned>(min(max(xx, 0), Wi - 1)) + static_cast<unsigned>(kq) * plane;
#pragma unroll
    for (int iy = 0; iy < TC_ROWS + 2; ++iy) {
      const int yy = y0 + iy - P;
      const bool ok = colok && yy >= 0 && yy < Hi;
      const unsigned off = (static_cast<unsigned>(min(max(yy, 0), Hi - 1)) * static_cast<unsigned>(Wi) + xo) * 4u;
#pragma unroll
      for (int c = 0; c < NC4; ++c) {
        const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb + static_cast<long>(4 * c) * plane) + off);
        v[c][iy] = ok ? t : 0.0f;
      }
    }
  };
  float cur[NC4][TC_ROWS + 2], nxt[NC4][TC_ROWS + 2];
  if (t0 < t1) fetch(t0, cur);
  for (int tx = t0; tx < t1; ++tx) {
    const int x0 = tx * 16;
    fetch(tx + 1, nxt);                  // one tile beyond the segment / image: predicated to zero where outside
    tc_f32x4 acc[TC_ROWS];
#pragma unroll
    for (int r = 0; r < TC_ROWS; ++r) acc[r] = tc_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NC4; ++c) {
#pragma unroll
      for (int iy = 0; iy < TC_ROWS + 2; ++iy) {
        const int a = __float_as_int(cur[c][iy]), n = __float_as_int(nxt[c][iy]);
        float v[3];
        v[0] = cur[c][iy];
        v[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, a, 0x101, 0xF, 0xF, true) | __builtin_amdgcn_update_dpp(0, n, 0x11F, 0xF, 0xF, true));
        v[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, a, 0x102, 0xF, 0xF, true) | __builtin_amdgcn_update_dpp(0, n, 0x11E, 0xF, 0xF, true));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int r = 0; r < TC_ROWS; ++r) {
            const int ky = iy - r;
            if (ky < 0 || ky > 2) continue;
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[kx], w[c][ky * 3 + kx], acc[r], 0, 0, 0);
          }
      }
    }
    // D[i = 4 kq + j][co = m]
#pragma unroll
    for (int r = 0; r < TC_ROWS; ++r) {
      const int y = y0 + r;
      if (y >= Ho) continue;
      float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(ob) + ((static_cast<unsigned>(m) * Ho + y) * static_cast<unsigned>(Wo) + x0 + 4 * kq) * 4u);
      if (x0 + 16 <= Wo && (Wo & 3) == 0) *reinterpret_cast<tc_f32x4*>(o) = acc[r];
      else if (x0 + 16 <= Wo && (Wo & 1) == 0) {       // rows of the padded gradient start 8-byte aligned
        *reinterpret_cast<TcF2*>(o) = TcF2{acc[r][0], acc[r][1]};
        *reinterpret_cast<TcF2*>(o + 2) = TcF2{acc[r][2], acc[r][3]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (x0 + 4 * kq + j < Wo) o[j] = acc[r][j];
      }
    }
#pragma unroll
    for (int c = 0; c < NC4; ++c)
#pragma unroll
      for (int iy = 0; iy < TC_ROWS + 2; ++iy) cur[c][iy] = nxt[c][iy];
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

extern "C" int dfe_thin_conv3x3(const float* in, const float* weight, float* out, int B, int Ci, int Co, int Hi, int Wi, int P,
                                int transposed_weight, void* stream) {
  if (!in || !weight || !out) return DFE_ERR_NULL;
  if (B <= 0 || Hi <= 0 || Wi <= 0 || P < 0 || P > 2) return DFE_ERR_DIMS;
  if (Co != 16 || (Ci != 16 && Ci != 32)) return DFE_ERR_UNSUPPORTED;
  if (transposed_weight && Ci != 16) return DFE_ERR_UNSUPPORTED;      // weight [16 (= Ci here)][16 (= Co here)][3][3] read transposed
  const int Ho = Hi + 2 * P - 2, Wo = Wi + 2 * P - 2;
  if (Ho <= 0 || Wo <= 0 || static_cast<long>(Hi) * Wi * Ci >= (1L << 30) || static_cast<long>(Ho) * Wo * Co >= (1L << 30)) return DFE_ERR_DIMS;   // 32-bit byte offsets per sample
  if (reinterpret_cast<uintptr_t>(out) & 15) return DFE_ERR_UNSUPPORTED;
  const int nrb = (Ho + TC_ROWS - 1) / TC_ROWS, ntx = (Wo + 15) / 16;
  int nseg = 1;                 // ~4096 waves: split the rows of tiles into segments when there are few row blocks
  while (nseg < ntx && static_cast<long>(B) * nrb * nseg < 4096) nseg *= 2;
  if (nseg > ntx) nseg = ntx;
  const int tps = (ntx + nseg - 1) / nseg;
  nseg = (ntx + tps - 1) / tps;
  const long units = static_cast<long>(B) * nrb * nseg;
  if (units >= (1L << 31)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Ci == 16) k_thin_conv3x3<4><<<static_cast<unsigned>(units), 64, 0, st>>>(in, weight, out, Hi, Wi, Ho, Wo, P, nrb, nseg, tps, transposed_weight);
  else k_thin_conv3x3<8><<<static_cast<unsigned>(units), 64, 0, st>>>(in, weight, out, Hi, Wi, Ho, Wo, P, nrb, nseg, tps, transposed_weight);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
