// Correctly rounded short fp32 sequences and the per-scale launch constants of the pointwise kernels.
//
// Cost model measured on MI355X (profiles/r02_issue_cost_model.md): instruction issue is ADDITIVE per CU --
//   vector-memory instruction (any width <= 16 B, coalesced)   ~14-16 CU-cycles   (a spatially coherent 8-byte gather ~22)
//   LDS read (ds_read_b32 / ds_read2_b32 / ds_read_b64)        ~1.2-1.8 CU-cycles
//   VALU instruction                                            ~0.45 CU-cycles (1.8 SIMD-cycles), packed fp32 no faster
// VALU work does not hide vector-memory issue, so both instruction counts are what the pointwise kernels pay for.
#pragma once
#include "loss_stack.h"

namespace dfe {

// ---- correctly rounded fp32 reciprocal / division / square root in 3 / 3 / 5 instructions.
// Verified EXHAUSTIVELY on MI355X against the IEEE sequences (dfe_exact_math_selftest, tests/test_hip_ops.py):
//   rcp_cr(z)      == 1.0f / z     for every z with 2^-125 <= |z| < 2^126;
//   sqrt_cr(x)     == sqrtf(x)     for every x >= 2^-97;
//   div_cr(x,z,r)  == x / z        (Markstein's theorem with r = RN(1/z); 2^34 random pairs + every divisor with an
//                                   all-ones significand checked) for finite normal quotients.
// hipcc expands the IEEE operators to 12 (division) and 18 (square root) instructions.
__device__ __forceinline__ float rcp_cr(float z) {
  const float r0 = __builtin_amdgcn_rcpf(z);
  return __fmaf_rn(__fmaf_rn(-z, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_cr(float x, float z, float r) {
  const float q = x * r;
  return __fmaf_rn(__fmaf_rn(-z, q, x), r, q);
}
__device__ __forceinline__ float sqrt_cr(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  return __fmaf_rn(__fmaf_rn(-y, y, x), h, y);
}
constexpr float SQRT_CR_MIN = 0x1p-97f;

// per-scale constants computed on the host (a uniform 1 / (W - 1) costs every thread a 12-instruction IEEE division)
struct GeomT {
  float rW[DFE_MAX_SCALES];                           // 1 / W_s: pixel index -> row without an integer division
  Divisor dw[DFE_MAX_SCALES], dh[DFE_MAX_SCALES];     // exact division by W_s - 1, H_s - 1 (dfe_device.h div_exact)
};
void tile_dev(const GeomLayout& L, GeomT* T);

// y = floor((p + 0.5) / W) through the reciprocal: exact for p < 2^23 and W < 8192 (p + 0.5 is at least 0.5 / W away
// from a multiple of W in units of 1 / W while the product's error stays below 2^-23 * H)
__device__ __forceinline__ void split_pixel(unsigned p, int W, float rW, unsigned& px, unsigned& py) {
  py = static_cast<unsigned>((static_cast<float>(p) + 0.5f) * rW);
  px = p - py * static_cast<unsigned>(W);
}

// quotient estimate q ~ p / d from a reciprocal multiply, made exact by one integer step either way (the float
// estimate's margin is ~0.5 / p relative; a wrong row GROUP would break the tile map's bijection silently as images grow)
__device__ __forceinline__ unsigned fix_quotient(unsigned q, unsigned p, unsigned d) {
  const unsigned lo = q * d;
  return p < lo ? q - 1u : (p - lo >= d ? q + 1u : q);
}

// Pixel order of the pointwise kernels.  TW = 0: thread p of a scale computes pixel p (a wave = 64 consecutive pixels of a
// row).  TW = 8 / 16 / 32: a wave covers a TW x (64 / TW) tile -- the bilinear taps of both warps then fall into a few
// cache lines of 5 rows instead of 2 x 64-pixel row segments shifted by the flow, which is what the texture path is busy
// with (round 4: k_geom_point_fwd 48.9 -> 43.3 us at B = 4, 616 -> 508 us at 375x1242 B = 16; results unchanged, the
// per-block partial sums cover other pixels but add up to the same totals).  The linear index is re-read as (group of TR
// rows, TW-column chunk, row in group, column in chunk); columns past the last full chunk and rows past the last full group
// keep the row-major order, so the map is a bijection of [0, H*W) for every size.  rW = 1 / W as split_pixel wants it.
template <int TW>
__device__ __forceinline__ void tile_pixel(unsigned p, int W, int H, float rW, unsigned& px, unsigned& py) {
  if (TW == 0) { split_pixel(p, W, rW, px, py); return; }
  constexpr unsigned TWu = TW > 0 ? TW : 1, TR = 64u / TWu, LG = TWu == 8 ? 3u : TWu == 16 ? 4u : 5u;
  static_assert(TW == 0 || TW == 8 || TW == 16 || TW == 32, "tile width");
  const unsigned Wu = static_cast<unsigned>(W);
  const unsigned g = fix_quotient(static_cast<unsigned>((static_cast<float>(p) + 0.5f) * (rW * (1.0f / static_cast<float>(TR)))), p, TR * Wu);   // p / (TR * W)
  if (g < static_cast<unsigned>(H) / TR) {
    const unsigned q = p - g * TR * Wu, WT = Wu & ~(TWu - 1u);
    if (q < TR * WT) { px = TWu * (q >> 6) + (q & (TWu - 1u)); py = TR * g + ((q & 63u) >> LG); }
    else {
      const unsigned e = q - TR * WT, wr = Wu - WT;
      const unsigned r = fix_quotient(static_cast<unsigned>((static_cast<float>(e) + 0.5f) / static_cast<float>(wr)), e, wr);
      px = WT + e - r * wr; py = TR * g + r;
    }
  } else { split_pixel(p, W, rW, px, py); py = fix_quotient(py, p, Wu); px = p - py * Wu; }
}

// the same map with integer divisions (kernels that carry no reciprocal table)
template <int TW>
__device__ __forceinline__ void tile_pixel(unsigned p, int W, int H, unsigned& px, unsigned& py) {
  const unsigned Wu = static_cast<unsigned>(W);
  if (TW == 0) { py = p / Wu; px = p - py * Wu; return; }
  constexpr unsigned TWu = TW > 0 ? TW : 1, TR = 64u / TWu, LG = TWu == 8 ? 3u : TWu == 16 ? 4u : 5u;
  const unsigned g = p / (TR * Wu);
  if (g < static_cast<unsigned>(H) / TR) {
    const unsigned q = p - g * TR * Wu, WT = Wu & ~(TWu - 1u);
    if (q < TR * WT) { px = TWu * (q >> 6) + (q & (TWu - 1u)); py = TR * g + ((q & 63u) >> LG); }
    else {
      const unsigned e = q - TR * WT, wr = Wu - WT, r = e / wr;
      px = WT + e - r * wr; py = TR * g + r;
    }
  } else { py = p / Wu; px = p - py * Wu; }
}

// projection with the correctly rounded short sequences: X / Z and Y / Z share RN(1 / Z) (same bits as project())
__device__ __forceinline__ Proj project_fast(const Camera& c, int x, int y, float depth) {
  Proj p;
  const float fx = static_cast<float>(x), fy = static_cast<float>(y);
  p.r0 = __fmaf_rn(c.kinv[1], fy, c.kinv[0] * fx) + c.kinv[2];
  p.r1 = __fmaf_rn(c.kinv[4], fy, c.kinv[3] * fx) + c.kinv[5];
  p.r2 = __fmaf_rn(c.kinv[7], fy, c.kinv[6] * fx) + c.kinv[8];
  const float c0 = p.r0 * depth, c1 = p.r1 * depth, c2 = p.r2 * depth;
  const float X = __fmaf_rn(c.A[2], c2, __fmaf_rn(c.A[1], c1, c.A[0] * c0)) + c.b[0];
  const float Y = __fmaf_rn(c.A[5], c2, __fmaf_rn(c.A[4], c1, c.A[3] * c0)) + c.b[1];
  const float Zr = __fmaf_rn(c.A[8], c2, __fmaf_rn(c.A[7], c1, c.A[6] * c0)) + c.b[2];
  p.q0 = c.A[0] * p.r0 + c.A[1] * p.r1 + c.A[2] * p.r2;   // Jacobian factors (backward only; dead code in the forward)
  p.q1 = c.A[3] * p.r0 + c.A[4] * p.r1 + c.A[5] * p.r2;
  p.q2 = c.A[6] * p.r0 + c.A[7] * p.r1 + c.A[8] * p.r2;
  p.clamped = !(Zr >= 1e-3f);
  p.Z = (Zr >= 1e-3f) ? Zr : 1e-3f;   // clamp(min=1e-3); NaN propagates like torch.clamp
  if (Zr != Zr) p.Z = Zr;
  const float rz = rcp_cr(p.Z);
  p.U = div_cr(X, p.Z, rz);
  p.V = div_cr(Y, p.Z, rz);
  return p;
}

// (1 - softmax([dl, dr])) > 0.48 with e / sum through RN(1 / sum): same bits as occ_weights()
__device__ __forceinline__ void occ_decide(float dl, float dr, bool& occ_bwd, bool& occ_fwd) {
  const float tt = occ_exp(fabsf(dl - dr));
  const float el = (dl >= dr) ? 1.0f : tt, er = (dr >= dl) ? 1.0f : tt;
  const float sum = el + er, rs = rcp_cr(sum);
  occ_bwd = (1.0f - div_cr(el, sum, rs)) > 0.48f;
  occ_fwd = (1.0f - div_cr(er, sum, rs)) > 0.48f;
}

// sqrt_cr for any finite x >= 0: arguments below 2^-97 (the residual of sqrt_cr would go denormal) are scaled by 2^128
// (an even power of two: the scaled root, times 2^-64, is the same correctly rounded value), zero maps to zero
__device__ __forceinline__ float sqrt_cr_full(float x) {
  const bool tiny = x < SQRT_CR_MIN;
  const float r = sqrt_cr(tiny ? x * 0x1p+100f * 0x1p+28f : x);
  return (x == 0.0f) ? 0.0f : (tiny ? r * 0x1p-64f : r);
}

// dyna_decision() with the short square roots (same bits for finite inputs)
__device__ __forceinline__ bool dyna_decide(float fu, float fv, float ru, float rv, float du, float dv, float alpha, float beta) {
  const float nf = sqrt_cr_full(fu * fu + fv * fv) + 1e-12f, nr = sqrt_cr_full(ru * ru + rv * rv) + 1e-12f;
  const float nd = sqrt_cr_full(du * du + dv * dv) + 1e-12f;
  return (nd * nd) < (alpha * (nf * nf + nr * nr) + beta);
}

}  // namespace dfe
