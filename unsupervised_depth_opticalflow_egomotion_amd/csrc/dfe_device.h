// Shared device-side arithmetic for the loss-stack kernels (gfx950, wave64).
//
// Every expression that feeds a {0,1} mask decision is written in the association order
// the reference's ATen CPU path evaluates it in (probed bit-for-bit in the build
// container): the bilinear coordinate is fma(g+1, size/2, -0.5) (align_corners=False) or
// (g+1)*((size-1)/2) (True), the four weights are rounded products and the interpolation
// is the chain t=v_nw*nw; t=fma(v_ne,ne,t); t=fma(v_sw,sw,t); t=fma(v_se,se,t).
// All translation units are compiled with -ffp-contract=off; FMAs appear only where
// written explicitly.
#pragma once
#include <hip/hip_runtime.h>

#define DFE_WAVE 64

namespace dfe {

// ---------------------------------------------------------------- bilinear sampling
// grid_sample(bilinear, zeros padding) geometry for one sample point.
struct Tap {
  int x0, y0;          // north-west corner (may be out of bounds)
  float nw, ne, sw, se; // weights (not yet masked by bounds)
  bool in_nw, in_ne, in_sw, in_se;
  float wx, wy;        // fractional offsets (w = ix - x0, n = iy - y0)
};

__device__ __forceinline__ float unnormalize(float g, int size, int align_corners) {
  // reference semantics: torch grid_sample; SURVEY.md A.1
  if (align_corners) return (g + 1.0f) * (static_cast<float>(size - 1) / 2.0f);
  return __fmaf_rn(g + 1.0f, static_cast<float>(size) / 2.0f, -0.5f);
}

__device__ __forceinline__ Tap make_tap(float ix, float iy, int H, int W) {
  Tap t;
  float xw = floorf(ix), yn = floorf(iy);
  float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
  t.wx = w; t.wy = n;
  t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
  // NaN / huge coordinates: comparisons below are false for NaN -> fully out of bounds.
  bool xin0 = (xw > -1.0f) && (xw < static_cast<float>(W));
  bool xin1 = (xw + 1.0f > -1.0f) && (xw + 1.0f < static_cast<float>(W));
  bool yin0 = (yn > -1.0f) && (yn < static_cast<float>(H));
  bool yin1 = (yn + 1.0f > -1.0f) && (yn + 1.0f < static_cast<float>(H));
  t.in_nw = xin0 && yin0; t.in_ne = xin1 && yin0; t.in_sw = xin0 && yin1; t.in_se = xin1 && yin1;
  // keep the int conversion defined for out-of-range coordinates
  float xc = fminf(fmaxf(xw, -2.0f), static_cast<float>(W) + 1.0f);
  float yc = fminf(fmaxf(yn, -2.0f), static_cast<float>(H) + 1.0f);
  t.x0 = (xw == xw) ? static_cast<int>(xc) : -2;
  t.y0 = (yn == yn) ? static_cast<int>(yc) : -2;
  return t;
}

// Sum of the in-bounds weights, in ATen's accumulation order (== grid_sample(ones)).
__device__ __forceinline__ float tap_cover(const Tap& t) {
  float c = t.in_nw ? t.nw : 0.0f;
  c = c + (t.in_ne ? t.ne : 0.0f);
  c = c + (t.in_sw ? t.sw : 0.0f);
  c = c + (t.in_se ? t.se : 0.0f);
  return c;
}

struct Corners { float nw, ne, sw, se; };

// Branch-free and pair-wise: each row of the 2x2 footprint is one dword-aligned 8-byte load at
// xs = clamp(x0, 0, W-2) (memory-instruction count is what bounds these kernels: the texture-address unit
// takes ~16 cycles per wave-wide load whatever its width); the values are then routed to (nw, ne) / (sw, se)
// and zeroed by the in-bounds flags -- same values as ATen's masked gather.  W == 1 falls back to scalars.
struct __attribute__((packed, aligned(4))) PairF { float a, b; };

__device__ __forceinline__ Corners load_corners(const float* __restrict__ plane, const Tap& t, int W, int H) {
  Corners c;
  const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
  if (W >= 2) {
    const int xs = min(max(t.x0, 0), W - 2);
    const bool lo = t.x0 < xs, hi = t.x0 > xs;
    const PairF r0 = *reinterpret_cast<const PairF*>(plane + static_cast<long>(ya) * W + xs);
    const PairF r1 = *reinterpret_cast<const PairF*>(plane + static_cast<long>(yb) * W + xs);
    c.nw = t.in_nw ? (hi ? r0.b : r0.a) : 0.0f;
    c.ne = t.in_ne ? (lo ? r0.a : r0.b) : 0.0f;
    c.sw = t.in_sw ? (hi ? r1.b : r1.a) : 0.0f;
    c.se = t.in_se ? (lo ? r1.a : r1.b) : 0.0f;
  } else {
    const float v0 = plane[static_cast<long>(ya) * W], v1 = plane[static_cast<long>(yb) * W];
    // single column: whichever corner is in bounds is column 0
    c.nw = t.in_nw ? v0 : 0.0f; c.ne = t.in_ne ? v0 : 0.0f; c.sw = t.in_sw ? v1 : 0.0f; c.se = t.in_se ? v1 : 0.0f;
  }
  return c;
}

__device__ __forceinline__ float interp(const Corners& c, const Tap& t) {
  float r = c.nw * t.nw;
  r = __fmaf_rn(c.ne, t.ne, r);
  r = __fmaf_rn(c.sw, t.sw, r);
  r = __fmaf_rn(c.se, t.se, r);
  return r;
}

// d(out)/d(ix), d(out)/d(iy) of the bilinear interpolation
__device__ __forceinline__ void interp_grad(const Corners& c, const Tap& t, float& dix, float& diy) {
  float s = 1.0f - t.wy, e = 1.0f - t.wx;
  dix = (c.ne - c.nw) * s + (c.se - c.sw) * t.wy;
  diy = (c.sw - c.nw) * e + (c.se - c.ne) * t.wx;
}

// ---------------------------------------------------------------- exact division by a known divisor
// Correctly rounded x / y in 3 instructions when r = RN(1/y) is available (Markstein): q = RN(x r);
// q' = RN(q + RN(x - q y) r).  Bit-identical to IEEE division for the divisors used here (3, 20, W-1, H-1:
// small integers, no all-ones significand) on normal-range operands; replaces ~11-instruction v_div_* chains.
struct Divisor { float y, r; };
__host__ __device__ __forceinline__ Divisor make_divisor(float y) { return Divisor{y, 1.0f / y}; }
__device__ __forceinline__ float div_exact(float x, const Divisor d) {
  const float q = x * d.r;
  return __fmaf_rn(__fmaf_rn(-d.y, q, x), d.r, q);
}

// ---------------------------------------------------------------- fast forward-only bilinear tap
// Same values as make_tap/load_corners/interp for finite inputs, fewer instructions: the coordinate is
// clamped to [-2, size+1] first (everything outside is fully out of bounds either way), bounds are integer
// compares, and the in-bounds mask is applied to the four WEIGHTS once instead of to 4 values per channel.
// Horizontal corner pairs are adjacent in memory, so each row of the 2x2 footprint is ONE 8-byte load
// (dword-aligned dwordx2): the texture-address unit handles 4 lanes/clk, i.e. a wave-wide load costs ~16 TA
// cycles whatever its width, and the kernel is bound by the number of memory instructions (TA_BUSY 80 %).
// The pair is fetched at xs = clamp(x0, 0, W-2); at the left/right border the in-bounds corner sits in the
// other half of the pair, which is folded into the pair WEIGHTS (one of the two is zero there).
struct FastTap { unsigned o0, o1; float wa0, wb0, wa1, wb1; };   // row y0: (a,b) pair weights, row y1 likewise

__device__ __forceinline__ FastTap make_fast_tap(float ix, float iy, int H, int W) {
  FastTap t;
  ix = __builtin_amdgcn_fmed3f(ix, -2.0f, static_cast<float>(W) + 1.0f);   // clamp (NaN -> out of bounds)
  iy = __builtin_amdgcn_fmed3f(iy, -2.0f, static_cast<float>(H) + 1.0f);
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
  const int x0 = static_cast<int>(xw), y0 = static_cast<int>(yn);
  // ATen masks the products s*e, s*w, n*e, n*w; masking the non-negative factors first gives the same bits
  const float wx0 = static_cast<unsigned>(x0) < static_cast<unsigned>(W) ? e : 0.0f;
  const float wx1 = static_cast<unsigned>(x0 + 1) < static_cast<unsigned>(W) ? w : 0.0f;
  const float wy0 = static_cast<unsigned>(y0) < static_cast<unsigned>(H) ? s : 0.0f;
  const float wy1 = static_cast<unsigned>(y0 + 1) < static_cast<unsigned>(H) ? n : 0.0f;
  // the pair is fetched at xs = clamp(x0, 0, W-2): at x0 = -1 or W-1 the single in-bounds corner sits in the
  // other half of the pair (and the out-of-bounds weight is already 0) -> swap the two x weights
  const bool swap = static_cast<unsigned>(x0) > static_cast<unsigned>(W - 2);
  const float wa = swap ? wx1 : wx0, wb = swap ? wx0 : wx1;
  t.wa0 = wy0 * wa; t.wb0 = wy0 * wb; t.wa1 = wy1 * wa; t.wb1 = wy1 * wb;
  const int xs = min(max(x0, 0), W - 2);
  const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
  // 24-bit multiplies (full rate): rows and widths are < 2^24
  t.o0 = (__umul24(ya, W) + xs) * 4u;    // BYTE offsets (32-bit): scalar base + VGPR offset addressing
  t.o1 = (__umul24(yb, W) + xs) * 4u;
  return t;
}
// load a float at a 32-bit byte offset from a (block-uniform) base
__device__ __forceinline__ float ldb(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stb(float* __restrict__ base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ float fast_cover(const FastTap& t) { return ((t.wa0 + t.wb0) + t.wa1) + t.wb1; }
__device__ __forceinline__ float fast_sample(const float* __restrict__ plane, const FastTap& t) {
  const PairF r0 = *reinterpret_cast<const PairF*>(reinterpret_cast<const char*>(plane) + t.o0);
  const PairF r1 = *reinterpret_cast<const PairF*>(reinterpret_cast<const char*>(plane) + t.o1);
  float r = r0.a * t.wa0;
  r = __fmaf_rn(r0.b, t.wb0, r);
  r = __fmaf_rn(r1.a, t.wa1, r);
  r = __fmaf_rn(r1.b, t.wb1, r);
  return r;
}

// ---------------------------------------------------------------- flow warp coordinates
// net_utils.py:42-43: g = 2*(x+u)/max(W-1,1) - 1, then grid_sample's unnormalisation.
__device__ __forceinline__ void flow_coords(int x, int y, float u, float v, int H, int W, int ac,
                                            float& ix, float& iy) {
  float gx = 2.0f * (static_cast<float>(x) + u) / static_cast<float>(W > 1 ? W - 1 : 1) - 1.0f;
  float gy = 2.0f * (static_cast<float>(y) + v) / static_cast<float>(H > 1 ? H - 1 : 1) - 1.0f;
  ix = unnormalize(gx, W, ac);
  iy = unnormalize(gy, H, ac);
}
__device__ __forceinline__ void flow_coords_d(int x, int y, float u, float v, int H, int W, int ac,
                                              const Divisor dw, const Divisor dh, float& ix, float& iy) {
  const float gx = div_exact(2.0f * (static_cast<float>(x) + u), dw) - 1.0f;
  const float gy = div_exact(2.0f * (static_cast<float>(y) + v), dh) - 1.0f;
  ix = unnormalize(gx, W, ac);
  iy = unnormalize(gy, H, ac);
}
// d(ix)/d(u): chain of the two affine maps above
__device__ __forceinline__ float flow_coord_scale(int size, int ac) {
  float den = static_cast<float>(size > 1 ? size - 1 : 1);
  return ac ? (static_cast<float>(size - 1) / den) : (static_cast<float>(size) / den);
}

// ---------------------------------------------------------------- rigid projection
// Per (sample, direction, scale) camera block prepared by k_prepare_cameras:
//   kinv = inverse(K_s), A = K_s R, b = K_s t, plus what the backward needs.
struct Camera {
  float kinv[9];
  float A[9];
  float b[3];
  float K[9];      // scaled intrinsics K_s
  float R[9];
  float dR[27];    // dR/d(rx), dR/d(ry), dR/d(rz)
};

struct Proj {
  float q0, q1, q2;  // A * Kinv * (x,y,1)
  float r0, r1, r2;  // Kinv * (x,y,1)
  float Z, U, V;     // clamped depth and pixel coordinates
  bool clamped;
};

__device__ __forceinline__ Proj project(const Camera& c, int x, int y, float depth) {
  Proj p;
  float fx = static_cast<float>(x), fy = static_cast<float>(y);
  // The reference evaluates K^-1 @ pix and (K R) @ cam as large bmm calls: MKL sgemm accumulates the three
  // products of an output element as fma(a2, b2, fma(a1, b1, a0 * b0)) (probed bit for bit, torch 2.10 CPU);
  // the translation is a separate add (inverse_warp.py:238).  With the camera block of k_prepare_cameras this
  // reproduces the reference's X, Y, Z exactly, so border pixels of an identity pose land on |grid| = 1 as they
  // do there (golden G2 "identity") and every mask fed by the projection is decided on identical bits.
  p.r0 = __fmaf_rn(c.kinv[1], fy, c.kinv[0] * fx) + c.kinv[2];
  p.r1 = __fmaf_rn(c.kinv[4], fy, c.kinv[3] * fx) + c.kinv[5];
  p.r2 = __fmaf_rn(c.kinv[7], fy, c.kinv[6] * fx) + c.kinv[8];
  float c0 = p.r0 * depth, c1 = p.r1 * depth, c2 = p.r2 * depth;
  float X = __fmaf_rn(c.A[2], c2, __fmaf_rn(c.A[1], c1, c.A[0] * c0)) + c.b[0];
  float Y = __fmaf_rn(c.A[5], c2, __fmaf_rn(c.A[4], c1, c.A[3] * c0)) + c.b[1];
  float Zr = __fmaf_rn(c.A[8], c2, __fmaf_rn(c.A[7], c1, c.A[6] * c0)) + c.b[2];
  p.q0 = c.A[0] * p.r0 + c.A[1] * p.r1 + c.A[2] * p.r2;
  p.q1 = c.A[3] * p.r0 + c.A[4] * p.r1 + c.A[5] * p.r2;
  p.q2 = c.A[6] * p.r0 + c.A[7] * p.r1 + c.A[8] * p.r2;
  p.clamped = !(Zr >= 1e-3f);
  p.Z = (Zr >= 1e-3f) ? Zr : 1e-3f;   // clamp(min=1e-3); NaN propagates like torch.clamp
  if (Zr != Zr) p.Z = Zr;
  p.U = X / p.Z;
  p.V = Y / p.Z;
  return p;
}

// Normalised sampling grid of inverse_warp2 with the zeros-padding overwrite
// (inverse_warp.py:250-257).  live_x/live_y say whether gradient flows through Xn/Yn.
__device__ __forceinline__ void rigid_grid(const Proj& p, int H, int W, float& xn, float& yn,
                                           bool& live_x, bool& live_y) {
  xn = 2.0f * p.U / static_cast<float>(W - 1) - 1.0f;
  yn = 2.0f * p.V / static_cast<float>(H - 1) - 1.0f;
  live_x = !((xn > 1.0f) || (xn < -1.0f));
  live_y = !((yn > 1.0f) || (yn < -1.0f));
  if (!live_x) xn = 2.0f;
  if (!live_y) yn = 2.0f;
}

__device__ __forceinline__ void rigid_grid_d(const Proj& p, const Divisor dw, const Divisor dh, float& xn, float& yn,
                                             bool& live_x, bool& live_y) {
  xn = div_exact(2.0f * p.U, dw) - 1.0f;
  yn = div_exact(2.0f * p.V, dh) - 1.0f;
  live_x = !((xn > 1.0f) || (xn < -1.0f));
  live_y = !((yn > 1.0f) || (yn < -1.0f));
  if (!live_x) xn = 2.0f;
  if (!live_y) yn = 2.0f;
}

// Back-propagate (gU, gV) = dL/dU, dL/dV (and optionally gZ = dL/dZ) of one pixel to
// depth and to the 12 camera sums (dL/db[3], dL/dA[9]) -- SURVEY.md A.3.
// 1 / x for GRADIENT arithmetic (compared with a tolerance, never feeding a decision): v_rcp_f32 + one Newton step, ~1 ulp,
// 3 instructions where the IEEE division takes ~10 (two of them quarter rate).  x must be a normal, non-zero float.
__device__ __forceinline__ float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __fmaf_rn(__fmaf_rn(-x, r, 1.0f), r, r);
}

__device__ __forceinline__ void project_backward(const Proj& p, float depth, float gU, float gV, float gZ,
                                                 float& gdepth, float acc[12]) {
  float invZ = rcp_nr(p.Z);
  float gX = gU * invZ, gY = gV * invZ;
  float gZr = p.clamped ? 0.0f : (gZ - (gU * p.U + gV * p.V) * invZ);
  gdepth = gX * p.q0 + gY * p.q1 + gZr * p.q2;
  acc[0] += gX; acc[1] += gY; acc[2] += gZr;
  float c0 = p.r0 * depth, c1 = p.r1 * depth, c2 = p.r2 * depth;
  acc[3] += gX * c0;  acc[4] += gX * c1;  acc[5] += gX * c2;
  acc[6] += gY * c0;  acc[7] += gY * c1;  acc[8] += gY * c2;
  acc[9] += gZr * c0; acc[10] += gZr * c1; acc[11] += gZr * c2;
}

// ---------------------------------------------------------------- mask decisions
__device__ __forceinline__ float mean3_abs_diff(float a0, float a1, float a2, float b0, float b1, float b2) {
  return div_exact((fabsf(a0 - b0) + fabsf(a1 - b1)) + fabsf(a2 - b2), Divisor{3.0f, 1.0f / 3.0f});
}

// 1 - softmax([dl, dr]) > 0.48, evaluated through the softmax as the reference does
// (model_geometry.py:119-130).  Returns soft weights too (Model_flow uses them).
// Round 6: the ONE transcendental that feeds a mask decision, exp(-|dl - dr|), is evaluated so that the decision is the one the
// CORRECTLY ROUNDED exponential gives -- no library's last bit can move it.  The hard weight flips where
// tt / (1 + tt) crosses 0.48, i.e. at tt = 0.48 / 0.52 = 0.923077 (the other weight is >= 0.5 whatever tt is); a <= 2-ulp
// expf (1.2e-7 here) plus the few roundings of the chain behind it cannot carry a value across that point from further than
// 1e-6 away.  So: the fast expf everywhere, and inside the band |tt - 0.923077| < 4e-6 (about 4 pixels in 10^5; a divergent
// branch almost no wave takes) RN32 of the float64 exponential -- correctly rounded unless the float64 value lies within 2^-29
// (relative) of a float32 rounding boundary.  oracle/loss_stack_oracle.py's ``cr_exp`` mode states the same function.
__device__ __forceinline__ float occ_exp(float a) {      // exp(-a), a >= 0
  float tt = expf(-a);
  if (fabsf(tt - 0.92307692f) < 4e-6f) tt = static_cast<float>(exp(-static_cast<double>(a)));
  return tt;
}

__device__ __forceinline__ void occ_weights(float dl, float dr, float& w_bwd, float& w_fwd) {
  // softmax([dl, dr]): exp(x - max) is exactly 1 for the larger entry, so only one exp is evaluated
  const float tt = occ_exp(fabsf(dl - dr));
  const float el = (dl >= dr) ? 1.0f : tt, er = (dr >= dl) ? 1.0f : tt;
  const float sum = el + er;
  w_bwd = 1.0f - el / sum;
  w_fwd = 1.0f - er / sum;
}

__device__ __forceinline__ float l2norm2(float a, float b) {  // torch.norm(p=2,dim=1) + 1e-12
  return sqrtf(a * a + b * b) + 1e-12f;
}

// dynamic mask: ||diff||^2 < alpha (||flow||^2 + ||rigid||^2) + beta (model_geometry.py:701-707)
__device__ __forceinline__ bool dyna_decision(float fu, float fv, float ru, float rv, float du, float dv,
                                              float alpha, float beta) {
  float nf = l2norm2(fu, fv), nr = l2norm2(ru, rv), nd = l2norm2(du, dv);
  float bound = alpha * (nf * nf + nr * nr) + beta;
  return (nd * nd) < bound;
}

// ---------------------------------------------------------------- SSIM from 3x3 box sums
// sums are already divided by 9 (AvgPool2d(3,1,1), count_include_pad) -- ssim.py:4-19
__device__ __forceinline__ float ssim_from_means(float mx, float my, float exx, float eyy, float exy) {
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float sx = exx - mx * mx, sy = eyy - my * my, sxy = exy - mx * my;
  float num = (2.0f * mx * my + C1) * (2.0f * sxy + C2);
  float den = (mx * mx + my * my + C1) * (sx + sy + C2);
  return num / den;
}

// partial derivatives of SSIM wrt (mx, my, exx, eyy, exy)
__device__ __forceinline__ void ssim_partials(float mx, float my, float exx, float eyy, float exy,
                                              float& d_mx, float& d_my, float& d_exx, float& d_eyy, float& d_exy) {
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float sxy = exy - mx * my;
  float n1 = 2.0f * mx * my + C1, n2 = 2.0f * sxy + C2;
  float d1 = mx * mx + my * my + C1, d2 = (exx - mx * mx) + (eyy - my * my) + C2;
  float inv = 1.0f / (d1 * d2);
  float s = n1 * n2 * inv;
  // num = n1*n2 ; den = d1*d2
  // dn1/dmx = 2my, dn2/dmx = -2my, dd1/dmx = 2mx, dd2/dmx = -2mx
  d_mx = (2.0f * my * n2 - 2.0f * my * n1) * inv - s * (2.0f * mx * d2 - 2.0f * mx * d1) * inv;
  d_my = (2.0f * mx * n2 - 2.0f * mx * n1) * inv - s * (2.0f * my * d2 - 2.0f * my * d1) * inv;
  d_exy = 2.0f * n1 * inv;
  d_exx = -s * d1 * inv;
  d_eyy = d_exx;
}

// The backward pass's form: the SSIM value AND its partials wrt (my, eyy, exy) from one reciprocal (v_rcp_f32 + one Newton step,
// ~1 ulp) instead of two IEEE divisions (2 x ~10 instructions, the rcp at quarter rate).  Gradients are compared with a tolerance;
// the forward's value stays the exactly divided one (ssim_from_means).
__device__ __forceinline__ float ssim_value_partials_y(float mx, float my, float exx, float eyy, float exy,
                                                       float& d_my, float& d_eyy, float& d_exy) {
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float mxx = mx * mx, myy = my * my, mxy = mx * my;
  const float n1 = 2.0f * mxy + C1, n2 = 2.0f * (exy - mxy) + C2;
  const float d1 = mxx + myy + C1, d2 = (exx - mxx) + (eyy - myy) + C2;
  const float den = d1 * d2;
  float inv = __builtin_amdgcn_rcpf(den);
  inv = __fmaf_rn(__fmaf_rn(-den, inv, 1.0f), inv, inv);
  const float s = n1 * n2 * inv;
  d_my = (2.0f * mx * (n2 - n1) - s * (2.0f * my * (d2 - d1))) * inv;
  d_exy = 2.0f * n1 * inv;
  d_eyy = -s * d1 * inv;
  return s;
}

// ---------------------------------------------------------------- resize helpers
// ATen upsample_bilinear2d source index, align_corners=False: max(0, fma(scale, dst+0.5, -0.5))
// (the CPU build contracts the expression into one FMA -- probed bit-for-bit)
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = __fmaf_rn(scale, static_cast<float>(dst) + 0.5f, -0.5f);
  if (src < 0.0f) src = 0.0f;
  i0 = static_cast<int>(src);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - static_cast<float>(i0);
  l0 = 1.0f - l1;
}

// ATen's upsample_bilinear2d inner loop (UpSampleKernel.cpp, `output = t0 * w0; output += t1 * w1` per dimension) is
// compiled to fma(t0, w0, t1 * w1) -- probed bit for bit (torch 2.10 CPU, up- and down-sampling, exact and
// non-exact ratios): horizontally t = fma(v_x0, lx0, v_x1 * lx1), then out = fma(t_y0, ly0, t_y1 * ly1).
__device__ __forceinline__ float lerp_aten(float v0, float v1, float l0, float l1) { return __fmaf_rn(v0, l0, v1 * l1); }
__device__ __forceinline__ float lerp2_aten(float v00, float v01, float v10, float v11, float lx0, float lx1,
                                            float ly0, float ly1) {
  return lerp_aten(lerp_aten(v00, v01, lx0, lx1), lerp_aten(v10, v11, lx0, lx1), ly0, ly1);
}

// When the OUTPUT of the resize has H + W <= 128, ATen dispatches to another kernel (UpSampleKernel.cpp,
// _use_vectorized_kernel_cond_2d -> cpu_upsample_linear) whose four products are associated differently:
// w_ij = ly_i * lx_j (rounded), out = fma(w11, v11, fma(w10, v10, fma(w00, v00, w01 * v01))) -- also probed bit for
// bit, with the switch at exactly 128.  (Levels 4-5 of the 375x1242 six-scale pyramid and tiny test images.)
__device__ __forceinline__ bool aten_small_resize(int outH, int outW) { return outH + outW <= 128; }
__device__ __forceinline__ float lerp2_aten_small(float v00, float v01, float v10, float v11, float lx0, float lx1,
                                                  float ly0, float ly1) {
  const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
  return __fmaf_rn(w11, v11, __fmaf_rn(w10, v10, __fmaf_rn(w00, v00, w01 * v01)));
}
__device__ __forceinline__ float lerp2_aten_sel(bool small, float v00, float v01, float v10, float v11, float lx0,
                                                float lx1, float ly0, float ly1) {
  return small ? lerp2_aten_small(v00, v01, v10, v11, lx0, lx1, ly0, ly1)
               : lerp2_aten(v00, v01, v10, v11, lx0, lx1, ly0, ly1);
}

__device__ __forceinline__ float resize_bilinear_at(const float* __restrict__ plane, int inH, int inW,
                                                    int y, int x, float sh, float sw, bool small) {
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  bilinear_src(y, sh, inH, y0, y1, ly0, ly1);
  bilinear_src(x, sw, inW, x0, x1, lx0, lx1);
  const float* r0 = plane + static_cast<long>(y0) * inW;
  const float* r1 = plane + static_cast<long>(y1) * inW;
  return lerp2_aten_sel(small, r0[x0], r0[x1], r1[x0], r1[x1], lx0, lx1, ly0, ly1);
}

// ---------------------------------------------------------------- adjoint of the bilinear resize (gather form)
// weights of the (at most G2_MAX) output indices lo .. lo+cnt-1 whose taps touch input index i (scale r = in / out)
constexpr int G2_MAX = 12;   // candidates per axis: 2/ratio + 4 (ratio >= 1/4 in registers, coarser scales loop)

__device__ __forceinline__ void adj_weights(int i, float r, int lowN, int fullN, int& lo, int& cnt, float (&w)[G2_MAX]) {
  lo = max(static_cast<int>(floorf((i - 0.5f) / r - 0.5f)) - 1, 0);
  const int hi = min(static_cast<int>(ceilf((i + 1.5f) / r - 0.5f)) + 1, fullN - 1);
  cnt = hi - lo + 1;
#pragma unroll
  for (int k = 0; k < G2_MAX; ++k) {
    int a0, a1; float l0, l1;
    bilinear_src(min(lo + k, fullN - 1), r, lowN, a0, a1, l0, l1);
    w[k] = (k < cnt) ? ((a0 == i ? l0 : 0.0f) + (a1 == i ? l1 : 0.0f)) : 0.0f;
  }
}

// ---------------------------------------------------------------- reductions
// Block-wide sums of N per-thread values with DPP adds.  The N reductions are interleaved step-major (all
// values take butterfly step k before any takes step k+1) so that the DPP read-after-write hazards are
// covered by independent instructions instead of s_nop.  Four steps leave each row of 16 lanes holding
// its row sum; one lane per row stores it to LDS and a final thread per value adds the 4*waves row sums
// in a fixed order.  Result: out[0..N) written by threads 0..N-1.  smem: N * 4 * (blockDim.x/64) floats.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(moved);
}

// lane l receives lane l-1 / l+1 of the same wave (0 at the wave edge): gfx9 DPP wave_shr:1 / wave_shl:1
__device__ __forceinline__ float wave_nbr_sum(float v) {
  const int a = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true);
  const int b = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true);
  return (v + __int_as_float(a)) + __int_as_float(b);
}

// lane l receives lane l+1 of the same wave (0 at the wave edge)
__device__ __forceinline__ float wave_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}
// lane l receives lane l-1
__device__ __forceinline__ float wave_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}

template <int N>
__device__ __forceinline__ void block_sum(float (&vals)[N], float* smem, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = dpp_add<0xB1>(vals[i]);    // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = dpp_add<0x4E>(vals[i]);    // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = dpp_add<0x141>(vals[i]);   // row_half_mirror
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = dpp_add<0x140>(vals[i]);   // row_mirror -> every lane holds its row sum
  if ((lane & 15) == 0) {
    const int slot = wave * 4 + (lane >> 4);
#pragma unroll
    for (int i = 0; i < N; ++i) smem[slot * N + i] = vals[i];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.0f;
    for (int w = 0; w < nwaves * 4; ++w) s += smem[w * N + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

// (n, c) plane of an NCHW tensor for launches with grid = (chunks, C, N): n * C + c (both limits 65535)
__device__ __forceinline__ unsigned plane_id() { return blockIdx.z * gridDim.y + blockIdx.y; }

// XCD-aware block index: consecutive logical blocks land on the same XCD (blocks are dealt
// round-robin over the 8 XCDs; MI355X_MICROARCH.md "Workgroup dispatch").  Speed only.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nblocks) {
  const unsigned nx = 8;
  unsigned per = nblocks / nx;
  if (per == 0 || bid >= per * nx) return bid;   // tail blocks keep their index
  return (bid % nx) * per + (bid / nx);
}

}  // namespace dfe
