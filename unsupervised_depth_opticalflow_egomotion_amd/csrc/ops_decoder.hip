// Glue of the depth decoder between its MIOpen convolutions (depth_model.py:60-211 of the reference):
//   ConvBlock = Conv3x3(ReflectionPad2d(1) + conv) + ELU;  decoder stage = ConvBlock, bilinear x2, cat(skip), ConvBlock.
// In ATen that glue is elu -> upsample_bilinear2d -> cat -> reflection_pad2d (and the four backward kernels plus
// the slice adds), each a full pass over the widest activations of the model.  Two fused passes replace it:
//   dfe_elu_pad_*          p = reflect_pad1(elu(x + bias))                               (elu, bias optional)
//   dfe_elu_up2_cat_pad_*  p = reflect_pad1(cat(bilinear_x2(elu(x + bias)), skip))       (skip, bias optional)
// x is the convolution output *without* its bias (the convolution is called bias-free): the broadcast bias add is
// folded into these reads and the bias gradient is a by-product of the backward pass (per-block sums finished in a
// fixed order).
// Both read the convolution output once and write the next convolution's padded input once; the backward passes
// are gathers (no atomics: bitwise reproducible).  Bound: HBM (1 read + 1 write per element, 4 B each).
// Arithmetic: elu(x) = x > 0 ? x : exp(x) - 1; elu'(x) = x > 0 ? 1 : exp(x) (ATen: expm1 / exp);
// bilinear x2 with align_corners=False: src = max(0.5*(dst+0.5)-0.5, 0), (v0*l0 + v1*l1) horizontally first.
#include "dfe_internal.h"
#include "dfe_device.h"
#include <hip/hip_runtime.h>

namespace dfe {

// exp through the hardware v_exp_f32 (absolute error ~1e-7 on these O(1) activations; expm1's extra digits near 0
// are far below the convolutions' own rounding)
__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
__device__ __forceinline__ float elu1_grad(float v) { return v > 0.0f ? 1.0f : __expf(v); }
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// taps of F.interpolate(scale_factor=2, bilinear, align_corners=False) at destination index d (source length n)
__device__ __forceinline__ void up2_tap(int d, int n, int& i0, int& i1, float& l0, float& l1) {
  const float s = fmaxf(0.5f * (static_cast<float>(d) + 0.5f) - 0.5f, 0.0f);
  i0 = static_cast<int>(s);
  i1 = min(i0 + 1, n - 1);
  l1 = s - static_cast<float>(i0);
  l0 = 1.0f - l1;
}

// adjoint of reflect_pad1 at unpadded (y, x) of an H x W plane: sum of the padded-plane entries that mirror onto it
__device__ __forceinline__ float pad_adjoint(const float* __restrict__ gp, int y, int x, int H, int W) {
  const int Wp = W + 2;
  int ry[3], rx[3], ny = 1, nx = 1;
  ry[0] = y + 1; rx[0] = x + 1;
  if (y == 1) ry[ny++] = 0;
  if (y == H - 2) ry[ny++] = H + 1;
  if (x == 1) rx[nx++] = 0;
  if (x == W - 2) rx[nx++] = W + 1;
  float s = 0.0f;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b) s += gp[static_cast<long>(ry[a]) * Wp + rx[b]];
  return s;
}

// ---------------------------------------------------------------- p = pad(elu(x))
// grid: x over the elements of one padded plane, y = c, z = b
__global__ void __launch_bounds__(256) k_elu_pad_fwd(const float* __restrict__ x, const float* __restrict__ bias,
                                                     float* __restrict__ out, int C, int H, int W, int elu) {
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>((H + 2) * (W + 2))) return;
  const int oy = e / static_cast<unsigned>(W + 2), ox = e - oy * (W + 2);
  const long pl = plane_id();
  const float v = x[(pl * H + reflect1(oy - 1, H)) * W + reflect1(ox - 1, W)] + (bias ? bias[plane_id() % C] : 0.0f);
  out[(pl * (H + 2) + oy) * (W + 2) + ox] = elu ? elu1(v) : v;
}

__global__ void __launch_bounds__(256) k_elu_pad_bwd(const float* __restrict__ x, const float* __restrict__ bias,
                                                     const float* __restrict__ gp, float* __restrict__ gx,
                                                     float* __restrict__ part, int C, int H, int W, int elu) {
  __shared__ float red[4 * 4];
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[1] = {0.0f};
  if (e < static_cast<unsigned>(H * W)) {
    const int iy = e / static_cast<unsigned>(W), ix = e - iy * W;
    const long pl = plane_id();
    float g = pad_adjoint(gp + pl * (H + 2) * (W + 2), iy, ix, H, W);
    const long o = (pl * H + iy) * W + ix;
    if (elu) g *= elu1_grad(x[o] + (bias ? bias[plane_id() % C] : 0.0f));
    gx[o] = g;
    acc[0] = g;
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

// ---- two elements per thread (even W): these passes are pure streaming, so halving the instruction count per byte
// is what moves them towards the HBM roofline.  Interior pairs take one 8-byte load / store per stream (the padded
// plane's pair sits at an odd offset: a dword-aligned 8-byte load); the pairs touching a mirrored column or row
// take the scalar path.  grid: x over H * W/2 (bwd) or (H+2) * (W+2)/2 (fwd) pairs, y = c, z = b.
struct __attribute__((aligned(8))) AF2 { float a, b; };

__global__ void __launch_bounds__(256) k_elu_pad_fwd_pair(const float* __restrict__ x, const float* __restrict__ bias,
                                                          float* __restrict__ out, int C, int H, int W, int elu) {
  const int Wh = (W + 2) / 2;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>((H + 2) * Wh)) return;
  const int oy = e / static_cast<unsigned>(Wh), ox = 2 * (e - oy * Wh);
  const long pl = plane_id();
  const float bv = bias ? bias[plane_id() % C] : 0.0f;
  const float* row = x + (pl * H + reflect1(oy - 1, H)) * W;
  float v0, v1;
  if (ox >= 2 && ox <= W - 2) { const PairF q = *reinterpret_cast<const PairF*>(row + ox - 1); v0 = q.a; v1 = q.b; }
  else { v0 = row[reflect1(ox - 1, W)]; v1 = row[reflect1(ox, W)]; }
  v0 += bv; v1 += bv;
  AF2 o;
  o.a = elu ? elu1(v0) : v0; o.b = elu ? elu1(v1) : v1;
  *reinterpret_cast<AF2*>(out + (pl * (H + 2) + oy) * (W + 2) + ox) = o;
}

__global__ void __launch_bounds__(256) k_elu_pad_bwd_pair(const float* __restrict__ x, const float* __restrict__ bias,
                                                          const float* __restrict__ gp, float* __restrict__ gx,
                                                          float* __restrict__ part, int C, int H, int W, int elu) {
  __shared__ float red[4 * 4];
  const int Wh = W / 2;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[1] = {0.0f};
  if (e < static_cast<unsigned>(H * Wh)) {
    const int iy = e / static_cast<unsigned>(Wh), ix = 2 * (e - iy * Wh);
    const long pl = plane_id();
    const float* g = gp + pl * (H + 2) * (W + 2);
    float g0, g1;
    if (iy != 1 && iy != H - 2 && ix != 0 && ix != W - 2) {
      const PairF q = *reinterpret_cast<const PairF*>(g + static_cast<long>(iy + 1) * (W + 2) + ix + 1);
      g0 = q.a; g1 = q.b;
    } else { g0 = pad_adjoint(g, iy, ix, H, W); g1 = pad_adjoint(g, iy, ix + 1, H, W); }
    const long o = (pl * H + iy) * W + ix;
    if (elu) {
      const AF2 xv = *reinterpret_cast<const AF2*>(x + o);
      const float bv = bias ? bias[plane_id() % C] : 0.0f;
      g0 *= elu1_grad(xv.a + bv); g1 *= elu1_grad(xv.b + bv);
    }
    AF2 r; r.a = g0; r.b = g1;
    *reinterpret_cast<AF2*>(gx + o) = r;
    acc[0] = g0 + g1;
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

__global__ void __launch_bounds__(256) k_cat_pad_bwd_skip_pair(const float* __restrict__ gp, float* __restrict__ gskip,
                                                               int C1, int C2, int H, int W) {
  const int Wh = W / 2;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>(H * Wh)) return;
  const int iy = e / static_cast<unsigned>(Wh), ix = 2 * (e - iy * Wh);
  const int b = plane_id() / C2, c = plane_id() - b * C2;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + C1 + c) * (H + 2) * (W + 2);
  AF2 r;
  if (iy != 1 && iy != H - 2 && ix != 0 && ix != W - 2) {
    const PairF q = *reinterpret_cast<const PairF*>(g + static_cast<long>(iy + 1) * (W + 2) + ix + 1);
    r.a = q.a; r.b = q.b;
  } else { r.a = pad_adjoint(g, iy, ix, H, W); r.b = pad_adjoint(g, iy, ix + 1, H, W); }
  *reinterpret_cast<AF2*>(gskip + (static_cast<long>(plane_id()) * H + iy) * W + ix) = r;
}

// ---------------------------------------------------------------- p = pad(cat(up2(elu(x)), skip))
// x [B,C1,h,w], skip [B,C2,2h,2w] (C2 may be 0), out [B,C1+C2,2h+2,2w+2]
// grid: x over the elements of one padded plane, y = c, z = b
__global__ void __launch_bounds__(256) k_elu_up2_cat_pad_fwd(const float* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ out,
                                                             int C1, int C2, int h, int w) {
  const int H = 2 * h, W = 2 * w;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>((H + 2) * (W + 2))) return;
  const int oy = e / static_cast<unsigned>(W + 2), ox = e - oy * (W + 2);
  const int C = C1 + C2;
  const int b = plane_id() / C, c = plane_id() - b * C;
  const int y = reflect1(oy - 1, H), xx = reflect1(ox - 1, W);
  float v;
  if (c < C1) {
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    up2_tap(y, h, y0, y1, ly0, ly1);
    up2_tap(xx, w, x0, x1, lx0, lx1);
    const float* p = x + (static_cast<long>(b) * C1 + c) * h * w;
    const float* r0 = p + static_cast<long>(y0) * w;
    const float* r1 = p + static_cast<long>(y1) * w;
    const float bv = bias ? bias[c] : 0.0f;
    float v00, v01, v10, v11;
    if (x1 == x0 + 1) {       // the two taps of a row in one dword-aligned 8-byte load
      const PairF a = *reinterpret_cast<const PairF*>(r0 + x0), d = *reinterpret_cast<const PairF*>(r1 + x0);
      v00 = a.a; v01 = a.b; v10 = d.a; v11 = d.b;
    } else { v00 = v01 = r0[x0]; v10 = v11 = r1[x0]; }
    v = lerp2_aten(elu1(v00 + bv), elu1(v01 + bv), elu1(v10 + bv), elu1(v11 + bv), lx0, lx1, ly0, ly1);
  } else {
    v = skip[((static_cast<long>(b) * C2 + (c - C1)) * H + y) * W + xx];
  }
  out[(static_cast<long>(plane_id()) * (H + 2) + oy) * (W + 2) + ox] = v;
}

// two padded output columns (ox, ox+1), ox even, per thread.  In the interior both come from the same two source
// columns j = ox/2 - 1 and j + 1 (weights 3/4, 1/4 and 1/4, 3/4): 4 ELUs and 2 pair loads for 2 outputs.
__global__ void __launch_bounds__(256) k_elu_up2_cat_pad_fwd_pair(const float* __restrict__ x, const float* __restrict__ bias,
                                                                  const float* __restrict__ skip, float* __restrict__ out,
                                                                  int C1, int C2, int h, int w) {
  const int H = 2 * h, W = 2 * w, Wh = w + 1;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>((H + 2) * Wh)) return;
  const int oy = e / static_cast<unsigned>(Wh), ox = 2 * (e - oy * Wh);
  const int C = C1 + C2;
  const int b = plane_id() / C, c = plane_id() - b * C;
  const int y = reflect1(oy - 1, H);
  const bool inner = ox >= 2 && ox <= W - 2;
  AF2 o;
  if (c < C1) {
    int y0, y1; float ly0, ly1;
    up2_tap(y, h, y0, y1, ly0, ly1);
    const float* p = x + (static_cast<long>(b) * C1 + c) * h * w;
    const float* r0 = p + static_cast<long>(y0) * w;
    const float* r1 = p + static_cast<long>(y1) * w;
    const float bv = bias ? bias[c] : 0.0f;
    if (inner) {
      const int j = ox / 2 - 1;
      const PairF a = *reinterpret_cast<const PairF*>(r0 + j), d = *reinterpret_cast<const PairF*>(r1 + j);
      const float e00 = elu1(a.a + bv), e01 = elu1(a.b + bv), e10 = elu1(d.a + bv), e11 = elu1(d.b + bv);
      o.a = lerp2_aten(e00, e01, e10, e11, 0.75f, 0.25f, ly0, ly1);
      o.b = lerp2_aten(e00, e01, e10, e11, 0.25f, 0.75f, ly0, ly1);
    } else {
      float v[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int x0, x1; float lx0, lx1;
        up2_tap(reflect1(ox - 1 + k, W), w, x0, x1, lx0, lx1);
        v[k] = lerp2_aten(elu1(r0[x0] + bv), elu1(r0[x1] + bv), elu1(r1[x0] + bv), elu1(r1[x1] + bv), lx0, lx1, ly0, ly1);
      }
      o.a = v[0]; o.b = v[1];
    }
  } else {
    const float* row = skip + ((static_cast<long>(b) * C2 + (c - C1)) * H + y) * W;
    if (inner) { const PairF q = *reinterpret_cast<const PairF*>(row + ox - 1); o.a = q.a; o.b = q.b; }
    else { o.a = row[reflect1(ox - 1, W)]; o.b = row[reflect1(ox, W)]; }
  }
  *reinterpret_cast<AF2*>(out + (static_cast<long>(plane_id()) * (H + 2) + oy) * (W + 2) + ox) = o;
}

// gradient wrt x: thread per low-res element; the <= 4x4 full-res outputs whose taps touch it, each through the
// adjoint of the reflection pad.  grid: x over the low-res plane, y = c, z = b
__global__ void __launch_bounds__(256) k_elu_up2_cat_pad_bwd_x(const float* __restrict__ x, const float* __restrict__ bias,
                                                               const float* __restrict__ gp, float* __restrict__ gx,
                                                               float* __restrict__ part, int C1, int C2, int h, int w) {
  __shared__ float red[4 * 4];
  const int H = 2 * h, W = 2 * w;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[1] = {0.0f};
  if (e < static_cast<unsigned>(h * w)) {
  const int i = e / static_cast<unsigned>(w), j = e - i * w;
  const int b = plane_id() / C1, c = plane_id() - b * C1;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + c) * (H + 2) * (W + 2);
  float total = 0.0f;
  if (i >= 2 && i < h - 2 && j >= 2 && j < w - 2) {
    // interior: the 4x4 outputs (rows 2i-1..2i+2, columns 2j-1..2j+2) map one-to-one onto the padded plane and the
    // tent weights are (1/4, 3/4, 3/4, 1/4) on both axes; two 8-byte loads per row
    const float* q = g + static_cast<long>(2 * i) * (W + 2) + 2 * j;
    const float wk[4] = {0.25f, 0.75f, 0.75f, 0.25f};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const PairF a = *reinterpret_cast<const PairF*>(q + static_cast<long>(ky) * (W + 2));
      const PairF d = *reinterpret_cast<const PairF*>(q + static_cast<long>(ky) * (W + 2) + 2);
      total += wk[ky] * (((0.25f * a.a + 0.75f * a.b) + 0.75f * d.a) + 0.25f * d.b);
    }
  } else {
  float wy[4], wx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int a0, a1; float l0, l1;
    const int y = 2 * i - 1 + k;
    up2_tap(min(max(y, 0), H - 1), h, a0, a1, l0, l1);
    wy[k] = (y >= 0 && y < H) ? ((a0 == i ? l0 : 0.0f) + (a1 == i ? l1 : 0.0f)) : 0.0f;
    const int xq = 2 * j - 1 + k;
    up2_tap(min(max(xq, 0), W - 1), w, a0, a1, l0, l1);
    wx[k] = (xq >= 0 && xq < W) ? ((a0 == j ? l0 : 0.0f) + (a1 == j ? l1 : 0.0f)) : 0.0f;
  }
#pragma unroll
  for (int ky = 0; ky < 4; ++ky) {
    if (wy[ky] == 0.0f) continue;
    float acc = 0.0f;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
      if (wx[kx] != 0.0f) acc += wx[kx] * pad_adjoint(g, 2 * i - 1 + ky, 2 * j - 1 + kx, H, W);
    total += wy[ky] * acc;
  }
  }
  const long o = ((static_cast<long>(b) * C1 + c) * h + i) * w + j;
  acc[0] = total * elu1_grad(x[o] + (bias ? bias[c] : 0.0f));
  gx[o] = acc[0];
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

// The same gradient as a rolling-window wave kernel (the thread-per-element kernel above reads every padded gradient
// four times through L1 with ten vector-memory instructions per output: 0.8 TB/s).  A wave owns 64 low-res columns
// and marches down UB_ROWS low-res rows: low-res row i needs the padded rows 2i..2i+3, two of which the next row
// re-uses, so each step loads two padded rows -- one 8-byte load per lane (columns 2j, 2j+1), columns 2j+2, 2j+3 come
// from the right-hand lane by DPP wave shifts -- folds them with the horizontal tent (1/4, 3/4, 3/4, 1/4) and keeps two
// folded rows in registers: every padded gradient is loaded once, 4 vector-memory instructions per output row.
// Border elements (two rows / columns on each side: clamped taps and the reflection-pad adjoint) take the general
// path of the kernel above.  Same arithmetic and summation order per element -> identical gx.
// grid: x = strips * row blocks, y = c, z = b; block = one wave.
constexpr int UB_ROWS = 8;

__global__ void __launch_bounds__(64) k_elu_up2_cat_pad_bwd_x_roll(const float* __restrict__ x, const float* __restrict__ bias,
                                                                   const float* __restrict__ gp, float* __restrict__ gx,
                                                                   float* __restrict__ part, int C1, int C2, int h, int w,
                                                                   int strips) {
  __shared__ float red[4];
  const int H = 2 * h, W = 2 * w, Wp = W + 2;
  const int strip = blockIdx.x % strips, rb = blockIdx.x / strips;
  const int lane = threadIdx.x;
  const int j = strip * 64 + lane, jc = min(j, w - 1);
  const int i0 = rb * UB_ROWS, i1 = min(i0 + UB_ROWS, h);
  const int b = plane_id() / C1, c = plane_id() - b * C1;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + c) * (H + 2) * Wp;
  const float bv = bias ? bias[c] : 0.0f;
  const bool col_fast = j >= 2 && j < w - 2;
  const bool own_d = lane == 63 && 2 * jc + 3 < Wp;
  // raw loads of padded row r (columns 2j, 2j+1; the last lane also fetches 2j+2, 2j+3 itself) are issued one step
  // ahead of their use; the fold is the horizontal tent over columns 2j..2j+3 (every lane executes the shifts)
  struct Raw { PairF a, d; };
  auto load_row = [&](int r) {
    Raw o;
    const float* q = g + static_cast<long>(min(r, H + 1)) * Wp + 2 * jc;
    o.a = *reinterpret_cast<const PairF*>(q);
    o.d = o.a;
    if (own_d) o.d = *reinterpret_cast<const PairF*>(q + 2);
    return o;
  };
#define DFE_UB_FOLD(dst, raw)                                                   \
  {                                                                             \
    float da_ = wave_shl1(raw.a.a), db_ = wave_shl1(raw.a.b);                   \
    if (own_d) { da_ = raw.d.a; db_ = raw.d.b; }                                \
    dst = ((0.25f * raw.a.a + 0.75f * raw.a.b) + 0.75f * da_) + 0.25f * db_;    \
  }
  const long obase = (static_cast<long>(b) * C1 + c) * h * w + jc;
  float hA, hB;
  {
    const Raw r0 = load_row(2 * i0), r1 = load_row(2 * i0 + 1);
    DFE_UB_FOLD(hA, r0);
    DFE_UB_FOLD(hB, r1);
  }
  Raw nC = load_row(2 * i0 + 2), nD = load_row(2 * i0 + 3);
  float nx = x[obase + static_cast<long>(i0) * w];
  float acc[1] = {0.0f};
  for (int i = i0; i < i1; ++i) {
    const Raw cC = nC, cD = nD;
    const float xv = nx;
    nC = load_row(2 * i + 4); nD = load_row(2 * i + 5);          // rows of step i+1 (clamped at the plane's end)
    nx = x[obase + static_cast<long>(min(i + 1, h - 1)) * w];
    float hC, hD;
    DFE_UB_FOLD(hC, cC);
    DFE_UB_FOLD(hD, cD);
    if (j < w) {
      float total = 0.0f;
      if (col_fast && i >= 2 && i < h - 2) {
        total += 0.25f * hA; total += 0.75f * hB; total += 0.75f * hC; total += 0.25f * hD;
      } else {
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int a0, a1; float l0, l1;
          const int y = 2 * i - 1 + k;
          up2_tap(min(max(y, 0), H - 1), h, a0, a1, l0, l1);
          wy[k] = (y >= 0 && y < H) ? ((a0 == i ? l0 : 0.0f) + (a1 == i ? l1 : 0.0f)) : 0.0f;
          const int xq = 2 * j - 1 + k;
          up2_tap(min(max(xq, 0), W - 1), w, a0, a1, l0, l1);
          wx[k] = (xq >= 0 && xq < W) ? ((a0 == j ? l0 : 0.0f) + (a1 == j ? l1 : 0.0f)) : 0.0f;
        }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
          if (wy[ky] == 0.0f) continue;
          float a = 0.0f;
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
            if (wx[kx] != 0.0f) a += wx[kx] * pad_adjoint(g, 2 * i - 1 + ky, 2 * j - 1 + kx, H, W);
          total += wy[ky] * a;
        }
      }
      const float v = total * elu1_grad(xv + bv);
      gx[obase + static_cast<long>(i) * w] = v;
      acc[0] += v;
    }
    hA = hC; hB = hD;
  }
#undef DFE_UB_FOLD
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

// gbias[c] = sum over b, blocks of part[(b*C + c)*nblk + k] in a fixed order; one wave per channel
__global__ void __launch_bounds__(64) k_glue_bias_final(const float* __restrict__ part, float* __restrict__ gbias,
                                                        int B, int C, int nblk) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = part + (static_cast<long>(b) * C + c) * nblk;
    for (int k = lane; k < nblk; k += 64) s += p[k];
  }
  s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x141>(s); s = dpp_add<0x140>(s);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
  if (lane == 0) gbias[c] = (r0 + r1) + (r2 + r3);
}

// gradient wrt skip: adjoint of the pad only.  grid: x over the plane, y = c, z = b
__global__ void __launch_bounds__(256) k_cat_pad_bwd_skip(const float* __restrict__ gp, float* __restrict__ gskip,
                                                          int C1, int C2, int H, int W) {
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>(H * W)) return;
  const int iy = e / static_cast<unsigned>(W), ix = e - iy * W;
  const int b = plane_id() / C2, c = plane_id() - b * C2;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + C1 + c) * (H + 2) * (W + 2);
  gskip[(static_cast<long>(plane_id()) * H + iy) * W + ix] = pad_adjoint(g, iy, ix, H, W);
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)

using namespace dfe;

static inline bool grid_ok(long plane_elems, long batch, long channels) { return plane_elems < (1L << 31) && batch <= 65535 && channels <= 65535; }
static inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
static inline unsigned nblk(long n) { return static_cast<unsigned>((n + 255) / 256); }

extern "C" long dfe_glue_partials_floats(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const long roll = W >= 256 ? static_cast<long>((W + 63) / 64) * ((H + UB_ROWS - 1) / UB_ROWS) : 0;   // units of the rolling kernel (wide planes only)
  const long blocks = nblk(static_cast<long>(H) * W);
  return static_cast<long>(B) * C * (roll > blocks ? roll : blocks);
}

extern "C" int dfe_elu_pad_fwd(const float* x, const float* bias, float* out, int B, int C, int H, int W, int apply_elu,
                               void* stream) {
  if (!x || !out) return DFE_ERR_NULL;
  if (B <= 0 || C <= 0 || H < 2 || W < 2 || !grid_ok((H + 2L) * (W + 2L), B, C)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (W % 2 == 0 && W >= 4 && al8(out)) k_elu_pad_fwd_pair<<<dim3(nblk((H + 2L) * ((W + 2L) / 2)), C, B), 256, 0, st>>>(x, bias, out, C, H, W, apply_elu);
  else k_elu_pad_fwd<<<dim3(nblk((H + 2L) * (W + 2L)), C, B), 256, 0, st>>>(x, bias, out, C, H, W, apply_elu);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_elu_pad_bwd(const float* x, const float* bias, const float* gout, float* gx, float* gbias, float* partials,
                               int B, int C, int H, int W, int apply_elu, void* stream) {
  if (!gout || !gx || (apply_elu && !x) || (gbias && !partials)) return DFE_ERR_NULL;
  if (B <= 0 || C <= 0 || H < 2 || W < 2 || !grid_ok((H + 2L) * (W + 2L), B, C)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool pair = W % 2 == 0 && W >= 4 && al8(gx) && (!apply_elu || al8(x));
  const unsigned nb = nblk(pair ? static_cast<long>(H) * (W / 2) : static_cast<long>(H) * W);   // <= the scratch size
  if (pair) k_elu_pad_bwd_pair<<<dim3(nb, C, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C, H, W, apply_elu);
  else k_elu_pad_bwd<<<dim3(nb, C, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C, H, W, apply_elu);
  DFE_LAUNCH_CHECK();
  if (gbias) {
    k_glue_bias_final<<<C, 64, 0, st>>>(partials, gbias, B, C, static_cast<int>(nb));
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

extern "C" int dfe_elu_up2_cat_pad_fwd(const float* x, const float* bias, const float* skip, float* out, int B, int C1, int C2,
                                       int h, int w, void* stream) {
  if (!x || !out || (C2 > 0 && !skip)) return DFE_ERR_NULL;
  if (B <= 0 || C1 <= 0 || C2 < 0 || h < 1 || w < 1 || !grid_ok((2L * h + 2) * (2L * w + 2), B, C1 + C2)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w >= 2 && al8(out)) k_elu_up2_cat_pad_fwd_pair<<<dim3(nblk((2L * h + 2) * (w + 1L)), C1 + C2, B), 256, 0, st>>>(x, bias, skip, out, C1, C2, h, w);
  else k_elu_up2_cat_pad_fwd<<<dim3(nblk((2L * h + 2) * (2L * w + 2)), C1 + C2, B), 256, 0, st>>>(x, bias, skip, out, C1, C2, h, w);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_elu_up2_cat_pad_bwd(const float* x, const float* bias, const float* gout, float* gx, float* gskip,
                                       float* gbias, float* partials, int B, int C1, int C2, int h, int w, void* stream) {
  if (!x || !gout || (!gx && !gskip) || (gbias && (!partials || !gx))) return DFE_ERR_NULL;
  if (B <= 0 || C1 <= 0 || C2 < 0 || h < 1 || w < 1 || !grid_ok((2L * h + 2) * (2L * w + 2), B, C1 + C2)) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gx) {
    unsigned nb = nblk(static_cast<long>(h) * w);
    if (w >= 256 && al8(gout)) {   // measured (12 images): 16 ch @128x416 178 -> 136 us; at 64x208 and below the element kernel wins (116 vs 152 us)
      const int strips = (w + 63) / 64;
      nb = static_cast<unsigned>(strips * ((h + UB_ROWS - 1) / UB_ROWS));
      k_elu_up2_cat_pad_bwd_x_roll<<<dim3(nb, C1, B), 64, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C1, C2, h, w, strips);
    } else {
      k_elu_up2_cat_pad_bwd_x<<<dim3(nb, C1, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C1, C2, h, w);
    }
    DFE_LAUNCH_CHECK();
    if (gbias) {
      k_glue_bias_final<<<C1, 64, 0, st>>>(partials, gbias, B, C1, static_cast<int>(nb));
      DFE_LAUNCH_CHECK();
    }
  }
  if (gskip && C2 > 0) {
    if (w >= 2 && al8(gskip)) k_cat_pad_bwd_skip_pair<<<dim3(nblk(2L * h * w), C2, B), 256, 0, st>>>(gout, gskip, C1, C2, 2 * h, 2 * w);
    else k_cat_pad_bwd_skip<<<dim3(nblk(4L * h * w), C2, B), 256, 0, st>>>(gout, gskip, C1, C2, 2 * h, 2 * w);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
