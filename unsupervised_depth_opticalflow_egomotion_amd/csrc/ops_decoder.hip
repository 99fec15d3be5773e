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
#include <cstdlib>

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

// ---- four elements per thread (W % 4 == 0): 16-byte loads / stores of x and gx, and ONE dword-aligned 16-byte load
// of the padded gradient (it sits one element to the right: offset +1); the quads that contain column 1 or W-2 and
// the rows 1 and H-2 collect the mirrored padding entries element by element.  Round 3: the pair kernels above move
// 8 bytes per lane and instruction and stop at 3 TB/s on the decoder's large planes.
struct __attribute__((packed, aligned(4))) QuadU { float a, b, c, d; };     // dword-aligned 16 bytes

__device__ __forceinline__ float4 pad_adjoint_quad(const float* __restrict__ g, int iy, int ix, int H, int W) {
  if (iy != 1 && iy != H - 2 && ix != 0 && ix != W - 4) {
    const QuadU q = *reinterpret_cast<const QuadU*>(g + static_cast<long>(iy + 1) * (W + 2) + ix + 1);
    return make_float4(q.a, q.b, q.c, q.d);
  }
  return make_float4(pad_adjoint(g, iy, ix, H, W), pad_adjoint(g, iy, ix + 1, H, W), pad_adjoint(g, iy, ix + 2, H, W),
                     pad_adjoint(g, iy, ix + 3, H, W));
}

__global__ void __launch_bounds__(256) k_elu_pad_bwd_quad(const float* __restrict__ x, const float* __restrict__ bias,
                                                          const float* __restrict__ gp, float* __restrict__ gx,
                                                          float* __restrict__ part, int C, int H, int W, int elu) {
  __shared__ float red[4 * 4];
  const int Wq = W / 4;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[1] = {0.0f};
  if (e < static_cast<unsigned>(H * Wq)) {
    const int iy = e / static_cast<unsigned>(Wq), ix = 4 * (e - iy * Wq);
    const long pl = plane_id();
    float4 g = pad_adjoint_quad(gp + pl * (H + 2) * (W + 2), iy, ix, H, W);
    const long o = (pl * H + iy) * W + ix;
    if (elu) {
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      const float bv = bias ? bias[plane_id() % C] : 0.0f;
      g.x *= elu1_grad(xv.x + bv); g.y *= elu1_grad(xv.y + bv); g.z *= elu1_grad(xv.z + bv); g.w *= elu1_grad(xv.w + bv);
    }
    *reinterpret_cast<float4*>(gx + o) = g;
    acc[0] = (g.x + g.y) + (g.z + g.w);
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

__global__ void __launch_bounds__(256) k_cat_pad_bwd_skip_quad(const float* __restrict__ gp, float* __restrict__ gskip,
                                                               int C1, int C2, int H, int W) {
  const int Wq = W / 4;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= static_cast<unsigned>(H * Wq)) return;
  const int iy = e / static_cast<unsigned>(Wq), ix = 4 * (e - iy * Wq);
  const int b = plane_id() / C2, c = plane_id() - b * C2;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + C1 + c) * (H + 2) * (W + 2);
  *reinterpret_cast<float4*>(gskip + (static_cast<long>(plane_id()) * H + iy) * W + ix) = pad_adjoint_quad(g, iy, ix, H, W);
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

// The same gradient through an LDS tile (round 3): a block owns UT_H x TW low-res outputs of one plane, stages the
// (2 UT_H + 2) x (2 TW + 2) padded gradients they touch with coalesced 8-byte loads that are all in flight at once
// (the element kernel above issues ten tiny loads per output: 0.6-1.1 TB/s; round 2's rolling-window wave kernel, two
// loads per row with one row of prefetch, reached 1.9 TB/s on the widest plane and is gone), and every lane then folds
// its four columns and four rows out of LDS.  Depth decoder, 12 images: 393 -> 185 us over the five stages, 3.6 TB/s
// on the widest.  Tiles
// that touch the plane's ring apply the reflection-pad adjoint while staging (zero outside the plane) and use the
// clamped tap weights; same products and the same summation order per element as the two kernels above -> identical gx.
// grid: x = tiles_x * tiles_y, y = c, z = b; 256 threads = (256 / TW) row groups x TW columns.
constexpr int UT_H = 16;

__device__ __forceinline__ void up2_adjoint_weights(int i, int n_lo, float (&wgt)[4]) {
  const int n_hi = 2 * n_lo;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int a0, a1; float l0, l1;
    const int y = 2 * i - 1 + k;
    up2_tap(min(max(y, 0), n_hi - 1), n_lo, a0, a1, l0, l1);
    wgt[k] = (y >= 0 && y < n_hi) ? ((a0 == i ? l0 : 0.0f) + (a1 == i ? l1 : 0.0f)) : 0.0f;
  }
}

template <int TW>
__global__ void __launch_bounds__(256) k_elu_up2_cat_pad_bwd_x_tile(const float* __restrict__ x, const float* __restrict__ bias,
                                                                    const float* __restrict__ gp, float* __restrict__ gx,
                                                                    float* __restrict__ part, int C1, int C2, int h, int w,
                                                                    int tiles_x) {
  constexpr int GROUPS = 256 / TW, R = UT_H / GROUPS, LW = 2 * TW + 2, LH = 2 * UT_H + 2, LW2 = LW / 2;
  __shared__ __attribute__((aligned(16))) float tile[LH * LW];
  __shared__ float red[4 * 4];
  const int H = 2 * h, W = 2 * w, Wp = W + 2;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int i0 = ty * UT_H, j0 = tx * TW;
  const int b = plane_id() / C1, c = plane_id() - b * C1;
  const float* g = gp + (static_cast<long>(b) * (C1 + C2) + c) * (H + 2) * Wp;
  const int t = threadIdx.x, jl = t % TW, grp = t / TW;
  const int j = j0 + jl;
  const long obase = (static_cast<long>(b) * C1 + c) * h * w;
  // this thread's outputs: rows i0 + grp * R + q, column j; their x values are fetched while the tile is staged
  float xv[R];
#pragma unroll
  for (int q = 0; q < R; ++q) xv[q] = x[obase + static_cast<long>(min(i0 + grp * R + q, h - 1)) * w + min(j, w - 1)];
  const bool inner = i0 >= 2 && i0 + UT_H <= h - 2 && j0 >= 2 && j0 + TW <= w - 2;   // wave-uniform
  // entry (r, cc) = unpadded gradient at (y, xq) = (2 i0 - 1 + r, 2 j0 - 1 + cc); away from the plane's ring that is
  // the padded entry (2 i0 + r, 2 j0 + cc): a plain copy, 8 bytes per load (addresses clamped into the padded plane)
  {
    const float* src = g + static_cast<long>(2 * i0) * Wp + 2 * j0;
    const int rmax = H + 1 - 2 * i0, kmax = (W - 2 * j0) / 2;
#pragma unroll
    for (int it = 0; it < (LH * LW2 + 255) / 256; ++it) {
      const int idx = it * 256 + t;
      if (idx < LH * LW2) {
        const int r = idx / LW2, k = idx - r * LW2;
        const AF2 v = *reinterpret_cast<const AF2*>(src + static_cast<long>(inner ? r : min(r, rmax)) * Wp + 2 * (inner ? k : min(k, kmax)));
        *reinterpret_cast<AF2*>(tile + r * LW + 2 * k) = v;
      }
    }
  }
  if (!inner) {
    // ring tiles: rows / columns 1 and H-2 / W-2 of the plane also collect the mirrored padding entries (reflection-pad
    // adjoint): at most two rows and two columns of the tile.  Entries outside the plane keep whatever the clamped copy
    // brought in: their tap weights are zero and zero weights are skipped below.
    __syncthreads();
    const int rs[2] = {2 - 2 * i0, H - 1 - 2 * i0}, ys[2] = {1, H - 2};
    const int cs[2] = {2 - 2 * j0, W - 1 - 2 * j0}, xs[2] = {1, W - 2};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (rs[k] >= 0 && rs[k] < LH && t < LW) {
        const int xq = 2 * j0 - 1 + t;
        if (xq >= 0 && xq < W) tile[rs[k] * LW + t] = pad_adjoint(g, ys[k], xq, H, W);
      }
      if (cs[k] >= 0 && cs[k] < LW && t < LH) {
        const int y = 2 * i0 - 1 + t;
        if (y >= 0 && y < H) tile[t * LW + cs[k]] = pad_adjoint(g, y, xs[k], H, W);
      }
    }
  }
  __syncthreads();
  float wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  if (!inner) up2_adjoint_weights(min(j, w - 1), w, wx);
  // horizontal fold of the 2R + 2 staged rows this thread's outputs touch
  float f[2 * R + 2];
  const float* trow = tile + (2 * grp * R) * LW + 2 * jl;
#pragma unroll
  for (int rr = 0; rr < 2 * R + 2; ++rr) {
    const AF2 a = *reinterpret_cast<const AF2*>(trow + rr * LW), d = *reinterpret_cast<const AF2*>(trow + rr * LW + 2);
    if (inner) f[rr] = ((0.25f * a.a + 0.75f * a.b) + 0.75f * d.a) + 0.25f * d.b;
    else {
      float s = 0.0f;
      if (wx[0] != 0.0f) s += wx[0] * a.a;
      if (wx[1] != 0.0f) s += wx[1] * a.b;
      if (wx[2] != 0.0f) s += wx[2] * d.a;
      if (wx[3] != 0.0f) s += wx[3] * d.b;
      f[rr] = s;
    }
  }
  const float bv = bias ? bias[c] : 0.0f;
  float acc[1] = {0.0f};
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int i = i0 + grp * R + q;
    float total = 0.0f;
    if (inner) {
      total += 0.25f * f[2 * q]; total += 0.75f * f[2 * q + 1]; total += 0.75f * f[2 * q + 2]; total += 0.25f * f[2 * q + 3];
    } else {
      float wy[4];
      up2_adjoint_weights(min(i, h - 1), h, wy);
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
        if (wy[ky] != 0.0f) total += wy[ky] * f[2 * q + ky];
    }
    if (i < h && j < w) {
      const float v = total * elu1_grad(xv[q] + bv);
      gx[obase + static_cast<long>(i) * w + j] = v;
      acc[0] += v;
    }
  }
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
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline unsigned nblk(long n) { return static_cast<unsigned>((n + 255) / 256); }

extern "C" long dfe_glue_partials_floats(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return static_cast<long>(B) * C * nblk(static_cast<long>(H) * W);     // one partial per block of the widest grid used
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
  const bool quad = W % 4 == 0 && W >= 8 && al16(gx) && (!apply_elu || al16(x));
  const unsigned nb = nblk(quad ? static_cast<long>(H) * (W / 4) : pair ? static_cast<long>(H) * (W / 2) : static_cast<long>(H) * W);   // <= the scratch size
  if (quad) k_elu_pad_bwd_quad<<<dim3(nb, C, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C, H, W, apply_elu);
  else if (pair) k_elu_pad_bwd_pair<<<dim3(nb, C, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C, H, W, apply_elu);
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
    const int tw = w <= 32 ? 32 : 64;
    const int tiles_x = (w + tw - 1) / tw, tiles_y = (h + UT_H - 1) / UT_H;
    if (al8(gout) && static_cast<unsigned>(tiles_x * tiles_y) <= nb) {     // (the partial-sum scratch holds nb per plane)
      nb = static_cast<unsigned>(tiles_x * tiles_y);
      if (tw == 32) k_elu_up2_cat_pad_bwd_x_tile<32><<<dim3(nb, C1, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C1, C2, h, w, tiles_x);
      else k_elu_up2_cat_pad_bwd_x_tile<64><<<dim3(nb, C1, B), 256, 0, st>>>(x, bias, gout, gx, gbias ? partials : nullptr, C1, C2, h, w, tiles_x);
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
    if (w % 2 == 0 && w >= 4 && al16(gskip)) k_cat_pad_bwd_skip_quad<<<dim3(nblk(1L * h * w), C2, B), 256, 0, st>>>(gout, gskip, C1, C2, 2 * h, 2 * w);
    else if (w >= 2 && al8(gskip)) k_cat_pad_bwd_skip_pair<<<dim3(nblk(2L * h * w), C2, B), 256, 0, st>>>(gout, gskip, C1, C2, 2 * h, 2 * w);
    else k_cat_pad_bwd_skip<<<dim3(nblk(4L * h * w), C2, B), 256, 0, st>>>(gout, gskip, C1, C2, 2 * h, 2 * w);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}
