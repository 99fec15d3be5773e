// Grouped training-mode BatchNorm2d (+ residual add + ReLU) for the depth encoder (reference depth_model.py:60-95: a
// torchvision ResNet-18 whose BasicBlock is conv-bn-relu-conv-bn-(+identity)-relu).
//
// The reference runs the depth net once per frame of the triplet (model_geometry.py:786-788), so its BatchNorm
// statistics are those of one frame's B samples.  To run the three frames as ONE batch of 3B (larger convolutions, a
// third of the launches, no gradient-accumulation adds) the normalisation must keep that meaning: the batch is G
// groups of Bg consecutive samples, statistics are per (group, channel), and the running statistics receive the G
// momentum updates in group order -- exactly what G sequential calls do.  G = 1 is plain BatchNorm2d.
//
//   dfe_bn_fwd:  mean/var per (g,c) -> y = act((x - mean) * invstd * w + b [+ residual]),  act = ReLU or identity
//   dfe_bn_bwd:  g' = gy * [y > 0];  gw = sum g' * xhat, gb = sum g';  gx = w * invstd * (g' - mean(g') - xhat * mean(g' xhat));
//                gres = g'
// Two passes over x each way (statistics, then apply): the same traffic as an unfused batch norm, with the ReLU, its
// backward and the residual add folded in.  Statistics: per-block (mean, M2) merged with Chan's formula in double in
// a fixed order (no atomics; reproducible).  Bound: HBM.
#include "dfe_internal.h"
#include "dfe_device.h"
#include <hip/hip_runtime.h>

namespace dfe {

constexpr int BN_BLOCK = 256;
constexpr int BN_PER_THREAD = 8;
constexpr int BN_CHUNK = BN_BLOCK * BN_PER_THREAD;

// sum over the block, returned to every thread (fixed order); smem: 4 * (BN_BLOCK / 64) floats
__device__ __forceinline__ float block_allsum(float v, float* smem) {
  v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 15) == 0) smem[wave * 4 + (lane >> 4)] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 4 * (BN_BLOCK / 64); ++k) s += smem[k];
  __syncthreads();
  return s;
}

template <bool VEC>
__device__ __forceinline__ void load8(const float* __restrict__ p, int base, int HW, float (&v)[BN_PER_THREAD], bool (&ok)[BN_PER_THREAD]) {
  if (VEC) {
#pragma unroll
    for (int k = 0; k < BN_PER_THREAD / 4; ++k) {
      const int e = base + (k * BN_BLOCK + threadIdx.x) * 4;
      const bool in = e < HW;
      float4 q = in ? *reinterpret_cast<const float4*>(p + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      ok[4 * k] = ok[4 * k + 1] = ok[4 * k + 2] = ok[4 * k + 3] = in;
    }
  } else {
#pragma unroll
    for (int k = 0; k < BN_PER_THREAD; ++k) {
      const int e = base + k * BN_BLOCK + threadIdx.x;
      ok[k] = e < HW;
      v[k] = ok[k] ? p[e] : 0.0f;
    }
  }
}

template <bool VEC>
__device__ __forceinline__ void store8(float* __restrict__ p, int base, int HW, const float (&v)[BN_PER_THREAD]) {
  if (VEC) {
#pragma unroll
    for (int k = 0; k < BN_PER_THREAD / 4; ++k) {
      const int e = base + (k * BN_BLOCK + threadIdx.x) * 4;
      if (e < HW) *reinterpret_cast<float4*>(p + e) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < BN_PER_THREAD; ++k) {
      const int e = base + k * BN_BLOCK + threadIdx.x;
      if (e < HW) p[e] = v[k];
    }
  }
}

// ---- forward pass 1: per block (count, mean, M2) of its chunk of plane (n, c).  grid: (nchunk, C, N)
template <bool VEC>
__global__ void __launch_bounds__(BN_BLOCK) k_bn_stats(const float* __restrict__ x, float* __restrict__ part, int HW) {
  __shared__ float smem[4 * (BN_BLOCK / 64)];
  const int base = blockIdx.x * BN_CHUNK;
  float v[BN_PER_THREAD]; bool ok[BN_PER_THREAD];
  load8<VEC>(x + static_cast<long>(plane_id()) * HW, base, HW, v, ok);
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < BN_PER_THREAD; ++k) s += v[k];
  const float cnt = static_cast<float>(min(BN_CHUNK, HW - base));
  const float mean = block_allsum(s, smem) / cnt;
  float m2 = 0.0f;
#pragma unroll
  for (int k = 0; k < BN_PER_THREAD; ++k) { const float d = ok[k] ? v[k] - mean : 0.0f; m2 += d * d; }
  m2 = block_allsum(m2, smem);
  if (threadIdx.x == 0) {
    float* o = part + (static_cast<long>(plane_id()) * gridDim.x + blockIdx.x) * 3;
    o[0] = cnt; o[1] = mean; o[2] = m2;
  }
}

// ---- forward pass 1b: one wave per channel.  For each group the lanes take the partials round-robin, merge their
// own sequentially and then pairwise across the wave (Chan's parallel update, double; a fixed tree -> reproducible);
// lane 0 writes mean / invstd of (g,c) and applies the G running-statistics updates in group order.
struct Moments { double n, mean, m2; };
__device__ __forceinline__ Moments merge(const Moments& a, const Moments& b) {
  if (b.n == 0.0) return a;
  if (a.n == 0.0) return b;
  Moments r;
  r.n = a.n + b.n;
  const double delta = b.mean - a.mean;
  r.mean = a.mean + delta * b.n / r.n;
  r.m2 = a.m2 + b.m2 + delta * delta * a.n * b.n / r.n;
  return r;
}

__global__ void __launch_bounds__(64) k_bn_finalize(const float* __restrict__ part, float* __restrict__ mean_out,
                                                    float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, int G, int Bg, int C, int nchunk,
                                                    float eps, float momentum) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
  for (int g = 0; g < G; ++g) {
    Moments m{0.0, 0.0, 0.0};
    for (int i = lane; i < Bg * nchunk; i += 64) {
      const int b = i / nchunk, k = i - b * nchunk;
      const float* p = part + (((static_cast<long>(g) * Bg + b) * C + c) * nchunk + k) * 3;
      m = merge(m, Moments{static_cast<double>(p[0]), static_cast<double>(p[1]), static_cast<double>(p[2])});
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      Moments o{__shfl_xor(m.n, off), __shfl_xor(m.mean, off), __shfl_xor(m.m2, off)};
      m = (lane & off) ? merge(o, m) : merge(m, o);      // both partners form the same ordered pair
    }
    if (lane == 0) {
      const double var = m.m2 / m.n;
      mean_out[g * C + c] = static_cast<float>(m.mean);
      invstd_out[g * C + c] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
      // what nn.BatchNorm2d does per call, in fp32: running = (1 - m) * running + m * stat (variance unbiased)
      const float fm = static_cast<float>(m.mean), fv = static_cast<float>(m.n > 1.0 ? m.m2 / (m.n - 1.0) : var);
      rm = (1.0f - momentum) * rm + momentum * fm;
      rv = (1.0f - momentum) * rv + momentum * fv;
    }
  }
  if (lane == 0) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

// ---- forward pass 2: y = act((x - mean) * (invstd * w) + b [+ res]).  grid: (nchunk, C, N)
template <bool VEC>
__global__ void __launch_bounds__(BN_BLOCK) k_bn_apply(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ weight, const float* __restrict__ bias,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       float* __restrict__ y, int Bg, int C, int HW, int relu) {
  const int n = plane_id() / C, c = plane_id() - n * C, g = n / Bg;
  const float mu = mean[g * C + c], sc = invstd[g * C + c] * (weight ? weight[c] : 1.0f), sh = bias ? bias[c] : 0.0f;
  const int base = blockIdx.x * BN_CHUNK;
  const long off = static_cast<long>(plane_id()) * HW;
  float v[BN_PER_THREAD], r[BN_PER_THREAD]; bool ok[BN_PER_THREAD];
  load8<VEC>(x + off, base, HW, v, ok);
  if (res) load8<VEC>(res + off, base, HW, r, ok);
#pragma unroll
  for (int k = 0; k < BN_PER_THREAD; ++k) {
    float o = (v[k] - mu) * sc + sh;
    if (res) o += r[k];
    v[k] = (relu && o <= 0.0f) ? 0.0f : o;
  }
  store8<VEC>(y + off, base, HW, v);
}

// ---- backward pass 1: per block sums of g' and g' * xhat (g' = gy masked by the ReLU).  grid: (nchunk, C, N)
template <bool VEC>
__global__ void __launch_bounds__(BN_BLOCK) k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ gy, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ part,
                                                            int Bg, int C, int HW, int relu) {
  __shared__ float red[2 * 4 * (BN_BLOCK / 64)];
  const int n = plane_id() / C, c = plane_id() - n * C, g = n / Bg;
  const float mu = mean[g * C + c], is = invstd[g * C + c];
  const int base = blockIdx.x * BN_CHUNK;
  const long off = static_cast<long>(plane_id()) * HW;
  float v[BN_PER_THREAD], gg[BN_PER_THREAD], yy[BN_PER_THREAD]; bool ok[BN_PER_THREAD];
  load8<VEC>(x + off, base, HW, v, ok);
  load8<VEC>(gy + off, base, HW, gg, ok);
  if (relu) load8<VEC>(y + off, base, HW, yy, ok);
  float acc[2] = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < BN_PER_THREAD; ++k) {
    const float gm = (relu && yy[k] <= 0.0f) ? 0.0f : gg[k];
    acc[0] += gm;
    acc[1] += gm * ((v[k] - mu) * is);
  }
  block_sum<2>(acc, red, part + (static_cast<long>(plane_id()) * gridDim.x + blockIdx.x) * 2);
}

// ---- backward pass 1b: one wave per channel: per group means of g' and g' xhat, gw / gb over all groups (double,
// lanes round-robin over the partials, then a fixed butterfly)
__global__ void __launch_bounds__(64) k_bn_bwd_finalize(const float* __restrict__ part, float* __restrict__ gmean,
                                                        float* __restrict__ gxmean, float* __restrict__ gweight,
                                                        float* __restrict__ gbias, int G, int Bg, int C, int nchunk, int HW) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double gw = 0.0, gb = 0.0;
  for (int g = 0; g < G; ++g) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = lane; i < Bg * nchunk; i += 64) {
      const int b = i / nchunk, k = i - b * nchunk;
      const float* p = part + (((static_cast<long>(g) * Bg + b) * C + c) * nchunk + k) * 2;
      s0 += p[0]; s1 += p[1];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if (lane == 0) {
      const double cnt = static_cast<double>(Bg) * HW;
      gmean[g * C + c] = static_cast<float>(s0 / cnt);
      gxmean[g * C + c] = static_cast<float>(s1 / cnt);
    }
    gb += s0; gw += s1;
  }
  if (lane == 0) {
    if (gweight) gweight[c] = static_cast<float>(gw);
    if (gbias) gbias[c] = static_cast<float>(gb);
  }
}

// ---- backward pass 2: gx = w * invstd * (g' - mean(g') - xhat * mean(g' xhat));  gres = g'.  grid: (nchunk, C, N)
template <bool VEC>
__global__ void __launch_bounds__(BN_BLOCK) k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ gy, const float* __restrict__ weight,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gmean, const float* __restrict__ gxmean,
                                                           float* __restrict__ gx, float* __restrict__ gres, int Bg, int C,
                                                           int HW, int relu) {
  const int n = plane_id() / C, c = plane_id() - n * C, g = n / Bg;
  const float mu = mean[g * C + c], is = invstd[g * C + c], ws = (weight ? weight[c] : 1.0f) * is;
  const float m0 = gmean[g * C + c], m1 = gxmean[g * C + c];
  const int base = blockIdx.x * BN_CHUNK;
  const long off = static_cast<long>(plane_id()) * HW;
  float v[BN_PER_THREAD], gg[BN_PER_THREAD], yy[BN_PER_THREAD]; bool ok[BN_PER_THREAD];
  load8<VEC>(x + off, base, HW, v, ok);
  load8<VEC>(gy + off, base, HW, gg, ok);
  if (relu) load8<VEC>(y + off, base, HW, yy, ok);
#pragma unroll
  for (int k = 0; k < BN_PER_THREAD; ++k) {
    const float gm = (relu && yy[k] <= 0.0f) ? 0.0f : gg[k];
    gg[k] = gm;
    v[k] = ws * ((gm - m0) - ((v[k] - mu) * is) * m1);
  }
  store8<VEC>(gx + off, base, HW, v);
  if (gres) store8<VEC>(gres + off, base, HW, gg);
}

// ---- small planes (H*W <= 4096, Bg <= 4: the 32x104, 16x52 and 8x26 stages of the encoder -- 15 of its 20
// normalisations): ONE kernel per direction instead of three.  A 1024-thread block owns a channel and walks its G
// groups; a group's Bg x HW elements sit in registers (one float4 per sample and thread), so x is read once, the
// statistics are the exact two-pass ones (mean, then sum of squared deviations; both block sums in a fixed order) and
// the running statistics receive their G updates in group order from one thread.  These layers are launch- and
// latency-bound: 3 launches of 5-10 us each become one.
constexpr int BNS_THREADS = 1024, BNS_MAXB = 4, BNS_MAXHW = BNS_THREADS * 4;

// fixed-order sums of NV values over a block of T threads, returned to every thread; smem: NV * 4 * (T / 64) floats
template <int NV, int T>
__device__ __forceinline__ void bns_allsum(float (&v)[NV], float* smem) {
#pragma unroll
  for (int i = 0; i < NV; ++i) { v[i] = dpp_add<0xB1>(v[i]); v[i] = dpp_add<0x4E>(v[i]); v[i] = dpp_add<0x141>(v[i]); v[i] = dpp_add<0x140>(v[i]); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SL = 4 * (T / 64);
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[i * SL + wave * 4 + (lane >> 4)] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < SL; ++k) s += smem[i * SL + k];
    v[i] = s;
  }
  __syncthreads();
}

// GB groups are processed together (their loads are in flight at once and share the two block reductions); the
// host picks GB = 3 for the triplet's G = 3, else 1, and the block size T = 64 / 256 / 1024 that covers HW / 4
// (8x26 -> one wave per channel, no LDS traffic to speak of; measured with 1024 threads everywhere: 17.6 us for the
// 512-channel 8x26 layer, slower than the 128-channel 32x104 one).
template <int GB, int T>
__global__ void __launch_bounds__(T) k_bn_fwd_small(const float* __restrict__ x, const float* __restrict__ res,
                                                              const float* __restrict__ weight, const float* __restrict__ bias,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ y, float* __restrict__ mean_out,
                                                              float* __restrict__ invstd_out, int G, int Bg, int C, int HW,
                                                              float eps, float momentum, int relu) {
  __shared__ float smem[GB * 4 * (T / 64)];
  const int c = blockIdx.x, e = threadIdx.x * 4;
  const bool in = e < HW;
  const float w = weight ? weight[c] : 1.0f, sh = bias ? bias[c] : 0.0f;
  const float cnt = static_cast<float>(Bg) * static_cast<float>(HW);
  float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
  for (int g0 = 0; g0 < G; g0 += GB) {
    float4 v[GB][BNS_MAXB];
    float s[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      s[q] = 0.0f;
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b) {
        v[q][b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g0 + q < G && b < Bg && in) v[q][b] = *reinterpret_cast<const float4*>(x + (static_cast<long>((g0 + q) * Bg + b) * C + c) * HW + e);
      }
    }
#pragma unroll
    for (int q = 0; q < GB; ++q)
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b) s[q] += (v[q][b].x + v[q][b].y) + (v[q][b].z + v[q][b].w);
    bns_allsum<GB, T>(s, smem);
    float mu[GB], m2[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      mu[q] = s[q] / cnt;
      m2[q] = 0.0f;
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b)
        if (g0 + q < G && b < Bg && in) {
          const float d0 = v[q][b].x - mu[q], d1 = v[q][b].y - mu[q], d2 = v[q][b].z - mu[q], d3 = v[q][b].w - mu[q];
          m2[q] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    bns_allsum<GB, T>(m2, smem);
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int g = g0 + q;
      if (g >= G) break;
      const float var = m2[q] / cnt;
      const float is = static_cast<float>(1.0 / sqrt(static_cast<double>(var) + static_cast<double>(eps)));
      if (threadIdx.x == 0) {
        mean_out[g * C + c] = mu[q];
        invstd_out[g * C + c] = is;
        const float fv = cnt > 1.0f ? m2[q] / (cnt - 1.0f) : var;     // unbiased, as nn.BatchNorm2d tracks it
        rm = (1.0f - momentum) * rm + momentum * mu[q];
        rv = (1.0f - momentum) * rv + momentum * fv;
      }
      const float sc = is * w;
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b)
        if (b < Bg && in) {
          const long off = (static_cast<long>(g * Bg + b) * C + c) * HW + e;
          const float4 t = v[q][b];
          float o[4] = {(t.x - mu[q]) * sc + sh, (t.y - mu[q]) * sc + sh, (t.z - mu[q]) * sc + sh, (t.w - mu[q]) * sc + sh};
          if (res) { const float4 r = *reinterpret_cast<const float4*>(res + off); o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
          if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] <= 0.0f ? 0.0f : o[k];
          }
          *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
  }
  if (threadIdx.x == 0) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

template <int GB, int T>
__global__ void __launch_bounds__(T) k_bn_bwd_small(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ gy, const float* __restrict__ weight,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ gx, float* __restrict__ gres,
                                                              float* __restrict__ gweight, float* __restrict__ gbias,
                                                              int G, int Bg, int C, int HW, int relu) {
  __shared__ float smem[2 * GB * 4 * (T / 64)];
  const int c = blockIdx.x, e = threadIdx.x * 4;
  const bool in = e < HW;
  const float w = weight ? weight[c] : 1.0f;
  const float cnt = static_cast<float>(Bg) * static_cast<float>(HW);
  double gw = 0.0, gb = 0.0;
  for (int g0 = 0; g0 < G; g0 += GB) {
    float4 xh[GB][BNS_MAXB], gm[GB][BNS_MAXB];
    float mu[GB], is[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int g = min(g0 + q, G - 1);
      mu[q] = mean[g * C + c]; is[q] = invstd[g * C + c];
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b) {
        xh[q][b] = make_float4(0.f, 0.f, 0.f, 0.f); gm[q][b] = xh[q][b];
        if (g0 + q < G && b < Bg && in) {
          const long off = (static_cast<long>((g0 + q) * Bg + b) * C + c) * HW + e;
          xh[q][b] = *reinterpret_cast<const float4*>(x + off);
          float4 gv = *reinterpret_cast<const float4*>(gy + off);
          if (relu) {
            const float4 yv = *reinterpret_cast<const float4*>(y + off);
            if (yv.x <= 0.0f) gv.x = 0.0f;
            if (yv.y <= 0.0f) gv.y = 0.0f;
            if (yv.z <= 0.0f) gv.z = 0.0f;
            if (yv.w <= 0.0f) gv.w = 0.0f;
          }
          gm[q][b] = gv;
        }
      }
    }
    float acc[2 * GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      acc[2 * q] = 0.0f; acc[2 * q + 1] = 0.0f;
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b)
        if (g0 + q < G && b < Bg && in) {
          float4& t = xh[q][b];
          t = make_float4((t.x - mu[q]) * is[q], (t.y - mu[q]) * is[q], (t.z - mu[q]) * is[q], (t.w - mu[q]) * is[q]);
          const float4 gv = gm[q][b];
          acc[2 * q] += (gv.x + gv.y) + (gv.z + gv.w);
          acc[2 * q + 1] += (gv.x * t.x + gv.y * t.y) + (gv.z * t.z + gv.w * t.w);
        }
    }
    bns_allsum<2 * GB, T>(acc, smem);
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int g = g0 + q;
      if (g >= G) break;
      gb += acc[2 * q]; gw += acc[2 * q + 1];
      const float m0 = acc[2 * q] / cnt, m1 = acc[2 * q + 1] / cnt, ws = w * is[q];
#pragma unroll
      for (int b = 0; b < BNS_MAXB; ++b)
        if (b < Bg && in) {
          const long off = (static_cast<long>(g * Bg + b) * C + c) * HW + e;
          const float4 t = xh[q][b], gv = gm[q][b];
          *reinterpret_cast<float4*>(gx + off) = make_float4(ws * ((gv.x - m0) - t.x * m1), ws * ((gv.y - m0) - t.y * m1),
                                                             ws * ((gv.z - m0) - t.z * m1), ws * ((gv.w - m0) - t.w * m1));
          if (gres) *reinterpret_cast<float4*>(gres + off) = gv;
        }
    }
  }
  if (threadIdx.x == 0) {
    if (gweight) gweight[c] = static_cast<float>(gw);
    if (gbias) gbias[c] = static_cast<float>(gb);
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int bn_nchunk(long hw) { return static_cast<int>((hw + BN_CHUNK - 1) / BN_CHUNK); }

static int bn_dims(int G, int Bg, int C, int H, int W) {
  const long hw = static_cast<long>(H) * W;
  if (G <= 0 || Bg <= 0 || C <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31) || C > 65535 || static_cast<long>(G) * Bg > 65535) return DFE_ERR_DIMS;
  if (static_cast<long>(Bg) * hw < 2) return DFE_ERR_DIMS;      // "Expected more than 1 value per channel when training"
  return DFE_OK;
}

extern "C" long dfe_bn_partials_floats(int G, int Bg, int C, int H, int W) {
  if (bn_dims(G, Bg, C, H, W) != DFE_OK) return 0;
  return static_cast<long>(G) * Bg * C * bn_nchunk(static_cast<long>(H) * W) * 3;
}

extern "C" int dfe_bn_fwd(const float* x, const float* residual, const float* weight, const float* bias, float* running_mean,
                          float* running_var, float* y, float* save_mean, float* save_invstd, float* partials, int G, int Bg,
                          int C, int H, int W, float eps, float momentum, int relu, void* stream) {
  if (!x || !y || !save_mean || !save_invstd || !partials) return DFE_ERR_NULL;
  int rc = bn_dims(G, Bg, C, H, W);
  if (rc != DFE_OK) return rc;
  const int hw = H * W, nchunk = bn_nchunk(hw);
  const dim3 grid(nchunk, C, G * Bg);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = hw % 4 == 0 && al16(x) && al16(y) && (!residual || al16(residual));
  if (vec && hw <= BNS_MAXHW && Bg <= BNS_MAXB) {
#define DFE_BNS_FWD(GB, T) k_bn_fwd_small<GB, T><<<C, T, 0, st>>>(x, residual, weight, bias, running_mean, running_var, y, \
                                                                save_mean, save_invstd, G, Bg, C, hw, eps, momentum, relu)
    if (G == 3) { if (hw <= 256) DFE_BNS_FWD(3, 64); else if (hw <= 1024) DFE_BNS_FWD(3, 256); else DFE_BNS_FWD(3, 1024); }
    else { if (hw <= 256) DFE_BNS_FWD(1, 64); else if (hw <= 1024) DFE_BNS_FWD(1, 256); else DFE_BNS_FWD(1, 1024); }
#undef DFE_BNS_FWD
    DFE_LAUNCH_CHECK();
    return DFE_OK;
  }
  if (vec) k_bn_stats<true><<<grid, BN_BLOCK, 0, st>>>(x, partials, hw);
  else k_bn_stats<false><<<grid, BN_BLOCK, 0, st>>>(x, partials, hw);
  DFE_LAUNCH_CHECK();
  k_bn_finalize<<<C, 64, 0, st>>>(partials, save_mean, save_invstd, running_mean, running_var, G, Bg, C, nchunk, eps, momentum);
  DFE_LAUNCH_CHECK();
  if (vec) k_bn_apply<true><<<grid, BN_BLOCK, 0, st>>>(x, residual, weight, bias, save_mean, save_invstd, y, Bg, C, hw, relu);
  else k_bn_apply<false><<<grid, BN_BLOCK, 0, st>>>(x, residual, weight, bias, save_mean, save_invstd, y, Bg, C, hw, relu);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_bn_bwd(const float* x, const float* y, const float* gy, const float* weight, const float* save_mean,
                          const float* save_invstd, float* gx, float* gres, float* gweight, float* gbias, float* partials,
                          float* scratch_means, int G, int Bg, int C, int H, int W, int relu, void* stream) {
  if (!x || !gy || !save_mean || !save_invstd || !gx || !partials || !scratch_means || (relu && !y)) return DFE_ERR_NULL;
  int rc = bn_dims(G, Bg, C, H, W);
  if (rc != DFE_OK) return rc;
  const int hw = H * W, nchunk = bn_nchunk(hw);
  const dim3 grid(nchunk, C, G * Bg);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* gmean = scratch_means;
  float* gxmean = scratch_means + static_cast<long>(G) * C;
  const bool vec = hw % 4 == 0 && al16(x) && al16(gy) && al16(gx) && (!relu || al16(y)) && (!gres || al16(gres));
  if (vec && hw <= BNS_MAXHW && Bg <= BNS_MAXB) {
#define DFE_BNS_BWD(GB, T) k_bn_bwd_small<GB, T><<<C, T, 0, st>>>(x, y, gy, weight, save_mean, save_invstd, gx, gres, gweight, \
                                                                gbias, G, Bg, C, hw, relu)
    if (G == 3) { if (hw <= 256) DFE_BNS_BWD(3, 64); else if (hw <= 1024) DFE_BNS_BWD(3, 256); else DFE_BNS_BWD(3, 1024); }
    else { if (hw <= 256) DFE_BNS_BWD(1, 64); else if (hw <= 1024) DFE_BNS_BWD(1, 256); else DFE_BNS_BWD(1, 1024); }
#undef DFE_BNS_BWD
    DFE_LAUNCH_CHECK();
    return DFE_OK;
  }
  if (vec) k_bn_bwd_reduce<true><<<grid, BN_BLOCK, 0, st>>>(x, y, gy, save_mean, save_invstd, partials, Bg, C, hw, relu);
  else k_bn_bwd_reduce<false><<<grid, BN_BLOCK, 0, st>>>(x, y, gy, save_mean, save_invstd, partials, Bg, C, hw, relu);
  DFE_LAUNCH_CHECK();
  k_bn_bwd_finalize<<<C, 64, 0, st>>>(partials, gmean, gxmean, gweight, gbias, G, Bg, C, nchunk, hw);
  DFE_LAUNCH_CHECK();
  if (vec) k_bn_bwd_apply<true><<<grid, BN_BLOCK, 0, st>>>(x, y, gy, weight, save_mean, save_invstd, gmean, gxmean, gx, gres, Bg, C, hw, relu);
  else k_bn_bwd_apply<false><<<grid, BN_BLOCK, 0, st>>>(x, y, gy, weight, save_mean, save_invstd, gmean, gxmean, gx, gres, Bg, C, hw, relu);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
