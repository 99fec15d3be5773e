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
