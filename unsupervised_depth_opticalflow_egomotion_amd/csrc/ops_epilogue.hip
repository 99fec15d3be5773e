// Convolution epilogue of the flow nets (reference net_utils.py conv(): Conv2d(bias=True) + LeakyReLU(0.1), used by
// FeaturePyramid and PWC_tf).  PyTorch-ROCm runs "conv, broadcast bias add, activation" as three passes forward and
// "activation backward, bias-gradient reduction" as two more; here the convolution is called without its bias and
//   dfe_bias_act_fwd   y = act(z + bias[c])            in place, one read + one write
//   dfe_bias_act_bwd   gz = gy * act'(y);  gbias[c] = sum_{b,h,w} gz     one pass + a fixed-order final sum
// act(v) = v > 0 ? v : slope * v  (slope 0.1 = LeakyReLU, 0 = ReLU, 1 = bias only); act' is taken from the sign of
// the output (same decision as ATen's, which tests the input: slope >= 0 preserves the sign).
// Bound: HBM.  The bias gradient is reduced per block and finished in a fixed order (no atomics: reproducible).
#include "dfe_internal.h"
#include "dfe_device.h"
#include <hip/hip_runtime.h>

namespace dfe {

constexpr int EP_BLOCK = 256;
constexpr int EP_PER_THREAD = 8;                       // elements per thread (two float4)
constexpr int EP_CHUNK = EP_BLOCK * EP_PER_THREAD;     // elements of one (b, c) plane per block

// grid: x = chunk of the plane, y = c, z = b
template <bool VEC>
__global__ void __launch_bounds__(EP_BLOCK) k_bias_act_fwd(float* __restrict__ z, const float* __restrict__ bias,
                                                           int C, int HW, float slope) {
  const int c = plane_id() % C;
  const float bv = bias ? bias[c] : 0.0f;
  float* p = z + static_cast<long>(plane_id()) * HW;
  const int base = blockIdx.x * EP_CHUNK;
  if (VEC) {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD / 4; ++k) {
      const int e = base + (k * EP_BLOCK + threadIdx.x) * 4;
      if (e < HW) {
        float4 v = *reinterpret_cast<float4*>(p + e);
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        v.x = v.x > 0.0f ? v.x : v.x * slope; v.y = v.y > 0.0f ? v.y : v.y * slope;
        v.z = v.z > 0.0f ? v.z : v.z * slope; v.w = v.w > 0.0f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(p + e) = v;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD; ++k) {
      const int e = base + k * EP_BLOCK + threadIdx.x;
      if (e < HW) { const float v = p[e] + bv; p[e] = v > 0.0f ? v : v * slope; }
    }
  }
}

template <bool VEC>
__global__ void __launch_bounds__(EP_BLOCK) k_bias_act_bwd(const float* __restrict__ y, const float* __restrict__ gy,
                                                           float* __restrict__ gz, float* __restrict__ part,
                                                           int HW, long gy_plane_stride, long gy_batch_stride, int C,
                                                           float slope) {
  __shared__ float red[4 * (EP_BLOCK / 64)];
  const int b = plane_id() / C, c = plane_id() - b * C;
  const float* py = y + static_cast<long>(plane_id()) * HW;
  const float* pg = gy + b * gy_batch_stride + c * gy_plane_stride;
  float* pz = gz + static_cast<long>(plane_id()) * HW;
  const int base = blockIdx.x * EP_CHUNK;
  float acc[1] = {0.0f};
  if (VEC) {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD / 4; ++k) {
      const int e = base + (k * EP_BLOCK + threadIdx.x) * 4;
      if (e < HW) {
        const float4 yv = *reinterpret_cast<const float4*>(py + e);
        float4 g = *reinterpret_cast<const float4*>(pg + e);
        g.x = yv.x > 0.0f ? g.x : g.x * slope; g.y = yv.y > 0.0f ? g.y : g.y * slope;
        g.z = yv.z > 0.0f ? g.z : g.z * slope; g.w = yv.w > 0.0f ? g.w : g.w * slope;
        *reinterpret_cast<float4*>(pz + e) = g;
        acc[0] += (g.x + g.y) + (g.z + g.w);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD; ++k) {
      const int e = base + k * EP_BLOCK + threadIdx.x;
      if (e < HW) { const float g = py[e] > 0.0f ? pg[e] : pg[e] * slope; pz[e] = g; acc[0] += g; }
    }
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

// gbias[c] = sum over b, chunks of part[(b*C + c)*nchunk + chunk] in a fixed order; one wave per channel
__global__ void __launch_bounds__(64) k_bias_grad_final(const float* __restrict__ part, float* __restrict__ gbias,
                                                        int B, int C, int nchunk) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = part + (static_cast<long>(b) * C + c) * nchunk;
    for (int k = lane; k < nchunk; k += 64) s += p[k];
  }
  s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x141>(s); s = dpp_add<0x140>(s);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
  if (lane == 0) gbias[c] = (r0 + r1) + (r2 + r3);
}

// The same for up to EP_MAXL layers in ONE launch (a PWC decoder level's five epilogues used to finish their bias
// gradients with five one-wave-per-channel launches of ~4.5 us each, serialised in the flow stream's backward chain).
// grid.x = sum of the layers' channel counts; block = one wave; same summation order as k_bias_grad_final.
constexpr int EP_MAXL = 8;
struct BiasFinalJobs { const float* part[EP_MAXL]; float* gbias[EP_MAXL]; int C[EP_MAXL]; int nchunk[EP_MAXL]; int first[EP_MAXL + 1]; int n; };

__global__ void __launch_bounds__(64) k_bias_grad_final_multi(BiasFinalJobs j, int B) {
  int l = 0;
  while (l + 1 < j.n && static_cast<int>(blockIdx.x) >= j.first[l + 1]) ++l;
  const int c = blockIdx.x - j.first[l], lane = threadIdx.x, C = j.C[l], nchunk = j.nchunk[l];
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = j.part[l] + (static_cast<long>(b) * C + c) * nchunk;
    for (int k = lane; k < nchunk; k += 64) s += p[k];
  }
  s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x141>(s); s = dpp_add<0x140>(s);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
  if (lane == 0) j.gbias[l][c] = (r0 + r1) + (r2 + r3);
}

// ---- the same epilogue for a DenseNet-style block (PWC_tf's decoder, pwc_tf.py:113-117: every layer output is
// consumed through torch.cat by the next two layers).  Forward: read the bias-free convolution output z once and write
// act(z + bias) straight into the channel slices of the (up to two) concatenated buffers that consume it -- dst1 may
// be z itself (in place).  Backward: the output's gradient is the sum of the matching channel slices of the consumers'
// input gradients (g2 optional); y is read from the slice it was written to.  All slices are given by a base pointer
// and a batch stride (channel planes are contiguous inside a sample).
// z, d1 and d2 may alias (d1 == z in place; d1 / d2 slices of one buffer): no __restrict__ on them.
template <bool VEC>
__global__ void __launch_bounds__(EP_BLOCK) k_bias_act_fwd2(const float* z, const float* __restrict__ bias,
                                                            float* d1, long d1_bs, float* d2,
                                                            long d2_bs, int C, int HW, float slope) {
  const int b = plane_id() / C, c = plane_id() - b * C;
  const float bv = bias ? bias[c] : 0.0f;
  const float* p = z + static_cast<long>(plane_id()) * HW;
  float* o1 = d1 + b * d1_bs + static_cast<long>(c) * HW;
  float* o2 = d2 ? d2 + b * d2_bs + static_cast<long>(c) * HW : nullptr;
  const int base = blockIdx.x * EP_CHUNK;
  if (VEC) {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD / 4; ++k) {
      const int e = base + (k * EP_BLOCK + threadIdx.x) * 4;
      if (e < HW) {
        float4 v = *reinterpret_cast<const float4*>(p + e);
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        v.x = v.x > 0.0f ? v.x : v.x * slope; v.y = v.y > 0.0f ? v.y : v.y * slope;
        v.z = v.z > 0.0f ? v.z : v.z * slope; v.w = v.w > 0.0f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(o1 + e) = v;
        if (o2) *reinterpret_cast<float4*>(o2 + e) = v;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD; ++k) {
      const int e = base + k * EP_BLOCK + threadIdx.x;
      if (e < HW) { float v = p[e] + bv; v = v > 0.0f ? v : v * slope; o1[e] = v; if (o2) o2[e] = v; }
    }
  }
}

template <bool VEC>
__global__ void __launch_bounds__(EP_BLOCK) k_bias_act_bwd2(const float* y, long y_bs, const float* g1,
                                                            long g1_bs, const float* g2, long g2_bs,
                                                            float* gz, float* __restrict__ part, int C, int HW,
                                                            float slope) {
  __shared__ float red[4 * (EP_BLOCK / 64)];
  const int b = plane_id() / C, c = plane_id() - b * C;
  const long co = static_cast<long>(c) * HW;
  const float* py = y + b * y_bs + co;
  const float* pa = g1 + b * g1_bs + co;
  const float* pb = g2 ? g2 + b * g2_bs + co : nullptr;
  float* pz = gz + static_cast<long>(plane_id()) * HW;
  const int base = blockIdx.x * EP_CHUNK;
  float acc[1] = {0.0f};
  if (VEC) {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD / 4; ++k) {
      const int e = base + (k * EP_BLOCK + threadIdx.x) * 4;
      if (e < HW) {
        const float4 yv = *reinterpret_cast<const float4*>(py + e);
        float4 g = *reinterpret_cast<const float4*>(pa + e);
        if (pb) { const float4 h = *reinterpret_cast<const float4*>(pb + e); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        g.x = yv.x > 0.0f ? g.x : g.x * slope; g.y = yv.y > 0.0f ? g.y : g.y * slope;
        g.z = yv.z > 0.0f ? g.z : g.z * slope; g.w = yv.w > 0.0f ? g.w : g.w * slope;
        *reinterpret_cast<float4*>(pz + e) = g;
        acc[0] += (g.x + g.y) + (g.z + g.w);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < EP_PER_THREAD; ++k) {
      const int e = base + k * EP_BLOCK + threadIdx.x;
      if (e < HW) {
        float g = pa[e];
        if (pb) g += pb[e];
        g = py[e] > 0.0f ? g : g * slope;
        pz[e] = g; acc[0] += g;
      }
    }
  }
  if (part) block_sum<1>(acc, red, part + static_cast<long>(plane_id()) * gridDim.x + blockIdx.x);
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" long dfe_bias_act_partials_floats(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const long hw = static_cast<long>(H) * W;
  return static_cast<long>(B) * C * ((hw + EP_CHUNK - 1) / EP_CHUNK);
}

extern "C" int dfe_bias_act_fwd(float* z, const float* bias, int B, int C, int H, int W, float slope, void* stream) {
  if (!z) return DFE_ERR_NULL;
  const long hw = static_cast<long>(H) * W;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31) || C > 65535 || B > 65535) return DFE_ERR_DIMS;
  const dim3 g(static_cast<unsigned>((hw + EP_CHUNK - 1) / EP_CHUNK), C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hw % 4 == 0 && aligned16(z)) k_bias_act_fwd<true><<<g, EP_BLOCK, 0, st>>>(z, bias, C, static_cast<int>(hw), slope);
  else k_bias_act_fwd<false><<<g, EP_BLOCK, 0, st>>>(z, bias, C, static_cast<int>(hw), slope);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_bias_act_bwd(const float* y, const float* gy, long gy_batch_stride, float* gz, float* gbias,
                                float* partials, int B, int C, int H, int W, float slope, void* stream) {
  if (!y || !gy || !gz || (gbias && !partials)) return DFE_ERR_NULL;
  const long hw = static_cast<long>(H) * W;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31) || C > 65535 || B > 65535) return DFE_ERR_DIMS;
  if (gy_batch_stride < static_cast<long>(C) * hw) return DFE_ERR_DIMS;
  const int nchunk = static_cast<int>((hw + EP_CHUNK - 1) / EP_CHUNK);
  const dim3 g(nchunk, C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = partials;      // written whenever given: gbias == NULL leaves the finish to dfe_bias_grad_final_multi
  const bool vec = hw % 4 == 0 && aligned16(y) && aligned16(gy) && aligned16(gz) && gy_batch_stride % 4 == 0;
  if (vec) k_bias_act_bwd<true><<<g, EP_BLOCK, 0, st>>>(y, gy, gz, part, static_cast<int>(hw), hw, gy_batch_stride, C, slope);
  else k_bias_act_bwd<false><<<g, EP_BLOCK, 0, st>>>(y, gy, gz, part, static_cast<int>(hw), hw, gy_batch_stride, C, slope);
  DFE_LAUNCH_CHECK();
  if (gbias) {
    k_bias_grad_final<<<C, 64, 0, st>>>(partials, gbias, B, C, nchunk);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

extern "C" int dfe_bias_act_fwd2(const float* z, const float* bias, float* dst1, long dst1_batch_stride, float* dst2,
                                 long dst2_batch_stride, int B, int C, int H, int W, float slope, void* stream) {
  if (!z || !dst1) return DFE_ERR_NULL;
  const long hw = static_cast<long>(H) * W;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31) || C > 65535 || B > 65535) return DFE_ERR_DIMS;
  if (dst1_batch_stride < C * hw || (dst2 && dst2_batch_stride < C * hw)) return DFE_ERR_DIMS;
  const dim3 g(static_cast<unsigned>((hw + EP_CHUNK - 1) / EP_CHUNK), C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = hw % 4 == 0 && aligned16(z) && aligned16(dst1) && dst1_batch_stride % 4 == 0 &&
                   (!dst2 || (aligned16(dst2) && dst2_batch_stride % 4 == 0));
  if (vec) k_bias_act_fwd2<true><<<g, EP_BLOCK, 0, st>>>(z, bias, dst1, dst1_batch_stride, dst2, dst2_batch_stride, C, static_cast<int>(hw), slope);
  else k_bias_act_fwd2<false><<<g, EP_BLOCK, 0, st>>>(z, bias, dst1, dst1_batch_stride, dst2, dst2_batch_stride, C, static_cast<int>(hw), slope);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_bias_act_bwd2(const float* y, long y_batch_stride, const float* g1, long g1_batch_stride, const float* g2,
                                 long g2_batch_stride, float* gz, float* gbias, float* partials, int B, int C, int H, int W,
                                 float slope, void* stream) {
  if (!y || !g1 || !gz || (gbias && !partials)) return DFE_ERR_NULL;
  const long hw = static_cast<long>(H) * W;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31) || C > 65535 || B > 65535) return DFE_ERR_DIMS;
  if (y_batch_stride < C * hw || g1_batch_stride < C * hw || (g2 && g2_batch_stride < C * hw)) return DFE_ERR_DIMS;
  const int nchunk = static_cast<int>((hw + EP_CHUNK - 1) / EP_CHUNK);
  const dim3 g(nchunk, C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = partials;      // written whenever given: gbias == NULL leaves the finish to dfe_bias_grad_final_multi
  const bool vec = hw % 4 == 0 && aligned16(y) && aligned16(g1) && aligned16(gz) && y_batch_stride % 4 == 0 &&
                   g1_batch_stride % 4 == 0 && (!g2 || (aligned16(g2) && g2_batch_stride % 4 == 0));
  if (vec) k_bias_act_bwd2<true><<<g, EP_BLOCK, 0, st>>>(y, y_batch_stride, g1, g1_batch_stride, g2, g2_batch_stride, gz, part, C, static_cast<int>(hw), slope);
  else k_bias_act_bwd2<false><<<g, EP_BLOCK, 0, st>>>(y, y_batch_stride, g1, g1_batch_stride, g2, g2_batch_stride, gz, part, C, static_cast<int>(hw), slope);
  DFE_LAUNCH_CHECK();
  if (gbias) {
    k_bias_grad_final<<<C, 64, 0, st>>>(partials, gbias, B, C, nchunk);
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

extern "C" int dfe_bias_grad_final_multi(const float* const* partials, float* const* gbias, const int* C, int n, int B, int H, int W,
                                         void* stream) {
  if (!partials || !gbias || !C) return DFE_ERR_NULL;
  const long hw = static_cast<long>(H) * W;
  if (n <= 0 || n > EP_MAXL || B <= 0 || H <= 0 || W <= 0 || hw >= (1L << 31)) return DFE_ERR_DIMS;
  BiasFinalJobs j;
  j.n = n; j.first[0] = 0;
  for (int l = 0; l < n; ++l) {
    if (!partials[l] || !gbias[l]) return DFE_ERR_NULL;
    if (C[l] <= 0 || C[l] > 65535) return DFE_ERR_DIMS;
    j.part[l] = partials[l]; j.gbias[l] = gbias[l]; j.C[l] = C[l];
    j.nchunk[l] = static_cast<int>((hw + EP_CHUNK - 1) / EP_CHUNK);
    j.first[l + 1] = j.first[l] + C[l];
  }
  k_bias_grad_final_multi<<<j.first[n], 64, 0, static_cast<hipStream_t>(stream)>>>(j, B);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
