// Layout-changing casts at the door of MIOpen's bf16 convolutions (the opt-in mixed-precision mode, SURVEY.md 8(f) rank 1;
// never the headline).  MIOpen's bf16 kernels are NHWC implicit GEMMs: handed NCHW tensors they transpose every operand
// themselves (3.4 ms per step) after ATen has already cast it (4 ms per step).  These two kernels do both in one pass:
//     dfe_cast_f32_nchw_to_bf16_nhwc:  y[b][p][c] = bf16_rne(x[b][c][p])          (activations, output gradients, weights)
//     dfe_cast_bf16_nhwc_to_f32_nchw:  y[b][c][p] = float(x[b][p][c])             (convolution results)
// as 32-channel x 64-pixel tile transposes through LDS: 256-byte row reads / 64-byte channel-vector writes (and the
// reverse), 6 bytes per element over the fabric instead of 10.  Rounding = torch's .to(torch.bfloat16) (nearest even, NaN
// -> quiet NaN), so the mode's numerics are what the ATen casts gave.  Bound: HBM.
#include "dfe_internal.h"
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfe {

constexpr int CT_C = 32, CT_P = 64, CT_PITCH = CT_P + 1;

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>(0x7fc0u);   // c10::BFloat16's NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<unsigned short>(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(static_cast<unsigned>(h) << 16); }

struct __attribute__((aligned(16))) Bf8 { unsigned short v[8]; };

// grid (ceil(HW / 64), ceil(C / 32), B), 256 threads
template <bool VEC>
__global__ void __launch_bounds__(256) k_cast_nchw_to_nhwc_bf16(const float* __restrict__ x, unsigned short* __restrict__ y, int C,
                                                               int HW) {
  __shared__ float tile[CT_C * CT_PITCH];
  const int t = threadIdx.x, p0 = blockIdx.x * CT_P, c0 = blockIdx.y * CT_C, b = blockIdx.z;
  const float* xb = x + static_cast<long>(b) * C * HW;
  const int pl = t & 63, cl = t >> 6;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + cl + 4 * k, p = p0 + pl;
    const bool ok = c < C && p < HW;
    v[k] = xb[ok ? static_cast<long>(c) * HW + p : 0L];
    v[k] = ok ? v[k] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[(cl + 4 * k) * CT_PITCH + pl] = v[k];
  __syncthreads();
  const int px = t >> 2, cg = (t & 3) * 8, p = p0 + px;
  if (p >= HW) return;
  unsigned short* yo = y + (static_cast<long>(b) * HW + p) * C + c0 + cg;
  if (VEC) {                      // C % 8 == 0, y 16-byte aligned
    if (c0 + cg < C) {
      Bf8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = f32_to_bf16(tile[(cg + j) * CT_PITCH + px]);
      *reinterpret_cast<Bf8*>(yo) = o;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + cg + j < C) yo[j] = f32_to_bf16(tile[(cg + j) * CT_PITCH + px]);
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_cast_nhwc_bf16_to_nchw(const unsigned short* __restrict__ x, float* __restrict__ y, int C,
                                                               int HW) {
  __shared__ float tile[CT_C * CT_PITCH];
  const int t = threadIdx.x, p0 = blockIdx.x * CT_P, c0 = blockIdx.y * CT_C, b = blockIdx.z;
  const int px = t >> 2, cg = (t & 3) * 8, p = p0 + px;
  const unsigned short* xi = x + (static_cast<long>(b) * HW + min(p, HW - 1)) * C + c0 + cg;
  if (VEC) {
    Bf8 in;
    const bool ok = p < HW && c0 + cg < C;
    in = *reinterpret_cast<const Bf8*>(ok ? xi : x);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[(cg + j) * CT_PITCH + px] = ok ? bf16_to_f32(in.v[j]) : 0.0f;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = p < HW && c0 + cg + j < C;
      const unsigned short h = ok ? xi[j] : static_cast<unsigned short>(0);
      tile[(cg + j) * CT_PITCH + px] = bf16_to_f32(h);
    }
  }
  __syncthreads();
  const int pl = t & 63, cl = t >> 6;
  float* yb = y + static_cast<long>(b) * C * HW;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + cl + 4 * k, pp = p0 + pl;
    if (c < C && pp < HW) yb[static_cast<long>(c) * HW + pp] = tile[(cl + 4 * k) * CT_PITCH + pl];
  }
}

}  // namespace dfe

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)
using namespace dfe;

static int cast_dims(int B, int C, long HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return DFE_ERR_DIMS;
  if (B > 65535 || (C + CT_C - 1) / CT_C > 65535 || HW >= (1L << 31) || static_cast<long>(C) * HW >= (1L << 40)) return DFE_ERR_DIMS;
  return DFE_OK;
}

extern "C" int dfe_cast_f32_nchw_to_bf16_nhwc(const float* x, void* y, int B, int C, long HW, void* stream) {
  if (!x || !y) return DFE_ERR_NULL;
  const int rc = cast_dims(B, C, HW);
  if (rc != DFE_OK) return rc;
  const dim3 g(static_cast<unsigned>((HW + CT_P - 1) / CT_P), (C + CT_C - 1) / CT_C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned short* yo = static_cast<unsigned short*>(y);
  if (C % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) k_cast_nchw_to_nhwc_bf16<true><<<g, 256, 0, st>>>(x, yo, C, static_cast<int>(HW));
  else k_cast_nchw_to_nhwc_bf16<false><<<g, 256, 0, st>>>(x, yo, C, static_cast<int>(HW));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" int dfe_cast_bf16_nhwc_to_f32_nchw(const void* x, float* y, int B, int C, long HW, void* stream) {
  if (!x || !y) return DFE_ERR_NULL;
  const int rc = cast_dims(B, C, HW);
  if (rc != DFE_OK) return rc;
  const dim3 g(static_cast<unsigned>((HW + CT_P - 1) / CT_P), (C + CT_C - 1) / CT_C, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* xi = static_cast<const unsigned short*>(x);
  if (C % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) k_cast_nhwc_bf16_to_nchw<true><<<g, 256, 0, st>>>(xi, y, C, static_cast<int>(HW));
  else k_cast_nhwc_bf16_to_nchw<false><<<g, 256, 0, st>>>(xi, y, C, static_cast<int>(HW));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}
