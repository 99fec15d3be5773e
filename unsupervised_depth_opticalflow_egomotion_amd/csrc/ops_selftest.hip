// Exhaustive on-device verification of the short correctly rounded sequences of loss_stack_exact.h against the IEEE
// operators hipcc expands (12-instruction division, 18-instruction square root).  Takes < 1 s on MI355X.
#include "loss_stack_exact.h"

namespace dfe {

__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// counts[0]: rcp_cr(z) != 1/z over every z with biased exponent in [2, 252]
// counts[1]: sqrt_cr_full(x) != sqrtf(x) over every non-negative finite x (incl. zero and denormal-range arguments)
// counts[2]: div_cr(x, z, rcp_cr(z)) != x/z over `npairs` pseudo-random pairs, z in [2^-12, 2^14), |x| in [2^-20, 2^24)
// counts[3]: the same with every divisor's significand all ones (the excluded case of Markstein's theorem)
__global__ void k_exact_selftest(unsigned long long* __restrict__ counts, unsigned long long npairs) {
  const unsigned long long tid = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  unsigned long long n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  for (unsigned long long i = tid; i < (1ull << 32); i += stride) {
    const unsigned u = static_cast<unsigned>(i), ex = (u >> 23) & 0xff;
    const float z = __uint_as_float(u);
    if (ex >= 2 && ex <= 252) n0 += __float_as_uint(rcp_cr(z)) != __float_as_uint(1.0f / z);
    if (!(u >> 31) && ex != 255) n1 += __float_as_uint(sqrt_cr_full(z)) != __float_as_uint(sqrtf(z));
  }
  for (unsigned long long i = tid; i < npairs; i += stride) {
    const unsigned h1 = mix32(static_cast<unsigned>(i) * 2u + 1u + static_cast<unsigned>(i >> 31));
    const unsigned h2 = mix32(static_cast<unsigned>(i) * 2u + 0x9e3779b9u + static_cast<unsigned>(i >> 32) * 77u);
    const unsigned zx = (127 - 12 + (h2 >> 23) % 26) << 23;
    const float x = __uint_as_float(((127 - 20 + (h1 >> 23) % 44) << 23) | (h1 & 0x7fffffu) | ((h1 >> 8) << 31));
    const float za = __uint_as_float(zx | (h2 & 0x7fffffu)), zb = __uint_as_float(zx | 0x7fffffu);
    n2 += __float_as_uint(div_cr(x, za, rcp_cr(za))) != __float_as_uint(x / za);
    n3 += __float_as_uint(div_cr(x, zb, rcp_cr(zb))) != __float_as_uint(x / zb);
  }
  if (n0) atomicAdd(counts + 0, n0);
  if (n1) atomicAdd(counts + 1, n1);
  if (n2) atomicAdd(counts + 2, n2);
  if (n3) atomicAdd(counts + 3, n3);
}

}  // namespace dfe

extern "C" int dfe_exact_math_selftest(unsigned long long* counts, unsigned long long npairs, void* stream) {
  if (!counts) return DFE_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return DFE_ERR_LAUNCH;
  dfe::k_exact_selftest<<<4096, 256, 0, st>>>(counts, npairs);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}
