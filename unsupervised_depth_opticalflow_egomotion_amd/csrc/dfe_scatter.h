// Order-independent scatter-add (the adjoint of a bilinear gather with respect to the sampled tensor).
//
// A float atomicAdd makes the sum depend on the order the hardware retires the atomics in: the result changes from run
// to run.  Here every contribution is first scaled by a power of two (exact) and rounded ONCE to a 64-bit integer,
// the integers are added with 64-bit integer atomics -- integer addition is associative and commutative, so the sum is
// the same whatever the order -- and one pass converts the sums back to float (one more rounding).  The result is
// bitwise reproducible and at least as accurate as a float accumulation: the quantum is 2^-SCATTER_BITS of the largest
// contribution's binade, against fp32's 2^-24 of the running sum.
//
// Scale: 2^(SCATTER_BITS - k) with |largest contribution| = m * 2^k, m in [0.5, 1).  Each scaled contribution is below
// 2^SCATTER_BITS = 2^36 in magnitude, so up to 2^26 of them can meet in one element before an int64 could overflow
// (a 4-tap scatter of a 4096 x 4096 map into one pixel).  The bound comes either from a max-reduction over the gradient
// that is being scattered (k_scatter_amax: max is order-independent too) or from an analytic bound of the caller.
//
// Workspace layout (dfe_scatter_ws_bytes): a 64-byte header (word 0: bit pattern of the bound, a non-negative float)
// followed by one int64 accumulator per element.  The library zero-fills it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfe {

constexpr int SCATTER_BITS = 36;
constexpr long SCATTER_HEADER_BYTES = 64;

struct ScatterScale {
  float to_fixed;    // 2^(SCATTER_BITS - k); 0 when the bound is 0 (nothing to add); NaN when the bound is not finite
  double to_float;   // its reciprocal
};

// bound_bits: bit pattern of a non-negative float (or of a NaN / Inf: the result is then NaN everywhere)
__device__ __forceinline__ ScatterScale scatter_scale(unsigned bound_bits) {
  ScatterScale s;
  if (bound_bits >= 0x7f800000u) { s.to_fixed = __int_as_float(0x7fc00000); s.to_float = 0.0; return s; }
  if (bound_bits == 0u) { s.to_fixed = 0.0f; s.to_float = 0.0; return s; }
  int k;
  (void)frexpf(__int_as_float(static_cast<int>(bound_bits)), &k);     // bound = m * 2^k, m in [0.5, 1)
  int e = SCATTER_BITS - k;
  e = e > 126 ? 126 : (e < -126 ? -126 : e);
  s.to_fixed = ldexpf(1.0f, e);
  s.to_float = ldexp(1.0, -e);
  return s;
}

__device__ __forceinline__ long long to_fixed(float v, float scale) {
  return __float2ll_rn(v * scale);     // v * scale is exact (power of two) and below 2^36 in magnitude
}

__device__ __forceinline__ void fixed_add(long long* acc, long long q) {
  if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(acc), static_cast<unsigned long long>(q));
}

// lane l receives lane l-1's value (0 in lane 0): two DPP wave_shr:1 moves.  All lanes of the wave must be active.
__device__ __forceinline__ long long fixed_from_left_lane(long long q) {
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(q), 0x138, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(q >> 32), 0x138, 0xF, 0xF, true);
  return (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo);
}
__device__ __forceinline__ int int_from_left_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ int int_from_right_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }

__device__ __forceinline__ float from_fixed(long long q, const ScatterScale& s) {
  if (s.to_fixed != s.to_fixed) return s.to_fixed;
  return static_cast<float>(static_cast<double>(q) * s.to_float);
}

__host__ __device__ __forceinline__ long long* scatter_acc(void* ws) {
  return reinterpret_cast<long long*>(static_cast<char*>(ws) + SCATTER_HEADER_BYTES);
}

// host side (ops_scatter.hip)
long scatter_ws_bytes(long n);
// zero-fills the n accumulators and sets the bound to max |amax_of[0 .. amax_n)| (the gradient about to be scattered:
// the taps' weights are <= 1, so it bounds every contribution)
int scatter_begin(void* ws, long n, const float* amax_of, long amax_n, hipStream_t st);
// max |amax_of[0 .. amax_n)| as a bit pattern into a zeroed header word (what scatter_begin does after its zero-fill)
int scatter_amax_into(unsigned* header, const float* amax_of, long amax_n, hipStream_t st);
// zero-fill only: a kernel of the caller writes the bound (an analytic one) into word 0 before the scatter
int scatter_begin_bound(void* ws, long n, hipStream_t st);
// zero-fill and a constant bound (e.g. 1 for a scatter of bilinear weights)
int scatter_begin_const(void* ws, long n, float bound, hipStream_t st);
// out[i] = float(acc[i] / scale)
int scatter_finish(const void* ws, float* out, long n, hipStream_t st);
// the same for a slice of the accumulators (header = the workspace's first word)
int scatter_finish_at(const void* header, const long long* acc, float* out, long n, hipStream_t st);

}  // namespace dfe
