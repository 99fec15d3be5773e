// Host side and the two helper kernels of the order-independent scatter-add (dfe_scatter.h).
#include "dfe_internal.h"
#include "dfe_scatter.h"

namespace dfe {

// header word 0 = max |v[i]| as a bit pattern (NaN patterns are larger than Inf's: a NaN input wins the max).
// max is associative and commutative: the unsigned atomicMax is order-independent.
__global__ void __launch_bounds__(256) k_scatter_amax(const float* __restrict__ v, long n, unsigned* __restrict__ header) {
  unsigned m = 0u;
  const long stride = static_cast<long>(gridDim.x) * 256 * 4;
  const bool vec = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  for (long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (vec && i + 3 < n) {
      const uint4 q = *reinterpret_cast<const uint4*>(v + i);
      m = max(max(m, q.x & 0x7fffffffu), max(q.y & 0x7fffffffu, max(q.z & 0x7fffffffu, q.w & 0x7fffffffu)));
    } else {
      for (long j = i; j < n && j < i + 4; ++j) m = max(m, static_cast<unsigned>(__float_as_int(v[j])) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), o)));
  __shared__ unsigned wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    // one atomic per block, and none when the word already holds at least this much (a stale read can only be smaller)
    if (m > __hip_atomic_load(header, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(header, m);
  }
}

__global__ void __launch_bounds__(256) k_scatter_to_float(const unsigned* __restrict__ header, const long long* __restrict__ acc,
                                                          float* __restrict__ out, long n) {
  const ScatterScale sc = scatter_scale(*header);
  const long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 2;
  if (i + 1 < n && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (reinterpret_cast<uintptr_t>(acc) & 15) == 0) {
    const longlong2 q = *reinterpret_cast<const longlong2*>(acc + i);
    *reinterpret_cast<float2*>(out + i) = make_float2(from_fixed(q.x, sc), from_fixed(q.y, sc));
  } else {
    if (i < n) out[i] = from_fixed(acc[i], sc);
    if (i + 1 < n) out[i + 1] = from_fixed(acc[i + 1], sc);
  }
}

long scatter_ws_bytes(long n) { return n > 0 ? SCATTER_HEADER_BYTES + 8 * n : 0; }

int scatter_begin_bound(void* ws, long n, hipStream_t st) {
  if (!ws || n <= 0) return DFE_ERR_NULL;
  if (reinterpret_cast<uintptr_t>(ws) & 15) return DFE_ERR_DIMS;
  if (hipMemsetAsync(ws, 0, static_cast<size_t>(scatter_ws_bytes(n)), st) != hipSuccess) return DFE_ERR_LAUNCH;
  return DFE_OK;
}

int scatter_amax_into(unsigned* header, const float* amax_of, long amax_n, hipStream_t st) {
  if (!header || !amax_of || amax_n <= 0) return DFE_ERR_NULL;
  const long blocks = (amax_n + 256 * 4 * 4 - 1) / (256 * 4 * 4);      // >= 4 float4 per thread
  k_scatter_amax<<<static_cast<unsigned>(blocks > 1024 ? 1024 : blocks), 256, 0, st>>>(amax_of, amax_n, header);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}

int scatter_begin(void* ws, long n, const float* amax_of, long amax_n, hipStream_t st) {
  const int rc = scatter_begin_bound(ws, n, st);
  if (rc != DFE_OK) return rc;
  return scatter_amax_into(static_cast<unsigned*>(ws), amax_of, amax_n, st);
}

__global__ void k_scatter_set_bound(unsigned* header, float bound) { *header = static_cast<unsigned>(__float_as_int(bound)); }

int scatter_begin_const(void* ws, long n, float bound, hipStream_t st) {
  const int rc = scatter_begin_bound(ws, n, st);
  if (rc != DFE_OK) return rc;
  k_scatter_set_bound<<<1, 1, 0, st>>>(static_cast<unsigned*>(ws), bound);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}

int scatter_finish_at(const void* header, const long long* acc, float* out, long n, hipStream_t st) {
  if (!header || !acc || !out || n <= 0) return DFE_ERR_NULL;
  k_scatter_to_float<<<static_cast<unsigned>((n + 511) / 512), 256, 0, st>>>(static_cast<const unsigned*>(header), acc, out, n);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}

int scatter_finish(const void* ws, float* out, long n, hipStream_t st) {
  if (!ws) return DFE_ERR_NULL;
  return scatter_finish_at(ws, reinterpret_cast<const long long*>(static_cast<const char*>(ws) + SCATTER_HEADER_BYTES), out, n, st);
}

}  // namespace dfe

extern "C" long dfe_scatter_ws_bytes(long n) { return dfe::scatter_ws_bytes(n); }
