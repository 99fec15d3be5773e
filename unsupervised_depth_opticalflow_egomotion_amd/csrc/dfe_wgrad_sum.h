// The fixed-order sum of a weight gradient's split partials (ops_wino_wgrad.hip, ops_sconv.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dfe {

// gw[i] = sum over the splits of part[s][i]: SG groups of consecutive splits are summed side by side (each in split order) and
// the SG group sums are added in group order -- a fixed association, whatever the timing.  32 groups when there are many splits
// (64 x 64 filters: up to 768 splits of 36 k outputs -- 8 groups left every thread a chain of ~100 dependent loads).
template <int SG>
__global__ void __launch_bounds__(32 * SG) k_wgrad_sum(const float* __restrict__ part, float* __restrict__ gw, int S, long n) {
  __shared__ float sm[SG][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long idx = static_cast<long>(blockIdx.x) * 32 + o;
  const int per = (S + SG - 1) / SG, s0 = g * per, s1 = min(S, s0 + per);
  float s = 0.0f;
  if (idx < n)
    for (int k = s0; k < s1; ++k) s += part[k * n + idx];
  sm[g][o] = s;
  __syncthreads();
  if (g == 0 && idx < n) {
    float t = sm[0][o];
#pragma unroll
    for (int j = 1; j < SG; ++j) t += sm[j][o];
    gw[idx] = t;
  }
}

}  // namespace dfe
