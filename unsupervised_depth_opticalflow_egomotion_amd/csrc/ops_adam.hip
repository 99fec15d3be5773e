// Adam update of all parameters in ONE launch (reference train.py:85-87: torch.optim.Adam(model.parameters(), lr);
// default betas / eps, no weight decay, no amsgrad).  The optimiser step is the serial tail of the training step:
// nothing else runs while it does.  ATen's fused multi-tensor Adam passes its pointer tables as kernel arguments, which
// limits one launch to 36 tensors and 320 blocks of 64 Ki elements: six launches of ~48 us at 21.6 M parameters in ~250
// tensors (2 TB/s of the 7 fp32 streams it moves).  Here the tables live in device memory, one block owns
// DFE_ADAM_CHUNK elements of one tensor, and one launch covers everything.  Bound: HBM (4 reads + 3 writes per element).
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p = p - (lr / c1) m / (sqrt(v) / sqrt(c2) + eps)
// (sqrtf here is the compiler's correctly rounded one: oclc_correctly_rounded_sqrt)
// with the bias corrections c1 = 1 - b1^t, c2 = 1 - b2^t of the step count t (the arithmetic of ATen's fused kernel).
#include "dfe_internal.h"
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dfe {

constexpr int ADAM_CHUNK = 4096;       // elements per block: 256 threads x 4 float4

struct AdamRec { float* p; const float* g; float* m; float* v; long long n; };   // 5 x 8 bytes per tensor

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1c, float b2, float b2c, float step_size,
                                      float c2s, float eps) {
  m = m + b1c * (g - m);
  v = b2 * v + b2c * g * g;
  const float denom = sqrtf(v) / c2s + eps;
  p = p - step_size * m / denom;
}

// The step count on the DEVICE (a training step replayed from a hipGraph cannot take its bias corrections as kernel arguments:
// they would stay those of the captured step).  One thread: t = ++count; coef = {lr / (1 - b1^t), sqrt(1 - b2^t)}, formed in double.
__global__ void k_adam_tick(double* __restrict__ count, float* __restrict__ coef, double lr, double b1, double b2) {
  const double t = count[0] + 1.0;
  count[0] = t;
  coef[0] = static_cast<float>(lr / (1.0 - pow(b1, t)));
  coef[1] = static_cast<float>(sqrt(1.0 - pow(b2, t)));
}

// grid: x = block of the block map; blockmap[2k] = tensor, blockmap[2k+1] = chunk.  coef != nullptr: step_size and c2s come from
// device memory (k_adam_tick), the arguments of those names are ignored.
__global__ void __launch_bounds__(256) k_adam_step(const AdamRec* __restrict__ table, const int* __restrict__ blockmap,
                                                   float b1c, float b2, float b2c, float step_size, float c2s, float eps,
                                                   const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; c2s = coef[1]; }
  const int t = blockmap[2 * blockIdx.x], c = blockmap[2 * blockIdx.x + 1];
  const AdamRec r = table[t];
  const long long base = static_cast<long long>(c) * ADAM_CHUNK;
  const long long left = r.n - base;
  const bool vec = ((reinterpret_cast<uintptr_t>(r.p) | reinterpret_cast<uintptr_t>(r.g) | reinterpret_cast<uintptr_t>(r.m) |
                     reinterpret_cast<uintptr_t>(r.v)) & 15) == 0;
  if (vec && left >= ADAM_CHUNK) {
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / (256 * 4); ++k) {
      const long long e = base + (static_cast<long long>(k) * 256 + threadIdx.x) * 4;
      float4 p = *reinterpret_cast<const float4*>(r.p + e), m = *reinterpret_cast<const float4*>(r.m + e);
      float4 v = *reinterpret_cast<const float4*>(r.v + e);
      const float4 g = *reinterpret_cast<const float4*>(r.g + e);
      adam1(p.x, g.x, m.x, v.x, b1c, b2, b2c, step_size, c2s, eps);
      adam1(p.y, g.y, m.y, v.y, b1c, b2, b2c, step_size, c2s, eps);
      adam1(p.z, g.z, m.z, v.z, b1c, b2, b2c, step_size, c2s, eps);
      adam1(p.w, g.w, m.w, v.w, b1c, b2, b2c, step_size, c2s, eps);
      *reinterpret_cast<float4*>(r.p + e) = p; *reinterpret_cast<float4*>(r.m + e) = m; *reinterpret_cast<float4*>(r.v + e) = v;
    }
  } else {
    const long long end = left < ADAM_CHUNK ? r.n : base + ADAM_CHUNK;
    for (long long e = base + threadIdx.x; e < end; e += 256) {
      float p = r.p[e], m = r.m[e], v = r.v[e];
      adam1(p, r.g[e], m, v, b1c, b2, b2c, step_size, c2s, eps);
      r.p[e] = p; r.m[e] = m; r.v[e] = v;
    }
  }
}

}  // namespace dfe

extern "C" int dfe_adam_chunk(void) { return dfe::ADAM_CHUNK; }

extern "C" int dfe_adam_step(const void* table, const int* blockmap, int nblocks, double lr, double beta1, double beta2, double eps,
                             double bias_correction1, double bias_correction2, void* stream) {
  if (!table || !blockmap) return DFE_ERR_NULL;
  if (nblocks <= 0 || !(bias_correction1 > 0.0) || !(bias_correction2 > 0.0)) return DFE_ERR_DIMS;
  // the coefficients are formed in double and rounded once (1 - 0.999f is 4.7e-5 away from 0.001)
  const float step_size = static_cast<float>(lr / bias_correction1);
  const float c2s = static_cast<float>(sqrt(bias_correction2));
  dfe::k_adam_step<<<nblocks, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const dfe::AdamRec*>(table), blockmap,
                                                                        static_cast<float>(1.0 - beta1), static_cast<float>(beta2),
                                                                        static_cast<float>(1.0 - beta2), step_size, c2s, static_cast<float>(eps), nullptr);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}

extern "C" int dfe_adam_step_dev(const void* table, const int* blockmap, int nblocks, double lr, double beta1, double beta2, double eps,
                                 double* step_count, float* coef, void* stream) {
  if (!table || !blockmap || !step_count || !coef) return DFE_ERR_NULL;
  if (nblocks <= 0) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dfe::k_adam_tick<<<1, 1, 0, st>>>(step_count, coef, lr, beta1, beta2);
  if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH;
  dfe::k_adam_step<<<nblocks, 256, 0, st>>>(static_cast<const dfe::AdamRec*>(table), blockmap, static_cast<float>(1.0 - beta1),
                                            static_cast<float>(beta2), static_cast<float>(1.0 - beta2), 0.0f, 1.0f, static_cast<float>(eps), coef);
  return hipGetLastError() == hipSuccess ? DFE_OK : DFE_ERR_LAUNCH;
}
