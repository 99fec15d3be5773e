// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate as a function of the accumulator tile (NA x NB tiles = NA A registers, NB B
// registers, NA NB accumulators, operands in registers): what a wave tile must look like to keep the fp32 matrix pipe busy.
//   hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int NB, int OCC>
__global__ void __launch_bounds__(256, OCC) k_chain(float* __restrict__ out, int steps) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NA][NB];
  float a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = 0.5f + i + lane;
#pragma unroll
  for (int j = 0; j < NB; ++j) b[j] = 0.25f * j + lane;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int st = 0; st < steps; ++st) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NA, int NB, int OCC>
void run(float* out) {
  const int blocks = 256 * OCC, steps = 4096 / (NA * NB) * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_chain<NA, NB, OCC><<<blocks, 256>>>(out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_chain<NA, NB, OCC><<<blocks, 256>>>(out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = double(steps) * NA * NB * OCC;          // per SIMD
  const double ideal_us = mfma * 32.0 / 2400.0;
  printf("NA %d x NB %d (%2d accumulators), %d wave(s) per SIMD: %8.1f us  ideal %8.1f us  pipe %5.1f %%  (%.1f cycles per MFMA at 2.4 GHz)\n", NA, NB, NA * NB, OCC,
         ms * 1e3, ideal_us, 100.0 * ideal_us / (ms * 1e3), ms * 1e3 * 2400.0 / mfma);
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * sizeof(float));
  run<1, 4, 1>(out); run<1, 4, 2>(out);
  run<1, 9, 1>(out); run<1, 9, 2>(out);
  run<2, 5, 1>(out); run<2, 5, 2>(out);
  run<1, 16, 1>(out); run<1, 16, 2>(out);
  run<2, 8, 1>(out); run<2, 8, 2>(out);
  run<4, 4, 1>(out); run<4, 4, 2>(out);
  run<2, 9, 1>(out); run<2, 9, 2>(out);
  run<4, 8, 1>(out); run<4, 8, 2>(out);
  return 0;
}
