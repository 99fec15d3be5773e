// Micro-benchmark: what takes the fp32 matrix pipe from the 91 % of a bare v_mfma_f32_16x16x4_f32 chain (mfma_chain.hip, 9 accumulators)
// to the ~50 % the LDS-fed convolution kernels reach?  The inner loop of k_sconv_wgrad<1, 9, 4, 1> rebuilt piece by piece:
//   bit 0: per-step pointer arithmetic (NB + 1 adds and shifts)      bit 1: operands read from LDS (ds_read_b32, one step ahead)
//   bit 2: chunks of 13 steps with a row prologue                     bit 3: a barrier per chunk
//   bit 4: staging traffic: 12 x 16-byte global loads per chunk in flight under the MFMAs, stored to LDS behind a second barrier
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma_feed.hip -o mfma_feed && ./mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 9;

template <int VAR>
__global__ void __launch_bounds__(256, 2) k_feed(const float* __restrict__ src, float* __restrict__ out, int chunks, int steps_per_chunk) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4, wv = tid >> 6;
  for (int e = tid; e < 9600; e += 256) lds[e] = 0.001f * (e % 97);
  __syncthreads();
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int boff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) boff[j] = i * 354 + (j / 3) * 108 + (j % 3 & 1) * 54 + (j % 3 >> 1) + kq;
  const int aoff = 5700 + (wv * 16 + i) * 58 + kq;
  float a0 = 0.5f + lane, b0[NB], a1 = 0.f, b1[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { b0[j] = 0.25f * j + lane; b1[j] = b0[j]; }
  f32x4 pre[12];
  const float* gsrc = src + (static_cast<long>(blockIdx.x) * 256 + tid) * 4;
  for (int c = 0; c < chunks; ++c) {
    if (VAR & 16) {
#pragma unroll
      for (int q = 0; q < 12; ++q) pre[q] = *reinterpret_cast<const f32x4*>(gsrc + (static_cast<long>(c % 8) * 12 + q) * 524288);
    }
    int pa = aoff, pb[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) pb[j] = boff[j];
    if (VAR & 2) {
      a0 = lds[pa];
#pragma unroll
      for (int j = 0; j < NB; ++j) b0[j] = lds[pb[j]];
    }
    int s = 0;
    for (; s + 2 <= steps_per_chunk; s += 2) {
      if (VAR & 2) {
        a1 = lds[pa + 4];
#pragma unroll
        for (int j = 0; j < NB; ++j) b1[j] = lds[pb[j] + 4];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR & 2) {
        a0 = lds[pa + 8];
#pragma unroll
        for (int j = 0; j < NB; ++j) b0[j] = lds[pb[j] + 8];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32((VAR & 2) ? a1 : a0, (VAR & 2) ? b1[j] : b0[j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR & 1) {
        pa += 8;
#pragma unroll
        for (int j = 0; j < NB; ++j) pb[j] += 8;
        if (!(VAR & 2)) {      // keep the arithmetic alive without LDS reads
          a0 += __int_as_float(pa & 1);
#pragma unroll
          for (int j = 0; j < NB; ++j) b0[j] += __int_as_float(pb[j] & 1);
        }
      }
    }
    if (s < steps_per_chunk) {
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[j], acc[j], 0, 0, 0);
    }
    if (VAR & 8) __syncthreads();
    if (VAR & 16) {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        float* d = lds + (q * 256 + tid) * 3 % 9000;
        d[0] = pre[q][0] + pre[q][1]; d[1] = pre[q][2] + pre[q][3];
      }
      __syncthreads();
    }
  }
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NB; ++j) t += acc[j];
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}

template <int VAR>
void run(const char* name, const float* src, float* out, int spc) {
  const int blocks = 512, chunks = 1404 / (spc * NB) * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feed<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  k_feed<VAR><<<blocks, 256, 40000>>>(src, out, chunks, spc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_feed<VAR><<<blocks, 256, 40000>>>(src, out, chunks, spc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma = double(chunks) * spc * NB * 2;          // per SIMD: two waves
  const double ideal_us = mfma * 32.0 / 2400.0;
  printf("%-68s %2d steps/chunk: %7.1f us  ideal %7.1f us  pipe %5.1f %%\n", name, spc, ms * 1e3, ideal_us, 100.0 * ideal_us / (ms * 1e3));
}

int main() {
  float *out, *src;
  (void)hipMalloc(&out, 512 * 256 * sizeof(float));
  (void)hipMalloc(&src, size_t(8) * 12 * 524288 * 4 + (1 << 22));
  (void)hipMemset(src, 0, size_t(8) * 12 * 524288 * 4 + (1 << 22));
  run<0>("bare chain, 9 accumulators, 2 waves per SIMD", src, out, 13);
  run<1>("+ pointer arithmetic per step pair", src, out, 13);
  run<2>("LDS operands a step ahead (no pointer arithmetic)", src, out, 13);
  run<3>("LDS operands + pointer arithmetic", src, out, 13);
  run<3>("  ... 26-step chunks", src, out, 26);
  run<3>("  ... 104-step chunks", src, out, 104);
  run<11>("+ a barrier per chunk", src, out, 13);
  run<11>("  ... 26-step chunks", src, out, 26);
  run<27>("+ staging: 12 global 16-byte loads under the MFMAs, LDS stores, 2nd barrier", src, out, 13);
  run<27>("  ... 26-step chunks", src, out, 26);
  run<27>("  ... 104-step chunks", src, out, 104);
  return 0;
}
