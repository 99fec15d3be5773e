// Micro-benchmark of the Winograd kernels' inner loop on gfx950: where do the fp32 matrix pipe's idle cycles come from?
// One block = 4 waves; per step a wave reads its A operands (U: 8 x ds_read_b128) and a raw 4x4 patch (4 rows) from LDS,
// transforms the patch (32 adds) and issues 32 v_mfma_f32_16x16x4_f32 on 32 accumulators (the loop of k_wino_fwd16 /
// k_wino_wgrad2 without global memory).  Variants (bit mask): 1 = no patch read / transform (B operand constant),
// 2 = no U reads (A operand constant), 4 = no MFMAs, 8 = 64 accumulators instead of 128 (NH = 1), 16 = 32x32x2 MFMAs.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize mfma_loop.hip -o mfma_loop && ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR, int OCC>
__global__ void __launch_bounds__(256, OCC) k_loop(float* __restrict__ out, int steps) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 12288; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  constexpr int NH = (VAR & 8) ? 1 : 2;
  f32x4 acc[NH][16];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[h][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ub = lds + (kq * 32 + n) * 20;          // U slab: [4 ch][32 k][20]
  const float* xb = lds + 4096 + kq * 480 + (wv * 2) * 24 + 2 * n + 3;   // raw rows
  float cu[16], cv[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) { cu[s] = 0.5f + s; cv[s] = 0.25f * s + lane; }
  for (int st = 0; st < steps; ++st) {
    float v[16];
    if (!(VAR & 1)) {
      float d[16], t[16];
      const float* p = xb + (st & 1) * 1920;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 mid = *reinterpret_cast<const f32x2*>(p + r * 24 + 1);
        d[4 * r] = p[r * 24]; d[4 * r + 1] = mid[0]; d[4 * r + 2] = mid[1]; d[4 * r + 3] = p[r * 24 + 3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { t[j] = d[j] - d[8 + j]; t[4 + j] = d[4 + j] + d[8 + j]; t[8 + j] = d[8 + j] - d[4 + j]; t[12 + j] = d[4 + j] - d[12 + j]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i * 4] = t[i * 4] - t[i * 4 + 2]; v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2]; v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1]; v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3]; }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) v[s] = cv[s];
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float u[16];
      if (!(VAR & 2)) {
        const float* up = ub + (st & 1) * 2560 + h * 320;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 w4 = *reinterpret_cast<const f32x4*>(up + 4 * q); u[4 * q] = w4[0]; u[4 * q + 1] = w4[1]; u[4 * q + 2] = w4[2]; u[4 * q + 3] = w4[3]; }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) u[s] = cu[s];
      }
      if (!(VAR & 4)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[h][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[s], v[s], acc[h][s], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[h][s][0] += u[s] * v[s];
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int s = 0; s < 16; ++s) sum += acc[h][s][0] + acc[h][s][1] + acc[h][s][2] + acc[h][s][3];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int VAR, int OCC>
void run(const char* name, int blocks, int steps, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<VAR, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  for (int i = 0; i < 2; ++i) k_loop<VAR, OCC><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k_loop<VAR, OCC><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const int nh = (VAR & 8) ? 1 : 2;
  const double mfma = double(blocks) * 4 * steps * 16 * nh;                // MFMAs in the launch
  const double ideal_us = mfma * 32.0 / 1024.0 / 2400.0;                  // 32 cycles each, 1024 SIMDs, 2.4 GHz
  printf("%-44s blocks %4d occ %d: %8.1f us  ideal %7.1f us  pipe %5.1f %%\n", name, blocks, OCC, ms * 1e3, ideal_us, 100.0 * ideal_us / (ms * 1e3));
}


// software-pipelined variant: the next step's operands are read while this step's MFMAs issue (MODE 1: compiler-scheduled,
// 2: sched_barrier between the prefetch block and the MFMAs, 3: sched_group_barrier interleave)
template <int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) k_pipe(float* __restrict__ out, int steps) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 12288; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[h][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ub = lds + (kq * 32 + n) * 20;
  const float* xb = lds + 4096 + kq * 480 + (wv * 2) * 24 + 2 * n + 3;
  int opq = 0;       // an opaque zero, refreshed every iteration: the reads cannot be hoisted out of the loop
  auto rd_u = [&](int st, float (&u)[2][16]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* up = ub + (st & 1) * 2560 + h * 320 + opq;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const f32x4 w4 = *reinterpret_cast<const f32x4*>(up + 4 * q); u[h][4 * q] = w4[0]; u[h][4 * q + 1] = w4[1]; u[h][4 * q + 2] = w4[2]; u[h][4 * q + 3] = w4[3]; }
    }
  };
  auto rd_d = [&](int st, float (&d)[16]) {
    const float* p = xb + (st & 1) * 1920 + opq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 mid = *reinterpret_cast<const f32x2*>(p + r * 24 + 1);
      d[4 * r] = p[r * 24]; d[4 * r + 1] = mid[0]; d[4 * r + 2] = mid[1]; d[4 * r + 3] = p[r * 24 + 3];
    }
  };
  auto xf = [&](const float (&d)[16], float (&v)[16]) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = d[j] - d[8 + j]; t[4 + j] = d[4 + j] + d[8 + j]; t[8 + j] = d[8 + j] - d[4 + j]; t[12 + j] = d[4 + j] - d[12 + j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i * 4] = t[i * 4] - t[i * 4 + 2]; v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2]; v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1]; v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3]; }
  };
  float ua[2][16], ub2[2][16], da[16], db[16], va[16], vb[16];
  rd_u(0, ua); rd_d(0, da); xf(da, va);
  for (int st = 0; st < steps; st += 2) {
    // step st: operands (ua, va); prefetch (ub2, db) for st + 1
    asm volatile("" : "+s"(opq));
    rd_u(st + 1, ub2); rd_d(st + 1, db);
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[h][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[h][s], va[s], acc[h][s], 0, 0, 0);
    xf(db, vb);
    if (MODE == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+s"(opq));
    rd_u(st + 2, ua); rd_d(st + 2, da);
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[h][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub2[h][s], vb[s], acc[h][s], 0, 0, 0);
    xf(da, va);
    if (MODE == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
  }
  float sum = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 16; ++s) sum += acc[h][s][0] + acc[h][s][1] + acc[h][s][2] + acc[h][s][3];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE, int OCC>
void runp(const char* name, int blocks, int steps, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe<MODE, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  for (int i = 0; i < 2; ++i) k_pipe<MODE, OCC><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k_pipe<MODE, OCC><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double mfma = double(blocks) * 4 * steps * 32;
  const double ideal_us = mfma * 32.0 / 1024.0 / 2400.0;
  printf("%-44s blocks %4d occ %d: %8.1f us  ideal %7.1f us  pipe %5.1f %%\n", name, blocks, OCC, ms * 1e3, ideal_us, 100.0 * ideal_us / (ms * 1e3));
}

// 32x32x2 variant (weight-gradient shape): a lane owns (channel = lane & 31, tile = lane >> 5); per step of 2 tiles it reads a raw 2x2
// gy tile and a raw 4x4 x patch from LDS, transforms both (12 + 32 adds) and issues 16 v_mfma_f32_32x32x2_f32 on 16 accumulators of
// 16 registers (32 x 32 channels x 16 positions = 256 accumulator registers, one wave per SIMD).  PIPE: operands read a step ahead.
template <int PIPE>
__global__ void __launch_bounds__(256, 1) k_loop32(float* __restrict__ out, int steps) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ch = lane & 31, tq = lane >> 5;
  for (int i = tid; i < 12288; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[s][j] = 0.f;
  const float* gb = lds + ((wv & 1) * 32 + ch) * 56 + 2 * tq;            // gy: [64 ch][2 rows][24] + pad
  const float* xb = lds + 4096 + ((wv >> 1) * 32 + ch) * 136 + 2 * tq + 3;   // x: [64 ch][4 rows][32] + pad
  int opq = 0;
  auto rd = [&](int st, float (&m)[16], float (&v)[16]) {
    const float* p = gb + (st % 6) * 4 + opq;
    const f32x2 r0 = *reinterpret_cast<const f32x2*>(p), r1 = *reinterpret_cast<const f32x2*>(p + 24);
    const float a0[4] = {r0[0], r0[0] + r1[0], r0[0] - r1[0], r1[0]};
    const float a1[4] = {r0[1], r0[1] + r1[1], r0[1] - r1[1], r1[1]};
#pragma unroll
    for (int q = 0; q < 4; ++q) { m[4 * q] = a0[q]; m[4 * q + 1] = a0[q] + a1[q]; m[4 * q + 2] = a0[q] - a1[q]; m[4 * q + 3] = a1[q]; }
    const float* px = xb + (st % 6) * 4 + opq;
    float d[16], t[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 mid = *reinterpret_cast<const f32x2*>(px + r * 32 + 1);
      d[4 * r] = px[r * 32]; d[4 * r + 1] = mid[0]; d[4 * r + 2] = mid[1]; d[4 * r + 3] = px[r * 32 + 3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = d[j] - d[8 + j]; t[4 + j] = d[4 + j] + d[8 + j]; t[8 + j] = d[8 + j] - d[4 + j]; t[12 + j] = d[4 + j] - d[12 + j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i * 4] = t[i * 4] - t[i * 4 + 2]; v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2]; v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1]; v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3]; }
  };
  float ma[16], va[16], mb[16], vb[16];
  rd(0, ma, va);
  for (int st = 0; st < steps; st += 2) {
    asm volatile("" : "+s"(opq));
    if (PIPE) rd(st + 1, mb, vb);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[s], va[s], acc[s], 0, 0, 0);
    if (!PIPE) rd(st + 1, mb, vb);
    asm volatile("" : "+s"(opq));
    if (PIPE) rd(st + 2, ma, va);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(mb[s], vb[s], acc[s], 0, 0, 0);
    if (!PIPE) rd(st + 2, ma, va);
  }
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += acc[s][j];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int PIPE>
void run32(const char* name, int blocks, int steps, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop32<PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  for (int i = 0; i < 2; ++i) k_loop32<PIPE><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k_loop32<PIPE><<<blocks, 256, 49152>>>(out, steps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double mfma = double(blocks) * 4 * steps * 16;                    // 64-cycle MFMAs
  const double ideal_us = mfma * 64.0 / 1024.0 / 2400.0;
  printf("%-44s blocks %4d occ 1: %8.1f us  ideal %7.1f us  pipe %5.1f %%\n", name, blocks, ms * 1e3, ideal_us, 100.0 * ideal_us / (ms * 1e3));
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int steps = 2000;
  run<0, 2>("full loop (U + patch + transform + MFMA)", 512, steps, out);
  run<0, 1>("full loop, one wave per SIMD", 256, steps, out);
  run<1, 2>("no patch / transform", 512, steps, out);
  run<2, 2>("no U reads", 512, steps, out);
  run<3, 2>("MFMA only (operands in registers)", 512, steps, out);
  run<3, 1>("MFMA only, one wave per SIMD", 256, steps, out);
  run<4, 2>("no MFMA (reads + transform + 32 FMAs)", 512, steps, out);
  run<8, 2>("NH = 1 (64 accumulators), 2 waves", 512, steps, out);
  run<8, 4>("NH = 1, 4 waves per SIMD", 1024, steps, out);
  run<9, 4>("NH = 1, no patch, 4 waves", 1024, steps, out);
  runp<1, 2>("pipelined, compiler-scheduled, 2 waves", 512, steps, out);
  runp<2, 2>("pipelined, sched_barrier, 2 waves", 512, steps, out);
  runp<3, 2>("pipelined, sched_group_barrier, 2 waves", 512, steps, out);
  runp<1, 1>("pipelined, compiler-scheduled, 1 wave", 256, steps, out);
  runp<2, 1>("pipelined, sched_barrier, 1 wave", 256, steps, out);
  runp<3, 1>("pipelined, sched_group_barrier, 1 wave", 256, steps, out);
  run32<0>("32x32x2, 256 accumulators, 1 wave", 256, steps, out);
  run32<1>("32x32x2, 256 accumulators, pipelined, 1 wave", 256, steps, out);
  return 0;
}
