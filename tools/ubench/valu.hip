// VALU issue-rate microbenchmark: v_fma_f32 vs v_pk_fma_f32 vs mixed, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  float2v pa = {a, a}, pb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
      }
    } else if (MODE == 2) {   // v_mul + v_add (non-fused), scalar
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                     "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 3) {   // packed mul/add
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                     "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
      }
    } else if (MODE == 4) {   // v_cndmask / cmp mix
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                     "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
      }
    } else if (MODE == 5) {   // transcendental: rcp
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> float run(int blocks_per_cu, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  k<MODE><<<grid, 256>>>(d, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  const int iters = 20000;
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul/v_add", "v_pk_mul/pk_add", "v_cmp+v_cndmask", "v_rcp_f32"};
  for (int bpc : {1, 2, 4, 8}) {   // blocks of 256 threads per CU = waves per SIMD
    float ms[6] = {run<0>(bpc, iters, d), run<1>(bpc, iters, d), run<2>(bpc, iters, d), run<3>(bpc, iters, d), run<4>(bpc, iters, d), run<5>(bpc, iters, d)};
    for (int m = 0; m < 6; ++m) {
      double ninstr = (double)iters * 64;          // wave-instructions per wave
      double cyc_per_instr = ms[m] * 1e-3 * 2.4e9 / (ninstr * bpc);   // per SIMD, assuming 2.4 GHz
      printf("waves/SIMD %d  %-16s %8.3f ms  -> %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", bpc, names[m], ms[m], cyc_per_instr);
    }
  }
  return 0;
}
