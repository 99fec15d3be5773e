// Is VALU time + VMEM time additive or overlapped on gfx950?  Loop of M coalesced loads (L2-resident) and V fmas.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int M, int V, int WIDTH>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* out, int iters, int nfl) {
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned off = (lane * WIDTH) % nfl;
  for (int it = 0; it < iters; ++it) {
    float v[M * WIDTH];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const unsigned o = (off + m * 4096 * WIDTH) % nfl;
      if (WIDTH == 1) v[m] = in[o];
      else if (WIDTH == 2) { float2 t = *reinterpret_cast<const float2*>(in + o); v[2 * m] = t.x; v[2 * m + 1] = t.y; }
      else { float4 t = *reinterpret_cast<const float4*>(in + o); v[4 * m] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w; }
    }
    off = (off + 64 * 4096 * WIDTH + 17 * WIDTH) % nfl;
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j & 7] = __fmaf_rn(acc[j & 7], 1.0001f, v[j % (M * WIDTH)]);
  }
  out[lane] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
}
template <int M, int V, int WIDTH> void run(const float* in, float* out, int nfl) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8, iters = 2000;
  k<M, V, WIDTH><<<grid, 256>>>(in, out, 10, nfl);
  (void)hipEventRecord(e0);
  k<M, V, WIDTH><<<grid, 256>>>(in, out, iters, nfl);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per CU: 32 waves x iters iterations; cycles per (wave-iteration) per CU at 2.1 GHz
  const double cyc = ms * 1e-3 * 2.1e9 / (32.0 * iters);
  printf("M=%2d loads(x%d) V=%3d fmas : %7.3f ms  -> %6.1f CU-cycles per wave-iteration\n", M, WIDTH, V, ms, cyc);
}
int main() {
  const int nfl = 8 << 20;   // 32 MB: L2/MALL resident
  float *in, *out; (void)hipMalloc(&in, nfl * 4 + 64); (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMemset(in, 0, nfl * 4);
  run<8, 8, 1>(in, out, nfl); run<8, 64, 1>(in, out, nfl); run<8, 128, 1>(in, out, nfl); run<8, 256, 1>(in, out, nfl); run<8, 512, 1>(in, out, nfl);
  run<16, 16, 1>(in, out, nfl); run<16, 256, 1>(in, out, nfl); run<16, 512, 1>(in, out, nfl);
  run<8, 8, 2>(in, out, nfl); run<8, 256, 2>(in, out, nfl);
  run<8, 8, 4>(in, out, nfl); run<8, 256, 4>(in, out, nfl);
  run<2, 256, 1>(in, out, nfl); run<2, 512, 1>(in, out, nfl);
  return 0;
}
