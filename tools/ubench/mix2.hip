// Issue cost of LDS reads / gathers / partial-wave loads next to VALU work (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: ds_read_b32 x M ; 1: ds_read2_b32 (adjacent) x M ; 2: ds_read_b64 x M ; 3: global 8-byte gather (random-ish, dword aligned) x M
// 4: global dwordx4 with 16 active lanes x M ; 5: global dword store x M ; 6: global dwordx4 store (64 lanes) x M
template <int MODE, int M, int V>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* out, int iters, int nfl) {
  __shared__ float lds[8192];
  const int t = threadIdx.x, lane = blockIdx.x * blockDim.x + t;
  for (int i = t; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned off = (lane * 7) % 4096;
  for (int it = 0; it < iters; ++it) {
    float v[M * 2];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (MODE == 0) { v[2 * m] = lds[(off + m * 67) & 8191]; v[2 * m + 1] = 0; }
      else if (MODE == 1) { const unsigned o = (off + m * 67) & 8190; v[2 * m] = lds[o]; v[2 * m + 1] = lds[o + 1]; }
      else if (MODE == 2) { const float2 q = *reinterpret_cast<const float2*>(&lds[((off + m * 67) & 4095) * 2]); v[2 * m] = q.x; v[2 * m + 1] = q.y; }
      else if (MODE == 3) { const unsigned o = ((lane * 13 + it * 977 + m * 4099) * 3u) % (nfl - 2);
        struct __attribute__((packed, aligned(4))) P { float a, b; }; const P q = *reinterpret_cast<const P*>(in + o); v[2 * m] = q.a; v[2 * m + 1] = q.b; }
      else if (MODE == 4) { v[2 * m] = 0; v[2 * m + 1] = 0; if ((t & 63) < 16) { const float4 q = *reinterpret_cast<const float4*>(in + ((lane * 4 + m * 8192 + it * 64) % nfl)); v[2 * m] = q.x + q.z; v[2 * m + 1] = q.y + q.w; } }
      else if (MODE == 5) { out[(lane + m * 65536 + (it & 7) * 1024) % nfl] = acc[m & 7]; v[2 * m] = 1; v[2 * m + 1] = 1; }
      else { *reinterpret_cast<float4*>(out + ((lane * 4 + m * 262144 + (it & 7) * 1024) % nfl)) = float4{acc[0], acc[1], acc[2], acc[3]}; v[2 * m] = 1; v[2 * m + 1] = 1; }
    }
    off = (off + 131) & 4095;
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j & 7] = __fmaf_rn(acc[j & 7], 1.0001f, v[j % (M * 2)]);
  }
  out[lane] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
}
template <int MODE, int M, int V> void run(const float* in, float* out, int nfl, const char* nm) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8, iters = 2000;
  k<MODE, M, V><<<grid, 256>>>(in, out, 10, nfl);
  (void)hipEventRecord(e0);
  k<MODE, M, V><<<grid, 256>>>(in, out, iters, nfl);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s M=%2d V=%3d : %7.3f ms -> %6.1f CU-cycles per wave-iteration (2.1 GHz)\n", nm, M, V, ms, ms * 1e-3 * 2.1e9 / (32.0 * iters));
}
int main() {
  const int nfl = 8 << 20;
  float *in, *out; (void)hipMalloc(&in, nfl * 4 + 64); (void)hipMalloc(&out, nfl * 4 + 64); (void)hipMemset(in, 0, nfl * 4);
  run<0, 8, 16>(in, out, nfl, "ds_read_b32"); run<0, 8, 256>(in, out, nfl, "ds_read_b32"); run<0, 32, 64>(in, out, nfl, "ds_read_b32");
  run<1, 8, 16>(in, out, nfl, "ds_read2_b32 adjacent"); run<1, 8, 256>(in, out, nfl, "ds_read2_b32 adjacent"); run<1, 32, 64>(in, out, nfl, "ds_read2_b32 adjacent");
  run<2, 8, 16>(in, out, nfl, "ds_read_b64"); run<2, 32, 64>(in, out, nfl, "ds_read_b64");
  run<3, 8, 16>(in, out, nfl, "global 8B gather"); run<3, 8, 256>(in, out, nfl, "global 8B gather"); run<3, 24, 48>(in, out, nfl, "global 8B gather");
  run<4, 8, 16>(in, out, nfl, "global dwordx4, 16 lanes"); run<4, 8, 256>(in, out, nfl, "global dwordx4, 16 lanes");
  run<5, 8, 16>(in, out, nfl, "global store dword"); run<5, 8, 256>(in, out, nfl, "global store dword");
  run<6, 8, 16>(in, out, nfl, "global store dwordx4"); run<6, 8, 256>(in, out, nfl, "global store dwordx4");
  return 0;
}
