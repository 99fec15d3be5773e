// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of this build's kernels.
// MI355X_MICROARCH.md calibrates only 16-B-per-lane streams (FETCH_SIZE reads exactly half of the bytes, WRITE_SIZE all
// of them); the rolling stencil kernels load one dword per lane in 256-B row segments and the pointwise kernels gather
// 8-byte pairs.  Every kernel below moves a KNOWN number of bytes of a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o t --output-format csv -- ./fetch_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE ...                                   (tools/fetch_calib.sh does both)
// and tools/fetch_calib.py divides the counters by the known byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// coalesced grid-stride read, VEC floats per lane per access
template <int VEC>
__global__ void __launch_bounds__(256) k_read_stream(const float* __restrict__ in, float* __restrict__ out, size_t nfloat) {
  size_t i = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x * VEC;
  float acc = 0.0f;
  for (; i + VEC <= nfloat; i += stride) {
    if (VEC == 1) acc += in[i];
    else if (VEC == 2) { const float2 v = *reinterpret_cast<const float2*>(in + i); acc += v.x + v.y; }
    else { const float4 v = *reinterpret_cast<const float4*>(in + i); acc += (v.x + v.y) + (v.z + v.w); }
  }
  if (acc == 1234.5f) out[0] = acc;
}

// the rolling stencil kernels' shape: planes of H x W floats; one wave owns a strip of 62 columns (64 lanes, 2 of them a
// halo shared with the next strip) and marches down ROWS rows (+2 halo rows), one dword per lane per row
__global__ void __launch_bounds__(64) k_read_strips(const float* __restrict__ in, float* __restrict__ out, int H, int W, int strips, int rows) {
  const int strip = blockIdx.x % strips, rb = blockIdx.x / strips;
  const float* pl = in + static_cast<size_t>(blockIdx.y) * H * W;
  const int x = min(strip * 62 + static_cast<int>(threadIdx.x), W - 1);
  float acc = 0.0f;
  for (int y = rb * rows - 1; y <= rb * rows + rows; ++y) {
    const int yy = min(max(y, 0), H - 1);
    acc += pl[static_cast<size_t>(yy) * W + x];
  }
  if (acc == 1234.5f) out[0] = acc;
}

// 8-byte pair gathers around the lane's own pixel (the pointwise kernels' footprint rows), every pixel once per plane
__global__ void __launch_bounds__(256) k_read_pairs(const float* __restrict__ in, float* __restrict__ out, int H, int W, int shift) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const float* pl = in + static_cast<size_t>(blockIdx.y) * H * W;
  const int y = p / W, x = p - y * W;
  const int xs = min(max(x + shift, 0), W - 2), ya = min(max(y + shift, 0), H - 1), yb = min(ya + 1, H - 1);
  struct __attribute__((packed, aligned(4))) P2 { float a, b; };
  const P2 r0 = *reinterpret_cast<const P2*>(pl + static_cast<size_t>(ya) * W + xs);
  const P2 r1 = *reinterpret_cast<const P2*>(pl + static_cast<size_t>(yb) * W + xs);
  if ((r0.a + r0.b) + (r1.a + r1.b) == 1234.5f) out[0] = r0.a;
}

template <int VEC>
__global__ void __launch_bounds__(256) k_write_stream(float* __restrict__ outb, size_t nfloat) {
  size_t i = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x * VEC;
  for (; i + VEC <= nfloat; i += stride) {
    if (VEC == 1) outb[i] = 1.0f;
    else { *reinterpret_cast<float4*>(outb + i) = float4{1.0f, 2.0f, 3.0f, 4.0f}; }
  }
}

// one byte per lane (the mask pack)
__global__ void __launch_bounds__(256) k_write_bytes(unsigned char* __restrict__ outb, size_t n) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) outb[i] = static_cast<unsigned char>(i);
}

int main() {
  const size_t nfloat = 256u << 20;              // 1 GiB
  const int H = 256, W = 832, planes = static_cast<int>(nfloat / (static_cast<size_t>(H) * W));   // 1260 planes
  float *buf, *out;
  CHECK(hipMalloc(&buf, nfloat * 4));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0, nfloat * 4));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 16;
  k_read_stream<4><<<grid, 256>>>(buf, out, nfloat);
  k_read_stream<2><<<grid, 256>>>(buf, out, nfloat);
  k_read_stream<1><<<grid, 256>>>(buf, out, nfloat);
  const int strips = (W + 61) / 62, rows = 8;
  k_read_strips<<<dim3(strips * (H / rows), planes), 64>>>(buf, out, H, W, strips, rows);
  k_read_pairs<<<dim3((H * W + 255) / 256, planes), 256>>>(buf, out, H, W, 0);
  k_read_pairs<<<dim3((H * W + 255) / 256, planes), 256>>>(buf, out, H, W, 3);
  k_write_stream<4><<<grid, 256>>>(buf, nfloat);
  k_write_stream<1><<<grid, 256>>>(buf, nfloat);
  k_write_bytes<<<grid, 256>>>(reinterpret_cast<unsigned char*>(buf), nfloat);   // 256 MiB of bytes
  CHECK(hipDeviceSynchronize());
  printf("bytes: stream=%zu strips_algorithmic=%zu pairs_algorithmic=%zu write_stream=%zu write_bytes=%zu\n", nfloat * 4,
         static_cast<size_t>(planes) * H * W * 4, static_cast<size_t>(planes) * H * W * 4, nfloat * 4, nfloat);
  return 0;
}
