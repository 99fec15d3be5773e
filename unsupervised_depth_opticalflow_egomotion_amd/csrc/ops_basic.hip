// Per-operator HIP kernels behind the reference's free functions (drop-in level 3):
//   warp_flow            core/networks/structures/net_utils.py:16-54
//   pose_vec2mat/euler   core/networks/structures/inverse_warp.py:110-187
//   inverse_warp2        core/networks/structures/inverse_warp.py:227-303
//   calculate_rigid_flow core/networks/structures/inverse_warp.py:311-342
//   essential matrix     core/networks/structures/inverse_warp.py:344-364
//   SSIM                 core/networks/pytorch_ssim/ssim.py:4-19
//   corr_naive           core/networks/structures/pwc_tf.py:97-106
//   pyramids             core/networks/model_geometry.py:65-72,91 ; model_flow.py:58-64
// All kernels are HBM/L2-bound stencil or gather work: no MFMA, coalesced row-major
// accesses, LDS only where a tile is re-read (SSIM window, correlation window).
#include "dfe_camera.h"
#include "dfe_scatter.h"
#include <cstdint>
#include <cstdlib>

namespace dfe {

// One thread per camera.  cams[(b*ndir + d)*nscale + s]; pose laid out [B, ndir, 6];
// K [B,3,3]; K_s = K with rows 0-1 divided by downscale[s] (model_geometry.py:92-93).
__global__ void k_prepare_cameras(const float* __restrict__ pose, const float* __restrict__ K,
                                  Camera* __restrict__ cams, int B, int ndir, int nscale, ScaleList downs) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * ndir * nscale) return;
  int s = idx % nscale, d = (idx / nscale) % ndir, b = idx / (nscale * ndir);
  const float* pv = pose + (static_cast<long>(b) * ndir + d) * 6;
  Camera c;
  make_camera(pv, K + b * 9, downs.v[s], c);
  cams[idx] = c;
}

// grad_pose[cam] (6) from the 12 camera sums per (cam, scale): acc = [dL/db(3), dL/dA(9)].
// partials laid out [ncam][nscale][nblk_max][12]; nblk[s] valid rows per scale.  Fixed order,
// double accumulation -> bitwise reproducible.  One thread per camera (b,d).
__global__ void k_pose_finalize(const float* __restrict__ partials, const Camera* __restrict__ cams,
                                float* __restrict__ gpose, int ncam, int nscale, int nblk_max, IntList nblk,
                                int accumulate) {
  int cam = blockIdx.x * blockDim.x + threadIdx.x;
  if (cam >= ncam) return;
  double g[6] = {0, 0, 0, 0, 0, 0};
  for (int s = 0; s < nscale; ++s) {
    double acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = 0.0;
    const float* p = partials + (static_cast<long>(cam) * nscale + s) * nblk_max * 12;
    for (int k = 0; k < nblk.v[s]; ++k)
      for (int i = 0; i < 12; ++i) acc[i] += static_cast<double>(p[k * 12 + i]);
    const Camera& c = cams[cam * nscale + s];
    // dL/dt = K^T dL/db
    for (int j = 0; j < 3; ++j) g[j] += c.K[j] * acc[0] + c.K[3 + j] * acc[1] + c.K[6 + j] * acc[2];
    // dL/dR = K^T dL/dA ; dL/dtheta_k = <dL/dR, dR/dtheta_k>
    double gR[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        gR[i * 3 + j] = c.K[i] * acc[3 + j] + c.K[3 + i] * acc[6 + j] + c.K[6 + i] * acc[9 + j];
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
      for (int i = 0; i < 9; ++i) t += gR[i] * c.dR[k * 9 + i];
      g[3 + k] += t;
    }
  }
  for (int i = 0; i < 6; ++i) {
    float v = static_cast<float>(g[i]);
    gpose[cam * 6 + i] = accumulate ? gpose[cam * 6 + i] + v : v;
  }
}

// pose_vec2mat / essential matrix, forward and backward, one thread per pose vector.
__global__ void k_pose_mats(const float* __restrict__ vec, float* __restrict__ T34, float* __restrict__ E, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = vec + i * 6;
  // the reference's fp32 arithmetic (see k_prepare_cameras): R = (X @ Y) @ Z and E = [t]x @ R as small bmm loops
  float R[9];
  euler_rotation_f32(v[3], v[4], v[5], R);
  if (T34) for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T34[i * 12 + r * 4 + c] = R[r * 3 + c]; T34[i * 12 + r * 4 + 3] = v[r]; }
  if (E) {
    const float S[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    mat3_mul_f32(S, R, E + i * 9);
  }
}

__global__ void k_pose_mats_bwd(const float* __restrict__ vec, const float* __restrict__ gT34,
                                const float* __restrict__ gE, float* __restrict__ gvec, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = vec + i * 6;
  double X[9], Y[9], Z[9], dX[9], dY[9], dZ[9], XY[9], R[9], T1[9], dRk[3][9];
  euler_mats(v[3], v[4], v[5], X, Y, Z, dX, dY, dZ);
  mat3_mul(X, Y, XY); mat3_mul(XY, Z, R);
  mat3_mul(dX, Y, T1); mat3_mul(T1, Z, dRk[0]);
  mat3_mul(X, dY, T1); mat3_mul(T1, Z, dRk[1]);
  mat3_mul(XY, dZ, dRk[2]);
  double gR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gt[3] = {0, 0, 0};
  if (gT34) {
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) gR[r * 3 + c] += gT34[i * 12 + r * 4 + c]; gt[r] += gT34[i * 12 + r * 4 + 3]; }
  }
  if (gE) {
    // E = S R : dL/dR += S^T gE ; dL/dS = gE R^T
    double S[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    double gS[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
      double a = 0, bq = 0;
      for (int k = 0; k < 3; ++k) { a += S[k * 3 + r] * gE[i * 9 + k * 3 + c]; bq += gE[i * 9 + r * 3 + k] * R[c * 3 + k]; }
      gR[r * 3 + c] += a; gS[r * 3 + c] = bq;
    }
    gt[0] += gS[7] - gS[5]; gt[1] += gS[2] - gS[6]; gt[2] += gS[3] - gS[1];
  }
  for (int k = 0; k < 3; ++k) { double t = 0; for (int j = 0; j < 9; ++j) t += gR[j] * dRk[k][j]; gvec[i * 6 + 3 + k] = static_cast<float>(t); }
  for (int k = 0; k < 3; ++k) gvec[i * 6 + k] = static_cast<float>(gt[k]);
}

// ====================================================================== warp_flow
// One thread per (sample, chunk of WF_CK channels, pixel): the PWC feature warps have 32-196 channels on
// images as small as 4x13, so parallelism has to come from the channels; the chunk's 2*WF_CK pair loads are
// all in flight together.  grid: x = pixel blocks of 64, y = channel chunk, z = sample.
constexpr int WF_CK = 8;

// cnt != nullptr (dfe_pwc_level_fwd_map): the blocks of the first channel chunk also count, per target pixel, the taps that hit it --
// the first pass of the inverse map the backward's gather walks (k_wfg_count's work, on the taps this thread computes anyway).
__global__ void __launch_bounds__(64) k_warp_flow_fwd(const float* __restrict__ x, const float* __restrict__ flow,
                                                      float* __restrict__ out, int C, int H, int W, int use_mask, int ac,
                                                      int* __restrict__ cnt) {
  const int b = blockIdx.z, c0 = blockIdx.y * WF_CK, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int py = p / W, px = p - py * W;
  const float* f = flow + static_cast<long>(b) * 2 * HW;
  float ix, iy;
  flow_coords(px, py, f[p], f[HW + p], H, W, ac, ix, iy);
  Tap t = make_tap(ix, iy, H, W);
  float keep = 1.0f;
  if (use_mask) keep = (tap_cover(t) < 0.9999f) ? 0.0f : 1.0f;
  if (cnt && blockIdx.y == 0 && keep != 0.0f) {          // the same taps, in-bounds tests and non-zero-weight rule as wfg_taps
    int* cb = cnt + static_cast<long>(b) * HW + t.y0 * W + t.x0;
    if (t.in_nw && t.nw != 0.0f) atomicAdd(cb, 1);
    if (t.in_ne && t.ne != 0.0f) atomicAdd(cb + 1, 1);
    if (t.in_sw && t.sw != 0.0f) atomicAdd(cb + W, 1);
    if (t.in_se && t.se != 0.0f) atomicAdd(cb + W + 1, 1);
  }
  const int nch = min(WF_CK, C - c0);
  Corners q[WF_CK];
#pragma unroll
  for (int c = 0; c < WF_CK; ++c) q[c] = load_corners(x + (static_cast<long>(b) * C + c0 + (c < nch ? c : 0)) * HW, t, W, H);
#pragma unroll
  for (int c = 0; c < WF_CK; ++c)
    if (c < nch) out[(static_cast<long>(b) * C + c0 + c) * HW + p] = interp(q[c], t) * keep;
}

// grad wrt flow (a sum over all channels) and optionally wrt x (scatter-add into the 64-bit fixed-point accumulators
// of gx_ws, dfe_scatter.h: order-independent, so gx is bitwise reproducible too).
// Block = 64 pixels x WF_GROUPS channel groups: group g walks the channel chunks g, g + WF_GROUPS, ... and keeps its
// partial (d/dix, d/diy) sums in registers; the groups' partials meet in LDS and are added in group order, so gflow
// is written once per pixel and is bitwise reproducible (no float atomics, no zero-fill).
// grid: x = pixel blocks of 64, y = 1, z = sample.
// WF_GROUPS: 4 on the large planes; 16 on the small ones (a PWC level of 16x52 is 104 blocks: every block walks all the
// channels, so the groups are what parallelism there is -- one round of loads per group instead of three)
template <int WF_GROUPS>
__global__ void __launch_bounds__(64 * WF_GROUPS) k_warp_flow_bwd(const float* __restrict__ x, const float* __restrict__ flow,
                                                      const float* __restrict__ gout, float* __restrict__ gflow,
                                                      void* __restrict__ gx_ws, const float* __restrict__ gflow_add, long gfa_bs,
                                                      int C, int H, int W, int use_mask, int ac) {
  __shared__ float red[WF_GROUPS][2][64];
  ScatterScale sc{};
  long long* gxq = nullptr;
  if (gx_ws) { sc = scatter_scale(*static_cast<const unsigned*>(gx_ws)); gxq = scatter_acc(gx_ws); }
  const int b = blockIdx.z, HW = H * W;
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the channel loop is scalar control flow
  const int p = blockIdx.x * 64 + lane;
  const bool live = p < HW;
  const int pc = live ? p : HW - 1;
  const int py = pc / W, px = pc - py * W;
  const float* f = flow + static_cast<long>(b) * 2 * HW;
  float ix, iy;
  flow_coords(px, py, f[pc], f[HW + pc], H, W, ac, ix, iy);
  Tap t = make_tap(ix, iy, H, W);
  float keep = live ? 1.0f : 0.0f;
  if (use_mask && tap_cover(t) < 0.9999f) keep = 0.0f;
  float gix = 0.0f, giy = 0.0f;
  // Neighbouring lanes are neighbouring pixels: where the flow is smooth, lane l's north-east / south-east taps are lane
  // l+1's north-west / south-west taps.  Integer sums may be regrouped freely, so such pairs are added in registers
  // (one DPP move) and scattered with ONE atomic: about half the atomics on the flows a trained (or freshly
  // initialised) PWC net produces.  The channel loop is wave-uniform, so every lane takes part in the moves.
  // (the moves first, unconditionally: a DPP move reads 0 from a lane that a short-circuited condition has switched off)
  const int y0_r = int_from_right_lane(t.y0), x0_r = int_from_right_lane(t.x0);
  const int y0_l = int_from_left_lane(t.y0), x0_l = int_from_left_lane(t.x0);
  const bool gives_right = (lane < 63) & (y0_r == t.y0) & (x0_r == t.x0 + 1);
  const bool takes_left = (lane > 0) & (y0_l == t.y0) & (x0_l + 1 == t.x0);
  for (int c0 = grp * WF_CK; c0 < C; c0 += WF_GROUPS * WF_CK) {
    const int nch = min(WF_CK, C - c0);
    float g[WF_CK];
#pragma unroll
    for (int c = 0; c < WF_CK; ++c) g[c] = (c < nch) ? gout[(static_cast<long>(b) * C + c0 + c) * HW + pc] * keep : 0.0f;
    if (gflow) {
      Corners q[WF_CK];
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) q[c] = load_corners(x + (static_cast<long>(b) * C + c0 + (c < nch ? c : 0)) * HW, t, W, H);
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) {
        float dx, dy;
        interp_grad(q[c], t, dx, dy);
        gix += g[c] * dx; giy += g[c] * dy;
      }
    }
    if (gxq) {
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) {
        if (c < nch) {     // wave-uniform
          const float gs = g[c] * sc.to_fixed;
          long long q_nw = t.in_nw ? to_fixed(gs, t.nw) : 0, q_ne = t.in_ne ? to_fixed(gs, t.ne) : 0;
          long long q_sw = t.in_sw ? to_fixed(gs, t.sw) : 0, q_se = t.in_se ? to_fixed(gs, t.se) : 0;
          const long long l_ne = fixed_from_left_lane(q_ne), l_se = fixed_from_left_lane(q_se);
          q_nw += takes_left ? l_ne : 0; q_sw += takes_left ? l_se : 0;
          q_ne = gives_right ? 0 : q_ne; q_se = gives_right ? 0 : q_se;
          long long* base = gxq + (static_cast<long>(b) * C + c0 + c) * HW + static_cast<long>(t.y0) * W + t.x0;
          fixed_add(base, q_nw);           // a zero is not added (and an out-of-range corner is zero)
          fixed_add(base + 1, q_ne);
          fixed_add(base + W, q_sw);
          fixed_add(base + W + 1, q_se);
        }
      }
    }
  }
  if (gflow) {
    red[grp][0][lane] = gix; red[grp][1][lane] = giy;
    __syncthreads();
    if (grp == 0 && live) {
      float sx = red[0][0][lane], sy = red[0][1][lane];
#pragma unroll
      for (int k = 1; k < WF_GROUPS; ++k) { sx += red[k][0][lane]; sy += red[k][1][lane]; }
      float* gf = gflow + static_cast<long>(b) * 2 * HW;
      float vx = sx * flow_coord_scale(W, ac), vy = sy * flow_coord_scale(H, ac);
      if (gflow_add) {   // a second gradient of the same flow (the concatenated copy of a PWC level), added here
        const float* ga = gflow_add + static_cast<long>(b) * gfa_bs;
        vx += ga[p]; vy += ga[HW + p];
      }
      gf[p] = vx; gf[HW + p] = vy;
    }
  }
}

// ---------------------------------------------------------------------- gx of warp_flow as a GATHER (round 4)
// The scatter above issues one 64-bit atomic per (channel, pixel, tap with a non-zero weight): with the zero flows of a
// freshly initialised PWC net that is one atomic per element (61 us at 32 x 64 x 208 x 8), with real flows four (224 us:
// profiles/r04_pwc_roofline_table.md).  The taps depend on the flow only, not on the channel, so for the large levels the
// inverse map is built once per call -- per target pixel the list of (source pixel, weight) pairs that hit it: count
// (int atomics on H*W counters), exclusive scan, fill -- and ONE thread per (channel chunk, target pixel) then sums its
// list in registers.  Every contribution is rounded exactly as the scatter rounds it (to_fixed(g * scale, w)) and the
// 64-bit integer sums do not depend on the order of the list, so the result is bit-identical to the scatter's, and
// reproducible, with no atomic on the gradient and no 8-byte accumulator per element.
// The lists live in the caller's scatter workspace (64 + 8 B*C*H*W bytes): header | cnt[B*HW] | off[B*(HW+1)] | ent[B*4*HW].
constexpr int WFG_EXACT_F64 = 65536;     // contributions a double adds exactly (each below 2^36: sums stay below 2^53)
constexpr int WFG_MIN_C = 8, WFG_MIN_HW = 512;     // (measured at the PWC levels: 16x52 gains, 8x26 and below are launch-bound either way)
static inline bool wfg_eligible(int C, long HW) { return C >= WFG_MIN_C && HW >= WFG_MIN_HW && HW < (1l << 28); }
static inline size_t wfg_head_bytes(int B, long HW) { return static_cast<size_t>(SCATTER_HEADER_BYTES) + 4ul * B * HW; }    // header + counters: what must be zero
struct WfgWs { unsigned* header; int* cnt; int* off; int2* ent; };
static inline WfgWs wfg_layout(void* ws, int B, long HW) {
  char* base = static_cast<char*>(ws);
  WfgWs w;
  w.header = reinterpret_cast<unsigned*>(base);
  w.cnt = reinterpret_cast<int*>(base + SCATTER_HEADER_BYTES);
  w.off = w.cnt + static_cast<long>(B) * HW;
  size_t o = SCATTER_HEADER_BYTES + 4ul * B * HW + 4ul * B * (HW + 1);
  o = (o + 15) & ~static_cast<size_t>(15);
  w.ent = reinterpret_cast<int2*>(base + o);
  return w;
}

// the (at most four) in-bounds taps with a non-zero weight of source pixel p: target index and weight
__device__ __forceinline__ int wfg_taps(const float* __restrict__ f, int p, int H, int W, int use_mask, int ac, int (&q)[4], float (&w)[4]) {
  const int HW = H * W, py = p / W, px = p - py * W;
  float ix, iy;
  flow_coords(px, py, f[p], f[HW + p], H, W, ac, ix, iy);
  const Tap t = make_tap(ix, iy, H, W);
  if (use_mask && tap_cover(t) < 0.9999f) return 0;
  int n = 0;
  const int base = t.y0 * W + t.x0;
  if (t.in_nw && t.nw != 0.0f) { q[n] = base; w[n++] = t.nw; }
  if (t.in_ne && t.ne != 0.0f) { q[n] = base + 1; w[n++] = t.ne; }
  if (t.in_sw && t.sw != 0.0f) { q[n] = base + W; w[n++] = t.sw; }
  if (t.in_se && t.se != 0.0f) { q[n] = base + W + 1; w[n++] = t.se; }
  return n;
}

// grid: (ceil(HW / 256), B)
__global__ void __launch_bounds__(256) k_wfg_count(const float* __restrict__ flow, int* __restrict__ cnt, int H, int W, int use_mask, int ac) {
  const int b = blockIdx.y, HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  int q[4]; float w[4];
  const int n = wfg_taps(flow + static_cast<long>(b) * 2 * HW, p, H, W, use_mask, ac, q, w);
  for (int k = 0; k < n; ++k) atomicAdd(cnt + static_cast<long>(b) * HW + q[k], 1);
}

// exclusive scan of one sample's counters; grid: B, block: 1024 (thread t owns a contiguous run of the counters)
__global__ void __launch_bounds__(1024) k_wfg_scan(const int* __restrict__ cnt, int* __restrict__ off, int HW) {
  __shared__ int part[1024];
  const int b = blockIdx.x, t = threadIdx.x, per = (HW + 1023) / 1024, lo = min(t * per, HW), hi = min(lo + per, HW);
  const int* c = cnt + static_cast<long>(b) * HW;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += c[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 run sums
    const int v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  int* o = off + static_cast<long>(b) * (HW + 1);
  for (int i = lo; i < hi; ++i) { o[i] = run; run += c[i]; }
  if (t == 1023) o[HW] = part[1023];
}

// grid: (ceil(HW / 256), B); the counters are counted back down to zero (slot = the value before the decrement, minus 1)
__global__ void __launch_bounds__(256) k_wfg_fill(const float* __restrict__ flow, int* __restrict__ cnt, const int* __restrict__ off,
                                                  int2* __restrict__ ent, int H, int W, int use_mask, int ac) {
  const int b = blockIdx.y, HW = H * W, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  int q[4]; float w[4];
  const int n = wfg_taps(flow + static_cast<long>(b) * 2 * HW, p, H, W, use_mask, ac, q, w);
  for (int k = 0; k < n; ++k) {
    const int slot = atomicSub(cnt + static_cast<long>(b) * HW + q[k], 1) - 1;
    ent[static_cast<long>(b) * 4 * HW + off[static_cast<long>(b) * (HW + 1) + q[k]] + slot] = make_int2(p, __float_as_int(w[k]));
  }
}

// The whole map of a SMALL plane (H*W <= 1024: PWC levels 4-6) in one launch: a block per sample, thread = source pixel; counters,
// exclusive scan and slot cursors live in LDS, so there is no zero-fill, no global counter array and no second walk of the flow.
// (For the large planes this form loses -- eight blocks cannot walk 13 312 pixels as fast as 416 can: round 5 -- and the three
// launches above stay.)  Block 0 also clears the word that receives the backward's bound.  grid: B
constexpr int WFG_SMALL_HW = 1024;
__global__ void __launch_bounds__(WFG_SMALL_HW) k_wfg_map_small(const float* __restrict__ flow, unsigned* __restrict__ header, int* __restrict__ off,
                                                                int2* __restrict__ ent, int H, int W, int use_mask, int ac) {
  __shared__ int cnt[WFG_SMALL_HW], pre[WFG_SMALL_HW], first[WFG_SMALL_HW];
  const int b = blockIdx.x, HW = H * W, t = threadIdx.x;
  cnt[t] = 0;
  if (b == 0 && t == 0) *header = 0u;
  __syncthreads();
  int q[4]; float w[4];
  const int n = t < HW ? wfg_taps(flow + static_cast<long>(b) * 2 * HW, t, H, W, use_mask, ac, q, w) : 0;
  for (int k = 0; k < n; ++k) atomicAdd(&cnt[q[k]], 1);
  __syncthreads();
  const int mine = cnt[t];
  pre[t] = mine;
  __syncthreads();
  for (int d = 1; d < WFG_SMALL_HW; d <<= 1) {          // Hillis-Steele inclusive scan
    const int v = (t >= d) ? pre[t - d] : 0;
    __syncthreads();
    pre[t] += v;
    __syncthreads();
  }
  first[t] = pre[t] - mine;
  int* o = off + static_cast<long>(b) * (HW + 1);
  if (t < HW) o[t] = pre[t] - mine;
  if (t == HW - 1) o[HW] = pre[t];
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const int slot = atomicSub(&cnt[q[k]], 1) - 1;
    ent[static_cast<long>(b) * 4 * HW + first[q[k]] + slot] = make_int2(t, __float_as_int(w[k]));
  }
}

// grid: (ceil(HW / 64), ceil(C / 8), B); block: one wave, lane = target pixel
__global__ void __launch_bounds__(64) k_wfg_gather(const float* __restrict__ gout, const int* __restrict__ off, const int2* __restrict__ ent,
                                                   const unsigned* __restrict__ header, float* __restrict__ gx, int C, int HW) {
  const int b = blockIdx.z, c0 = blockIdx.y * WF_CK, q = blockIdx.x * 64 + threadIdx.x;
  if (q >= HW) return;
  const ScatterScale sc = scatter_scale(*header);
  const int nch = min(WF_CK, C - c0);
  const int* o = off + static_cast<long>(b) * (HW + 1);
  const int e0 = o[q], e1 = o[q + 1];
  const int2* e = ent + static_cast<long>(b) * 4 * HW;
  const float* g = gout + (static_cast<long>(b) * C + c0) * HW;
  float* out = gx + (static_cast<long>(b) * C + c0) * HW + q;
  if (e1 - e0 <= WFG_EXACT_F64) {
    // Every contribution is an INTEGER of at most 24 significant bits below 2^36 (rint of a float): a double adds up to
    // 2^17 of them exactly, so the sum is the scatter's 64-bit integer sum whatever the order of the list -- at one
    // conversion and one add per contribution instead of a software float -> int64 conversion (44.6 -> us at 32 x 64 x 208 x 8)
    double acc[WF_CK];
#pragma unroll
    for (int c = 0; c < WF_CK; ++c) acc[c] = 0.0;
    // two list entries per trip: their 2 x 8 gradient loads are in flight together (a list is ~4 entries long where the flow is
    // smooth: one entry per trip left the thread a chain of dependent round trips -- offset, entry, gradient -- per entry).  Integer
    // sums in double are exact, so pairing changes no bit.
    int k = e0;
    for (; k + 1 < e1; k += 2) {
      const int2 pa = e[k], pb = e[k + 1];
      const float wa = __int_as_float(pa.y), wb = __int_as_float(pb.y);
      float va[WF_CK], vb[WF_CK];
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) {
        const long co = static_cast<long>(c < nch ? c : 0) * HW;
        va[c] = g[co + pa.x]; vb[c] = g[co + pb.x];
      }
#pragma unroll
      for (int c = 0; c < WF_CK; ++c)
        acc[c] += static_cast<double>(rintf((va[c] * sc.to_fixed) * wa)) + static_cast<double>(rintf((vb[c] * sc.to_fixed) * wb));
    }
    if (k < e1) {
      const int2 pw = e[k];
      const float w = __int_as_float(pw.y);
      float v[WF_CK];
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) v[c] = g[static_cast<long>(c < nch ? c : 0) * HW + pw.x];
#pragma unroll
      for (int c = 0; c < WF_CK; ++c) acc[c] += static_cast<double>(rintf((v[c] * sc.to_fixed) * w));
    }
#pragma unroll
    for (int c = 0; c < WF_CK; ++c)
      if (c < nch) out[static_cast<long>(c) * HW] = (sc.to_fixed != sc.to_fixed) ? sc.to_fixed : static_cast<float>(acc[c] * sc.to_float);
    return;
  }
  long long acc[WF_CK];          // a pile-up of more than 2^16 contributions in one pixel: the integer accumulators
#pragma unroll
  for (int c = 0; c < WF_CK; ++c) acc[c] = 0;
  for (int k = e0; k < e1; ++k) {
    const int2 pw = e[k];
    const float w = __int_as_float(pw.y);
    float v[WF_CK];
#pragma unroll
    for (int c = 0; c < WF_CK; ++c) v[c] = g[static_cast<long>(c < nch ? c : 0) * HW + pw.x];
#pragma unroll
    for (int c = 0; c < WF_CK; ++c) acc[c] += to_fixed(v[c] * sc.to_fixed, w);       // the scatter's own rounding of each contribution
  }
#pragma unroll
  for (int c = 0; c < WF_CK; ++c)
    if (c < nch) out[static_cast<long>(c) * HW] = from_fixed(acc[c], sc);
}

// ====================================================================== inverse_warp2 / rigid flow
__global__ void k_inverse_warp2_fwd(const float* __restrict__ img, const float* __restrict__ depth,
                                    const float* __restrict__ ref_depth, const Camera* __restrict__ cams,
                                    float* __restrict__ out_img, float* __restrict__ out_valid,
                                    float* __restrict__ out_pdepth, float* __restrict__ out_cdepth,
                                    int H, int W, int ac) {
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int py = p / W, px = p - py * W;
  const Camera& cam = cams[b];
  Proj pr = project(cam, px, py, depth[static_cast<long>(b) * HW + p]);
  float xn, yn; bool lx, ly;
  rigid_grid(pr, H, W, xn, yn, lx, ly);
  Tap t = make_tap(unnormalize(xn, W, ac), unnormalize(yn, H, ac), H, W);
  for (int c = 0; c < 3; ++c) {
    const float* plane = img + (static_cast<long>(b) * 3 + c) * HW;
    out_img[(static_cast<long>(b) * 3 + c) * HW + p] = interp(load_corners(plane, t, W, H), t);
  }
  if (out_valid) out_valid[static_cast<long>(b) * HW + p] = (fmaxf(fabsf(xn), fabsf(yn)) <= 1.0f) ? 1.0f : 0.0f;
  if (out_pdepth) {
    float v = interp(load_corners(ref_depth + static_cast<long>(b) * HW, t, W, H), t);
    out_pdepth[static_cast<long>(b) * HW + p] = (v >= 1e-3f || v != v) ? v : 1e-3f;
  }
  if (out_cdepth) out_cdepth[static_cast<long>(b) * HW + p] = pr.Z;
}

// Backward of inverse_warp2: grads wrt depth, ref_depth (order-independent scatter, dfe_scatter.h) and the 12 camera sums.
// partials [B][1][gridDim.x][12].
__global__ void k_inverse_warp2_bwd(const float* __restrict__ img, const float* __restrict__ depth,
                                    const float* __restrict__ ref_depth, const Camera* __restrict__ cams,
                                    const float* __restrict__ g_img, const float* __restrict__ g_pdepth,
                                    const float* __restrict__ g_cdepth, float* __restrict__ g_depth,
                                    void* __restrict__ g_refdepth_ws, float* __restrict__ partials,
                                    int H, int W, int ac) {
  __shared__ float red[12 * 16];
  ScatterScale sc{};
  long long* g_refq = nullptr;
  if (g_refdepth_ws) { sc = scatter_scale(*static_cast<const unsigned*>(g_refdepth_ws)); g_refq = scatter_acc(g_refdepth_ws); }
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  if (p < HW) {
    const int py = p / W, px = p - py * W;
    const Camera& cam = cams[b];
    const float d = depth[static_cast<long>(b) * HW + p];
    Proj pr = project(cam, px, py, d);
    float xn, yn; bool lx, ly;
    rigid_grid(pr, H, W, xn, yn, lx, ly);
    Tap t = make_tap(unnormalize(xn, W, ac), unnormalize(yn, H, ac), H, W);
    float gix = 0.0f, giy = 0.0f;
    if (g_img) {
      for (int c = 0; c < 3; ++c) {
        const long off = (static_cast<long>(b) * 3 + c) * HW;
        float dx, dy;
        interp_grad(load_corners(img + off, t, W, H), t, dx, dy);
        float g = g_img[off + p];
        gix += g * dx; giy += g * dy;
      }
    }
    if (g_pdepth) {
      Corners q = load_corners(ref_depth + static_cast<long>(b) * HW, t, W, H);
      float v = interp(q, t);
      float g = (v >= 1e-3f) ? g_pdepth[static_cast<long>(b) * HW + p] : 0.0f;
      float dx, dy;
      interp_grad(q, t, dx, dy);
      gix += g * dx; giy += g * dy;
      if (g_refq && g != 0.0f) {
        long long* base = g_refq + static_cast<long>(b) * HW + static_cast<long>(t.y0) * W + t.x0;
        const float gs = g * sc.to_fixed;
        if (t.in_nw) fixed_add(base, to_fixed(gs, t.nw));
        if (t.in_ne) fixed_add(base + 1, to_fixed(gs, t.ne));
        if (t.in_sw) fixed_add(base + W, to_fixed(gs, t.sw));
        if (t.in_se) fixed_add(base + W + 1, to_fixed(gs, t.se));
      }
    }
    const float sx = ac ? static_cast<float>(W - 1) / 2.0f : static_cast<float>(W) / 2.0f;
    const float sy = ac ? static_cast<float>(H - 1) / 2.0f : static_cast<float>(H) / 2.0f;
    float gU = lx ? gix * sx * (2.0f / static_cast<float>(W - 1)) : 0.0f;
    float gV = ly ? giy * sy * (2.0f / static_cast<float>(H - 1)) : 0.0f;
    float gZ = g_cdepth ? g_cdepth[static_cast<long>(b) * HW + p] : 0.0f;
    float gd;
    project_backward(pr, d, gU, gV, gZ, gd, acc);
    g_depth[static_cast<long>(b) * HW + p] = gd;
  }
  block_sum<12>(acc, red, partials + (static_cast<long>(b) * gridDim.x + blockIdx.x) * 12);
}

__global__ void k_rigid_flow_fwd(const float* __restrict__ depth, const Camera* __restrict__ cams,
                                 float* __restrict__ out, int H, int W) {
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int py = p / W, px = p - py * W;
  Proj pr = project(cams[b], px, py, depth[static_cast<long>(b) * HW + p]);
  out[static_cast<long>(b) * 2 * HW + p] = pr.U - static_cast<float>(px);
  out[static_cast<long>(b) * 2 * HW + HW + p] = pr.V - static_cast<float>(py);
}

__global__ void k_rigid_flow_bwd(const float* __restrict__ depth, const Camera* __restrict__ cams,
                                 const float* __restrict__ gout, float* __restrict__ g_depth,
                                 float* __restrict__ partials, int H, int W) {
  __shared__ float red[12 * 16];
  const int b = blockIdx.y, HW = H * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  if (p < HW) {
    const int py = p / W, px = p - py * W;
    const float d = depth[static_cast<long>(b) * HW + p];
    Proj pr = project(cams[b], px, py, d);
    float gd;
    project_backward(pr, d, gout[static_cast<long>(b) * 2 * HW + p], gout[static_cast<long>(b) * 2 * HW + HW + p], 0.0f, gd, acc);
    g_depth[static_cast<long>(b) * HW + p] = gd;
  }
  block_sum<12>(acc, red, partials + (static_cast<long>(b) * gridDim.x + blockIdx.x) * 12);
}

// ====================================================================== SSIM (3x3 box, zero pad)
constexpr int SS_TX = 32, SS_TY = 8;

__global__ void k_ssim_fwd(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                           int H, int W) {
  __shared__ float sx[SS_TY + 2][SS_TX + 2], sy[SS_TY + 2][SS_TX + 2];
  const long plane = static_cast<long>(blockIdx.z) * H * W;
  const int x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
  for (int i = threadIdx.x; i < (SS_TY + 2) * (SS_TX + 2); i += blockDim.x) {
    int ly = i / (SS_TX + 2), lx = i - ly * (SS_TX + 2);
    int gy = y0 + ly - 1, gx = x0 + lx - 1;
    bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[ly][lx] = in ? x[plane + static_cast<long>(gy) * W + gx] : 0.0f;
    sy[ly][lx] = in ? y[plane + static_cast<long>(gy) * W + gx] : 0.0f;
  }
  __syncthreads();
  const int tx = threadIdx.x % SS_TX, ty = threadIdx.x / SS_TX;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx >= W || gy >= H) return;
  float a = 0, bq = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float u = sx[ty + dy][tx + dx], v = sy[ty + dy][tx + dx];
      a += u; bq += v; aa += u * u; bb += v * v; ab += u * v;
    }
  out[plane + static_cast<long>(gy) * W + gx] = ssim_from_means(a / 9.0f, bq / 9.0f, aa / 9.0f, bb / 9.0f, ab / 9.0f);
}

__global__ void k_ssim_bwd(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gout,
                           float* __restrict__ gx_out, float* __restrict__ gy_out, int H, int W) {
  __shared__ float sx[SS_TY + 4][SS_TX + 4], sy[SS_TY + 4][SS_TX + 4];
  __shared__ float co[5][SS_TY + 2][SS_TX + 2];
  const long plane = static_cast<long>(blockIdx.z) * H * W;
  const int x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
  for (int i = threadIdx.x; i < (SS_TY + 4) * (SS_TX + 4); i += blockDim.x) {
    int ly = i / (SS_TX + 4), lx = i - ly * (SS_TX + 4);
    int gy = y0 + ly - 2, gx = x0 + lx - 2;
    bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[ly][lx] = in ? x[plane + static_cast<long>(gy) * W + gx] : 0.0f;
    sy[ly][lx] = in ? y[plane + static_cast<long>(gy) * W + gx] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (SS_TY + 2) * (SS_TX + 2); i += blockDim.x) {
    int ly = i / (SS_TX + 2), lx = i - ly * (SS_TX + 2);
    int gy = y0 + ly - 1, gx = x0 + lx - 1;
    float c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      float a = 0, bq = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          float u = sx[ly + dy][lx + dx], v = sy[ly + dy][lx + dx];
          a += u; bq += v; aa += u * u; bb += v * v; ab += u * v;
        }
      float g = gout[plane + static_cast<long>(gy) * W + gx];
      ssim_partials(a / 9.0f, bq / 9.0f, aa / 9.0f, bb / 9.0f, ab / 9.0f, c0, c1, c2, c3, c4);
      c0 *= g; c1 *= g; c2 *= g; c3 *= g; c4 *= g;
    }
    co[0][ly][lx] = c0; co[1][ly][lx] = c1; co[2][ly][lx] = c2; co[3][ly][lx] = c3; co[4][ly][lx] = c4;
  }
  __syncthreads();
  const int tx = threadIdx.x % SS_TX, ty = threadIdx.x / SS_TX;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx >= W || gy >= H) return;
  float s[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[k] += co[k][ty + dy][tx + dx];
  float xv = sx[ty + 2][tx + 2], yv = sy[ty + 2][tx + 2];
  if (gx_out) gx_out[plane + static_cast<long>(gy) * W + gx] = (s[0] + 2.0f * xv * s[2] + yv * s[4]) / 9.0f;
  if (gy_out) gy_out[plane + static_cast<long>(gy) * W + gx] = (s[1] + 2.0f * yv * s[3] + xv * s[4]) / 9.0f;
}

// ====================================================================== correlation (d = 4, 81 taps): ops_corr.hip
constexpr int CR_D = 4, CR_K = 2 * CR_D + 1;

// ====================================================================== resize (pyramids)
// mode 0: bilinear align_corners=False (F.interpolate); mode 1: area (adaptive average pool)
// mult / pre: out = resize(in * mult) (pre) or resize(in) * mult (post), each rounded like the ATen composition
// (PWC_tf scales its flows on either side of F.interpolate, pwc_tf.py:118,175-178)
__global__ void k_resize(const float* __restrict__ in, float* __restrict__ out, int planes, int inH, int inW,
                         int outH, int outW, int mode, float mult, int pre) {
  const long n = static_cast<long>(planes) * outH * outW;
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ox = static_cast<int>(i % outW), oy = static_cast<int>((i / outW) % outH);
  const long pl = i / (static_cast<long>(outW) * outH);
  const float* src = in + pl * inH * inW;
  if (mode == 0) {
    if (mult == 1.0f) {
      out[i] = resize_bilinear_at(src, inH, inW, oy, ox, static_cast<float>(inH) / outH, static_cast<float>(inW) / outW,
                                  aten_small_resize(outH, outW));
    } else {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      bilinear_src(oy, static_cast<float>(inH) / outH, inH, y0, y1, ly0, ly1);
      bilinear_src(ox, static_cast<float>(inW) / outW, inW, x0, x1, lx0, lx1);
      const float* r0 = src + static_cast<long>(y0) * inW;
      const float* r1 = src + static_cast<long>(y1) * inW;
      const float pm = pre ? mult : 1.0f;
      const float v = lerp2_aten_sel(aten_small_resize(outH, outW), r0[x0] * pm, r0[x1] * pm, r1[x0] * pm, r1[x1] * pm, lx0, lx1, ly0, ly1);
      out[i] = pre ? v : v * mult;
    }
  } else {
    // adaptive_avg_pool2d window: [floor(o*in/out), ceil((o+1)*in/out))
    int ys = (oy * inH) / outH, ye = ((oy + 1) * inH + outH - 1) / outH;
    int xs = (ox * inW) / outW, xe = ((ox + 1) * inW + outW - 1) / outW;
    float s = 0.0f;
    for (int yy = ys; yy < ye; ++yy)
      for (int xx = xs; xx < xe; ++xx) s += src[static_cast<long>(yy) * inW + xx];
    out[i] = s / static_cast<float>((ye - ys) * (xe - xs));
  }
}

// Adjoint of the bilinear resize as a gather (no atomics: reproducible, unlike ATen's upsample backward): one thread
// per INPUT element collects the <= G2_MAX x G2_MAX outputs whose taps touch it.  gin = adjoint(gout) * mult (pre) or
// adjoint(gout * mult) (post), mirroring the order of the autograd chain.
__global__ void __launch_bounds__(256) k_resize_bilinear_bwd(const float* __restrict__ gout, float* __restrict__ gin, int planes,
                                                             int inH, int inW, int outH, int outW, float mult, int pre) {
  const long n = static_cast<long>(planes) * inH * inW;
  const long e = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int j = static_cast<int>(e % inW), i = static_cast<int>((e / inW) % inH);
  const long pl = e / (static_cast<long>(inW) * inH);
  const float* g = gout + pl * outH * outW;
  int ylo, ny, xlo, nx;
  float wy[G2_MAX], wx[G2_MAX];
  adj_weights(i, static_cast<float>(inH) / outH, inH, outH, ylo, ny, wy);
  adj_weights(j, static_cast<float>(inW) / outW, inW, outW, xlo, nx, wx);
  const float gm = pre ? 1.0f : mult;
  float total = 0.0f;
#pragma unroll
  for (int ky = 0; ky < G2_MAX; ++ky) {
    if (wy[ky] == 0.0f) continue;
    const float* row = g + static_cast<long>(ylo + ky) * outW + xlo;
    float acc = 0.0f;
#pragma unroll
    for (int kx = 0; kx < G2_MAX; ++kx)
      if (wx[kx] != 0.0f) acc += wx[kx] * (row[kx] * gm);
    total += wy[ky] * acc;
  }
  gin[e] = pre ? total * mult : total;
}

}  // namespace dfe

// ====================================================================== C ABI
using namespace dfe;

#define DFE_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define DFE_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return DFE_ERR_LAUNCH; } while (0)

static inline dim3 grid1d(long n, int bs) { return dim3(static_cast<unsigned>((n + bs - 1) / bs)); }

// warp_flow's backward kernel with 4 channel groups per block, or 16 where the plane is small
static void launch_warp_flow_bwd(dim3 g, hipStream_t st, const float* x, const float* flow, const float* gout, float* gflow, void* gx_ws,
                                 const float* gflow_add, long gfa_bs, int C, int H, int W, int use_mask, int ac) {
  if (static_cast<long>(H) * W < 2048 && C > 4 * WF_CK)
    k_warp_flow_bwd<16><<<g, 64 * 16, 0, st>>>(x, flow, gout, gflow, gx_ws, gflow_add, gfa_bs, C, H, W, use_mask, ac);
  else
    k_warp_flow_bwd<4><<<g, 64 * 4, 0, st>>>(x, flow, gout, gflow, gx_ws, gflow_add, gfa_bs, C, H, W, use_mask, ac);
}

// gx of warp_flow by the gather path: the workspace's header holds the bound and its counters are zero (wfg_head_bytes).
// (Round 5 measured the map built by ONE launch -- a block per sample, counters and cursors in LDS, in-block scan -- against these
// three: 143.6 / 81.3 / 50.0 us for the level backward at 64x208 / 32x104 / 16x52, B = 8, against 133.2 / 82.1 / 54.2: eight blocks
// cannot walk 13 312 pixels three times as fast as 416 can; removed.)
static int warp_gx_gather(const float* flow, const float* gout, float* gx, void* ws, int B, int C, int H, int W, int use_mask, int ac,
                          hipStream_t st) {
  const long HW = static_cast<long>(H) * W;
  const WfgWs w = wfg_layout(ws, B, HW);
  const dim3 gp(static_cast<unsigned>((HW + 255) / 256), B);
  k_wfg_count<<<gp, 256, 0, st>>>(flow, w.cnt, H, W, use_mask, ac);
  DFE_LAUNCH_CHECK();
  k_wfg_scan<<<B, 1024, 0, st>>>(w.cnt, w.off, static_cast<int>(HW));
  DFE_LAUNCH_CHECK();
  k_wfg_fill<<<gp, 256, 0, st>>>(flow, w.cnt, w.off, w.ent, H, W, use_mask, ac);
  DFE_LAUNCH_CHECK();
  k_wfg_gather<<<dim3(static_cast<unsigned>((HW + 63) / 64), (C + WF_CK - 1) / WF_CK, B), 64, 0, st>>>(gout, w.off, w.ent, w.header, gx, C, static_cast<int>(HW));
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

extern "C" {

int dfe_camera_floats(void) { return static_cast<int>(sizeof(Camera) / sizeof(float)); }

int dfe_prepare_cameras(const float* pose, const float* K, float* cams, int B, int ndir, int nscale,
                        const float* downscale_host, void* stream) {
  DFE_REQUIRE(pose && K && cams && downscale_host, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && ndir > 0 && nscale > 0 && nscale <= DFE_MAX_SCALES, DFE_ERR_DIMS);
  ScaleList dl;
  for (int s = 0; s < nscale; ++s) dl.v[s] = downscale_host[s];
  int n = B * ndir * nscale;
  k_prepare_cameras<<<grid1d(n, 64), 64, 0, static_cast<hipStream_t>(stream)>>>(pose, K, reinterpret_cast<Camera*>(cams), B, ndir, nscale, dl);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_pose_vec2mat_fwd(const float* vec, float* T34, float* E, int n, void* stream) {
  DFE_REQUIRE(vec && (T34 || E), DFE_ERR_NULL);
  DFE_REQUIRE(n > 0, DFE_ERR_DIMS);
  k_pose_mats<<<grid1d(n, 64), 64, 0, static_cast<hipStream_t>(stream)>>>(vec, T34, E, n);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_pose_vec2mat_bwd(const float* vec, const float* gT34, const float* gE, float* gvec, int n, void* stream) {
  DFE_REQUIRE(vec && gvec && (gT34 || gE), DFE_ERR_NULL);
  DFE_REQUIRE(n > 0, DFE_ERR_DIMS);
  k_pose_mats_bwd<<<grid1d(n, 64), 64, 0, static_cast<hipStream_t>(stream)>>>(vec, gT34, gE, gvec, n);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_warp_flow_fwd(const float* x, const float* flow, float* out, int B, int C, int H, int W, int use_mask,
                      int align_corners, void* stream) {
  DFE_REQUIRE(x && flow && out, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(B <= 65535 && (C + WF_CK - 1) / WF_CK <= 65535, DFE_ERR_DIMS);
  dim3 g(static_cast<unsigned>((static_cast<long>(H) * W + 63) / 64), (C + WF_CK - 1) / WF_CK, B);
  k_warp_flow_fwd<<<g, 64, 0, static_cast<hipStream_t>(stream)>>>(x, flow, out, C, H, W, use_mask, align_corners, nullptr);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_warp_flow_bwd(const float* x, const float* flow, const float* gout, float* gflow, float* gx, void* gx_ws, int B,
                      int C, int H, int W, int use_mask, int align_corners, void* stream) {
  DFE_REQUIRE(x && flow && gout && (gflow || gx), DFE_ERR_NULL);
  DFE_REQUIRE(!gx || gx_ws, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(B <= 65535 && (C + WF_CK - 1) / WF_CK <= 65535, DFE_ERR_DIMS);
  dim3 g(static_cast<unsigned>((static_cast<long>(H) * W + 63) / 64), 1, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long HW = static_cast<long>(H) * W, n = static_cast<long>(B) * C * HW;
  if (gx && wfg_eligible(C, HW) && getenv("DFE_WARP_SCATTER") == nullptr) {
    // large planes: gx as a gather over the inverse map of the flow (above); the bound of the contributions first
    if (reinterpret_cast<uintptr_t>(gx_ws) & 15) return DFE_ERR_DIMS;
    if (hipMemsetAsync(gx_ws, 0, wfg_head_bytes(B, HW), st) != hipSuccess) return DFE_ERR_LAUNCH;
    { const int rc = scatter_amax_into(static_cast<unsigned*>(gx_ws), gout, n, st); if (rc != DFE_OK) return rc; }
    if (gflow) {
      launch_warp_flow_bwd(g, st, x, flow, gout, gflow, nullptr, nullptr, 0, C, H, W, use_mask, align_corners);
      DFE_LAUNCH_CHECK();
    }
    return warp_gx_gather(flow, gout, gx, gx_ws, B, C, H, W, use_mask, align_corners, st);
  }
  if (gx) { const int rc = scatter_begin(gx_ws, n, gout, n, st); if (rc != DFE_OK) return rc; }
  launch_warp_flow_bwd(g, st, x, flow, gout, gflow, gx ? gx_ws : nullptr, nullptr, 0, C, H, W, use_mask, align_corners);
  DFE_LAUNCH_CHECK();
  if (gx) return scatter_finish(gx_ws, gx, n, st);
  return DFE_OK;
}

int dfe_pose_partials_floats(int B, int H, int W) {
  return B * static_cast<int>((static_cast<long>(H) * W + 255) / 256) * 12;
}

int dfe_inverse_warp2_fwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          float* out_img, float* out_valid, float* out_pdepth, float* out_cdepth, int B, int H, int W,
                          int align_corners, void* stream) {
  DFE_REQUIRE(img && depth && cams && out_img, DFE_ERR_NULL);
  DFE_REQUIRE(!out_pdepth || ref_depth, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && H > 1 && W > 1, DFE_ERR_DIMS);
  dim3 g(static_cast<unsigned>((static_cast<long>(H) * W + 255) / 256), B);
  k_inverse_warp2_fwd<<<g, 256, 0, static_cast<hipStream_t>(stream)>>>(img, depth, ref_depth, reinterpret_cast<const Camera*>(cams), out_img, out_valid, out_pdepth, out_cdepth, H, W, align_corners);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_inverse_warp2_bwd(const float* img, const float* depth, const float* ref_depth, const float* cams,
                          const float* g_img, const float* g_pdepth, const float* g_cdepth, float* g_depth,
                          float* g_refdepth, void* g_refdepth_ws, float* g_pose, float* partials, int B, int H, int W,
                          int align_corners, void* stream) {
  DFE_REQUIRE(img && depth && cams && g_depth && g_pose && partials, DFE_ERR_NULL);
  DFE_REQUIRE(!g_pdepth || ref_depth, DFE_ERR_NULL);
  DFE_REQUIRE(!g_refdepth || (g_refdepth_ws && g_pdepth), DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && H > 1 && W > 1, DFE_ERR_DIMS);
  const int nblk = static_cast<int>((static_cast<long>(H) * W + 255) / 256);
  dim3 g(nblk, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long n = static_cast<long>(B) * H * W;
  if (g_refdepth) { const int rc = scatter_begin(g_refdepth_ws, n, g_pdepth, n, st); if (rc != DFE_OK) return rc; }
  k_inverse_warp2_bwd<<<g, 256, 0, st>>>(img, depth, ref_depth, reinterpret_cast<const Camera*>(cams), g_img, g_pdepth, g_cdepth, g_depth, g_refdepth ? g_refdepth_ws : nullptr, partials, H, W, align_corners);
  DFE_LAUNCH_CHECK();
  if (g_refdepth) { const int rc = scatter_finish(g_refdepth_ws, g_refdepth, n, st); if (rc != DFE_OK) return rc; }
  IntList nb; nb.v[0] = nblk;
  k_pose_finalize<<<grid1d(B, 64), 64, 0, st>>>(partials, reinterpret_cast<const Camera*>(cams), g_pose, B, 1, nblk, nb, 0);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_rigid_flow_fwd(const float* depth, const float* cams, float* out, int B, int H, int W, void* stream) {
  DFE_REQUIRE(depth && cams && out, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  dim3 g(static_cast<unsigned>((static_cast<long>(H) * W + 255) / 256), B);
  k_rigid_flow_fwd<<<g, 256, 0, static_cast<hipStream_t>(stream)>>>(depth, reinterpret_cast<const Camera*>(cams), out, H, W);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_rigid_flow_bwd(const float* depth, const float* cams, const float* gout, float* g_depth, float* g_pose,
                       float* partials, int B, int H, int W, void* stream) {
  DFE_REQUIRE(depth && cams && gout && g_depth && g_pose && partials, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  const int nblk = static_cast<int>((static_cast<long>(H) * W + 255) / 256);
  dim3 g(nblk, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  k_rigid_flow_bwd<<<g, 256, 0, st>>>(depth, reinterpret_cast<const Camera*>(cams), gout, g_depth, partials, H, W);
  DFE_LAUNCH_CHECK();
  IntList nb; nb.v[0] = nblk;
  k_pose_finalize<<<grid1d(B, 64), 64, 0, st>>>(partials, reinterpret_cast<const Camera*>(cams), g_pose, B, 1, nblk, nb, 0);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, void* stream) {
  DFE_REQUIRE(x && y && out, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && static_cast<long>(B) * C <= 65535, DFE_ERR_DIMS);
  dim3 g((W + SS_TX - 1) / SS_TX, (H + SS_TY - 1) / SS_TY, B * C);
  k_ssim_fwd<<<g, SS_TX * SS_TY, 0, static_cast<hipStream_t>(stream)>>>(x, y, out, H, W);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_ssim_bwd(const float* x, const float* y, const float* gout, float* gx, float* gy, int B, int C, int H, int W,
                 void* stream) {
  DFE_REQUIRE(x && y && gout && (gx || gy), DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && static_cast<long>(B) * C <= 65535, DFE_ERR_DIMS);
  dim3 g((W + SS_TX - 1) / SS_TX, (H + SS_TY - 1) / SS_TY, B * C);
  k_ssim_bwd<<<g, SS_TX * SS_TY, 0, static_cast<hipStream_t>(stream)>>>(x, y, gout, gx, gy, H, W);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_corr_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int d, void* stream) {
  DFE_REQUIRE(f1 && f2 && out, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(d == CR_D, DFE_ERR_UNSUPPORTED);
  const int rc = launch_corr_fwd(f1, f2, out, static_cast<long>(CR_K) * CR_K * H * W, nullptr, nullptr, B, C, H, W, static_cast<hipStream_t>(stream));
  if (rc != DFE_OK) return rc;
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_corr_bwd(const float* f1, const float* f2, const float* gout, float* g1, float* g2, int B, int C, int H, int W,
                 int d, void* stream) {
  DFE_REQUIRE(f1 && f2 && gout && (g1 || g2), DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(d == CR_D, DFE_ERR_UNSUPPORTED);
  const int rc = launch_corr_bwd(f1, f2, gout, static_cast<long>(CR_K) * CR_K * H * W, nullptr, 0, g1, g2, nullptr, B, C, H, W, static_cast<hipStream_t>(stream));
  if (rc != DFE_OK) return rc;
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

// ---------------------------------------------------------------- one PWC decoder level's input (pwc_tf.py:119-121)
// x = cat(corr(c1, warp(c2, flow)), c1, flow): the cost volume is written straight into its slice of x and the
// backward pass reads the three slices of dL/dx in place (no cat / slice copies, no gradient-accumulation passes).
int dfe_pwc_level_channels(int C) { return CR_K * CR_K + C + 2; }

int dfe_pwc_level_fwd(const float* c1, const float* c2, const float* flow, float* warped, float* x, int B, int C, int H,
                      int W, int align_corners, void* stream) {
  DFE_REQUIRE(c1 && c2 && flow && warped && x, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(B <= 65535 && (C + WF_CK - 1) / WF_CK <= 65535, DFE_ERR_DIMS);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long HW = static_cast<long>(H) * W, xbs = static_cast<long>(dfe_pwc_level_channels(C)) * HW;
  dim3 gw(static_cast<unsigned>((HW + 63) / 64), (C + WF_CK - 1) / WF_CK, B);
  k_warp_flow_fwd<<<gw, 64, 0, st>>>(c2, flow, warped, C, H, W, 0, align_corners, nullptr);
  DFE_LAUNCH_CHECK();
  // the correlation kernel also writes the c1 and flow planes behind the cost volume (the c1 tiles it stages anyway)
  { const int rc = launch_corr_fwd(c1, warped, x, xbs, x + static_cast<long>(CR_K) * CR_K * HW, flow, B, C, H, W, st); if (rc != DFE_OK) return rc; }
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_pwc_level_bwd(const float* c1, const float* c2, const float* flow, const float* warped, const float* gx,
                      float* g_warped, float* g_c1, float* g_c2, void* g_c2_ws, float* g_flow, int B, int C, int H, int W,
                      int align_corners, void* stream) {
  DFE_REQUIRE(c1 && c2 && flow && warped && gx && g_warped && g_c1, DFE_ERR_NULL);
  DFE_REQUIRE(g_c2 || g_flow, DFE_ERR_NULL);
  DFE_REQUIRE(!g_c2 || g_c2_ws, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(B <= 65535, DFE_ERR_DIMS);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long HW = static_cast<long>(H) * W, xbs = static_cast<long>(dfe_pwc_level_channels(C)) * HW;
  const float* gx_c1 = gx + static_cast<long>(CR_K) * CR_K * HW;
  const float* gx_flow = gx_c1 + static_cast<long>(C) * HW;
  // the feature-warp gradient's workspace is zeroed first: the correlation backward writes the bound of its contributions
  // (max |dL/dwarped|, the gradient about to be scattered) from its epilogue -- round 3 spent a launch of its own on that max
  const long n = static_cast<long>(B) * C * HW;
  const bool gather = g_c2 && wfg_eligible(C, HW) && getenv("DFE_WARP_SCATTER") == nullptr;
  if (gather) {
    if (reinterpret_cast<uintptr_t>(g_c2_ws) & 15) return DFE_ERR_DIMS;
    if (hipMemsetAsync(g_c2_ws, 0, wfg_head_bytes(B, HW), st) != hipSuccess) return DFE_ERR_LAUNCH;
  } else if (g_c2) { const int rc = scatter_begin_bound(g_c2_ws, n, st); if (rc != DFE_OK) return rc; }
  // dL/dc1 = correlation gradient + the concatenated copy's slice; dL/dwarped
  { const int rc = launch_corr_bwd(c1, warped, gx, xbs, gx_c1, xbs, g_c1, g_warped, g_c2 ? static_cast<unsigned*>(g_c2_ws) : nullptr, B, C, H, W, st); if (rc != DFE_OK) return rc; }
  DFE_LAUNCH_CHECK();
  dim3 g(static_cast<unsigned>((HW + 63) / 64), 1, B);
  if (!gather || g_flow) {
    launch_warp_flow_bwd(g, st, c2, flow, g_warped, g_flow, (g_c2 && !gather) ? g_c2_ws : nullptr, g_flow ? gx_flow : nullptr, xbs, C, H, W, 0, align_corners);
    DFE_LAUNCH_CHECK();
  }
  if (gather) return warp_gx_gather(flow, g_warped, g_c2, g_c2_ws, B, C, H, W, 0, align_corners, st);
  if (g_c2) return scatter_finish(g_c2_ws, g_c2, n, st);
  return DFE_OK;
}

// ---------------------------------------------------------------- the same level with the inverse map built in the FORWARD pass
// (round 6).  The backward's gather needs, per target pixel of c2, the list of (source pixel, weight) taps that
// hit it; the taps depend on the flow only, which the forward's feature warp reads anyway.  dfe_pwc_level_fwd_map counts them in
// the warp kernel and finishes the map (scan, fill) behind it; `map` (dfe_pwc_level_map_bytes) travels to the backward, which is then
// three launches at every level -- correlation gradients (+ the bound of dL/dwarped), flow gradient, gather -- with no zero-fill, no
// counting passes and, on the small levels, no 64-bit atomics and no conversion pass.  Same bits as dfe_pwc_level_bwd: every
// contribution is rounded as the scatter rounds it and integer sums do not depend on the order of a list.  C >= 8.
// A second backward pass through the same map (retain_graph) finds the first pass's bound in the header: the larger of the two is
// used (still a bound of every contribution; the quantum is then that of the larger one).
long dfe_pwc_level_map_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return DFE_ERR_DIMS;
  const long HW = static_cast<long>(H) * W;
  long o = SCATTER_HEADER_BYTES + 4L * B * HW + 4L * B * (HW + 1);      // wfg_layout: header | cnt | off | (16-byte aligned) ent
  o = (o + 15) & ~15L;
  return o + 8L * 4 * B * HW;
}

int dfe_pwc_level_fwd_map(const float* c1, const float* c2, const float* flow, float* warped, float* x, void* map, int B, int C,
                          int H, int W, int align_corners, void* stream) {
  DFE_REQUIRE(c1 && c2 && flow && warped && x && map, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(B <= 65535 && (C + WF_CK - 1) / WF_CK <= 65535, DFE_ERR_DIMS);
  const long HW = static_cast<long>(H) * W, xbs = static_cast<long>(dfe_pwc_level_channels(C)) * HW;
  DFE_REQUIRE(C >= WFG_MIN_C && HW < (1l << 28), DFE_ERR_UNSUPPORTED);
  if (reinterpret_cast<uintptr_t>(map) & 15) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const WfgWs w = wfg_layout(map, B, HW);
  dim3 gw(static_cast<unsigned>((HW + 63) / 64), (C + WF_CK - 1) / WF_CK, B);
  if (HW <= WFG_SMALL_HW && getenv("DFE_WFG_MAP_LARGE") == nullptr) {          // levels 4-6: the map in one launch beside the warp
    k_warp_flow_fwd<<<gw, 64, 0, st>>>(c2, flow, warped, C, H, W, 0, align_corners, nullptr);
    DFE_LAUNCH_CHECK();
    k_wfg_map_small<<<B, WFG_SMALL_HW, 0, st>>>(flow, w.header, w.off, w.ent, H, W, 0, align_corners);
    DFE_LAUNCH_CHECK();
  } else {
    if (hipMemsetAsync(map, 0, wfg_head_bytes(B, HW), st) != hipSuccess) return DFE_ERR_LAUNCH;     // the bound's word and the counters
    k_warp_flow_fwd<<<gw, 64, 0, st>>>(c2, flow, warped, C, H, W, 0, align_corners, w.cnt);
    DFE_LAUNCH_CHECK();
    k_wfg_scan<<<B, 1024, 0, st>>>(w.cnt, w.off, static_cast<int>(HW));
    DFE_LAUNCH_CHECK();
    k_wfg_fill<<<dim3(static_cast<unsigned>((HW + 255) / 256), B), 256, 0, st>>>(flow, w.cnt, w.off, w.ent, H, W, 0, align_corners);
    DFE_LAUNCH_CHECK();
  }
  { const int rc = launch_corr_fwd(c1, warped, x, xbs, x + static_cast<long>(CR_K) * CR_K * HW, flow, B, C, H, W, st); if (rc != DFE_OK) return rc; }
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_pwc_level_bwd_map(const float* c1, const float* c2, const float* flow, const float* warped, const float* gx,
                          float* g_warped, float* g_c1, float* g_c2, void* map, float* g_flow, int B, int C, int H, int W,
                          int align_corners, void* stream) {
  DFE_REQUIRE(c1 && c2 && flow && warped && gx && g_warped && g_c1 && map, DFE_ERR_NULL);
  DFE_REQUIRE(g_c2 || g_flow, DFE_ERR_NULL);
  DFE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && B <= 65535, DFE_ERR_DIMS);
  const long HW = static_cast<long>(H) * W, xbs = static_cast<long>(dfe_pwc_level_channels(C)) * HW;
  DFE_REQUIRE(C >= WFG_MIN_C && HW < (1l << 28), DFE_ERR_UNSUPPORTED);
  if (reinterpret_cast<uintptr_t>(map) & 15) return DFE_ERR_DIMS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* gx_c1 = gx + static_cast<long>(CR_K) * CR_K * HW;
  const float* gx_flow = gx_c1 + static_cast<long>(C) * HW;
  const WfgWs w = wfg_layout(map, B, HW);
  { const int rc = launch_corr_bwd(c1, warped, gx, xbs, gx_c1, xbs, g_c1, g_warped, g_c2 ? w.header : nullptr, B, C, H, W, st); if (rc != DFE_OK) return rc; }
  DFE_LAUNCH_CHECK();
  if (g_flow) {
    launch_warp_flow_bwd(dim3(static_cast<unsigned>((HW + 63) / 64), 1, B), st, c2, flow, g_warped, g_flow, nullptr, gx_flow, xbs, C, H, W, 0, align_corners);
    DFE_LAUNCH_CHECK();
  }
  if (g_c2) {
    k_wfg_gather<<<dim3(static_cast<unsigned>((HW + 63) / 64), (C + WF_CK - 1) / WF_CK, B), 64, 0, st>>>(g_warped, w.off, w.ent, w.header, g_c2, C, static_cast<int>(HW));
    DFE_LAUNCH_CHECK();
  }
  return DFE_OK;
}

int dfe_resize(const float* in, float* out, int planes, int inH, int inW, int outH, int outW, int mode, void* stream) {
  DFE_REQUIRE(in && out, DFE_ERR_NULL);
  DFE_REQUIRE(planes > 0 && inH > 0 && inW > 0 && outH > 0 && outW > 0, DFE_ERR_DIMS);
  DFE_REQUIRE(mode == 0 || mode == 1, DFE_ERR_UNSUPPORTED);
  long n = static_cast<long>(planes) * outH * outW;
  k_resize<<<grid1d(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(in, out, planes, inH, inW, outH, outW, mode, 1.0f, 0);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_resize_bilinear_fwd(const float* in, float* out, int planes, int inH, int inW, int outH, int outW, float mult,
                            int pre_scale, void* stream) {
  DFE_REQUIRE(in && out, DFE_ERR_NULL);
  DFE_REQUIRE(planes > 0 && inH > 0 && inW > 0 && outH > 0 && outW > 0, DFE_ERR_DIMS);
  long n = static_cast<long>(planes) * outH * outW;
  k_resize<<<grid1d(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(in, out, planes, inH, inW, outH, outW, 0, mult, pre_scale);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

int dfe_resize_bilinear_bwd(const float* gout, float* gin, int planes, int inH, int inW, int outH, int outW, float mult,
                            int pre_scale, void* stream) {
  DFE_REQUIRE(gout && gin, DFE_ERR_NULL);
  DFE_REQUIRE(planes > 0 && inH > 0 && inW > 0 && outH > 0 && outW > 0, DFE_ERR_DIMS);
  // footprint of one input element: 2 * out / in + 4 outputs per axis must fit the register path
  DFE_REQUIRE(2L * outH <= static_cast<long>(G2_MAX - 4) * inH && 2L * outW <= static_cast<long>(G2_MAX - 4) * inW, DFE_ERR_UNSUPPORTED);
  long n = static_cast<long>(planes) * inH * inW;
  k_resize_bilinear_bwd<<<grid1d(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(gout, gin, planes, inH, inW, outH, outW, mult, pre_scale);
  DFE_LAUNCH_CHECK();
  return DFE_OK;
}

}  // extern "C"
