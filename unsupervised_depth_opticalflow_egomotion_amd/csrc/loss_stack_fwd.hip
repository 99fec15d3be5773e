// Fused loss stack, forward: dfe_geom_loss_fwd (include/dfe_hip.h).
//
// Five launches replace the ~3 270 ATen dispatches of model_geometry.py:797-951:
//   k_geom_pyramids        bilinear pyramids of the 3 frames + area pyramids of the 2 sources
//   k_geom_point_fwd       per pixel, both directions: flow warp, validity, occlusion softmax,
//                          rigid projection, recon, texture/dynamic masks, masked-L1 sums,
//                          |rigid-flow| and epipolar sums, flow consistency; writes the 1-byte
//                          mask pack and the masked warped images (stage W of SURVEY.md 8(d))
//   k_geom_ssim_fwd_roll   3x3 SSIM of (I*m, warped*m): a wave owns a 62-column strip, DPP wave shifts for the
//                          horizontal sums, a 3-row register window for the vertical ones (stage P)
//   k_geom_smooth_fwd      disparity (1st order, full-res) and flow (2nd order) smoothness
//   k_geom_reduce_fwd + k_geom_assemble_fwd   fixed-order reduction of block partials -> loss vectors + normalisers
// All (sample, scale) images are batched into each launch: scale 2 alone (13 k px) cannot fill
// 256 CUs.  Partials are reduced in a fixed order (no float atomics): bitwise reproducible.
#include "loss_stack_exact.h"
#include "dfe_camera.h"
#include <cstdint>
#include <cstdlib>

// wave footprint of the pointwise kernels: DFE_PT_TILE x (64 / DFE_PT_TILE) pixels (tile_pixel, loss_stack_exact.h); 0 = a row segment
#ifndef DFE_PT_TILE
#define DFE_PT_TILE 16
#endif

namespace dfe {

static inline long align4(long v) { return (v + 3) & ~3L; }

int geom_layout(const dfe_geom_args* a, GeomLayout* L) {
  if (!a) return DFE_ERR_NULL;
  if (a->B <= 0 || a->H < 8 || a->W < 8 || a->num_scales <= 0 || a->num_scales > DFE_MAX_SCALES) return DFE_ERR_DIMS;
  if (a->mode < 0 || a->mode > 2) return DFE_ERR_UNSUPPORTED;
  L->B = a->B; L->S = a->num_scales;
  L->off_px[0] = 0; L->blk_start[0] = 0;
  for (int s = 0; s < L->S; ++s) {
    // int(H / 2**s) as the reference computes it (model_geometry.py:70)
    L->H[s] = static_cast<int>(static_cast<double>(a->H) / static_cast<double>(1 << s));
    L->W[s] = static_cast<int>(static_cast<double>(a->W) / static_cast<double>(1 << s));
    if (L->H[s] < 3 || L->W[s] < 3) return DFE_ERR_DIMS;
    L->N[s] = L->H[s] * L->W[s];
    L->off_px[s + 1] = L->off_px[s] + L->N[s];
    L->nblk[s] = (L->N[s] + GS_BLOCK - 1) / GS_BLOCK;
    L->blk_start[s + 1] = L->blk_start[s] + L->nblk[s];
  }
  L->nblk0 = L->nblk[0];
  L->roll_start[0] = 0;
  for (int s = 0; s < L->S; ++s) {
    L->roll_strips[s] = (L->W[s] + RS_COLS - 1) / RS_COLS;
    L->roll_start[s + 1] = L->roll_start[s] + L->roll_strips[s] * ((L->H[s] + RS_ROWS - 1) / RS_ROWS);
  }
  L->fs_start[0] = 0;
  for (int s = 0; s < L->S; ++s) L->fs_start[s + 1] = L->fs_start[s] + ((L->W[s] + RS_COLS - 1) / RS_COLS) * ((L->H[s] + FS_ROWS - 1) / FS_ROWS);
  L->dsm_strips = (L->W[0] + RS_COLS - 1) / RS_COLS;
  L->dsm_units = L->dsm_strips * ((L->H[0] + DSM_ROWS - 1) / DSM_ROWS);
  L->rollb_start[0] = 0;
  for (int s = 0; s < L->S; ++s) {
    L->rollb_strips[s] = (L->W[s] + RSB_COLS - 1) / RSB_COLS;
    L->rollb_start[s + 1] = L->rollb_start[s] + L->rollb_strips[s] * ((L->H[s] + RSB_ROWS - 1) / RSB_ROWS);
  }
  const long B = L->B, S = L->S, sumN = L->off_px[S];
  const long nblk_total = L->blk_start[S];
  L->pyr_plane = B * 3 * (sumN - L->N[0]);
  long o = 0;
  L->o_cams = o; o = align4(o + B * 2 * S * static_cast<long>(sizeof(Camera) / sizeof(float)));
  L->o_epi = o; o = align4(o + B * 2 * static_cast<long>(sizeof(Epi) / sizeof(float)));
  L->o_pyr = o; o = align4(o + 3 * L->pyr_plane);
  L->o_area = o; o = align4(o + 2 * L->pyr_plane);
  L->o_mask = o; o = align4(o + (B * sumN + 3) / 4);
  L->o_yw = o; o = align4(o + 2 * B * 3 * sumN);
  L->o_wgt = o; o = align4(o + (a->mode == 2 ? 2 * B * sumN : 0));
  L->o_part = o; o = align4(o + B * nblk_total * PT_COUNT);
  L->o_spart = o; o = align4(o + B * 2 * static_cast<long>(L->roll_start[S]));
  L->o_fpart = o; o = align4(o + 2 * B * (nblk_total > L->fs_start[S] ? nblk_total : static_cast<long>(L->fs_start[S])) * 2);
  L->o_dpart = o; o = align4(o + 3 * B * static_cast<long>(L->dsm_units > L->nblk0 ? L->dsm_units : L->nblk0) * 2);
  L->o_sums = o; o = align4(o + B * S * SUM_COUNT);
  L->o_coef = o; o = align4(o + B * S * CF_COUNT);
  L->o_dsum = o; o = align4(o + 3 * B * 2);
  L->o_gw = o; o = align4(o + 2 * B * 3 * sumN);
  L->o_gup = o; o = align4(o + 3 * (S - 1) * B * static_cast<long>(L->N[0]));
  L->o_bpart = o; o = align4(o + B * nblk_total * PB_COUNT);
  L->dt = (a->mode != 2) ? (a->depth_terms & (DFE_DEPTH_TERM_SSIM | DFE_DEPTH_TERM_CONSIS)) : 0;
  L->o_yr = o; o = align4(o + ((L->dt & DFE_DEPTH_TERM_SSIM) ? 2 * B * 3 * sumN : 0));
  L->o_gyr = o; o = align4(o + ((L->dt & DFE_DEPTH_TERM_SSIM) ? 2 * B * 3 * sumN : 0));
  L->o_part2 = o; o = align4(o + (L->dt ? B * nblk_total * 2 : 0));
  L->o_spart2 = o; o = align4(o + ((L->dt & DFE_DEPTH_TERM_SSIM) ? B * 2 * static_cast<long>(L->roll_start[S]) : 0));
  L->o_sums2 = o; o = align4(o + (L->dt ? B * S * 4 : 0));
  L->o_scq = o; o = align4(o + ((L->dt & DFE_DEPTH_TERM_CONSIS) ? 16 + 4 * B * sumN : 0));
  // adjoint of the bilinear up-sampling in the disparity-smoothness backward (loss_stack_bwd.hip: k_geom_adj_rows /
  // k_geom_adj_cols).  Modes: 0 "separable" = every level >= 1 in two passes (rows, then columns); 1 "mixed" = round 3's
  // register gather for ratios >= 1/4 and the two passes below that; 2 "gather" = round 3's two gather kernels.
  // Default: mixed where levels 1-2 are exact 1/2 and 1/4 (the dense tent fast path of the register gather: 23.8 us
  // against 26.5 at B = 4, 256 x 832), separable for every other size (48.9 against 72.0 us at B = 2, 375 x 1242, S = 6;
  // 235 against 335 at B = 16; round 3's gathers: 126 / 785).  DFE_ADJ_MODE=separable|mixed|gather overrides.
  {
    bool exact = true;
    for (int s = 1; s < L->S && s <= 2; ++s) exact = exact && (L->H[s] << s) == L->H[0] && (L->W[s] << s) == L->W[0];
    L->adj_mode = exact ? 1 : 0;
    const char* m = getenv("DFE_ADJ_MODE");
    if (m) L->adj_mode = m[0] == 'g' ? 2 : (m[0] == 'm' ? 1 : 0);
  }
  L->adj_s0 = 1;
  if (L->adj_mode != 0) while (L->adj_s0 < L->S && L->H[0] <= 4 * L->H[L->adj_s0] && L->W[0] <= 4 * L->W[L->adj_s0]) ++L->adj_s0;
  L->adj_nseg = 0;
  for (int s = 0; s < DFE_MAX_SCALES; ++s) L->adj_L[s] = 1;
  for (int s = L->adj_s0; s < L->S; ++s) {
    int len = static_cast<int>(6.0 * L->H[0] / L->H[s]);           // L * (H_s / H_0) <= 6  ->  at most 8 low-res rows touched
    len = len > ADJ_LMAX ? ADJ_LMAX : (len < 1 ? 1 : len);
    L->adj_L[s] = len;
    const int nseg = (L->H[0] + len - 1) / len;
    if (nseg > L->adj_nseg) L->adj_nseg = nseg;
  }
  L->o_adjp = o; o = align4(o + 3 * (S - L->adj_s0) * B * static_cast<long>(L->adj_nseg) * ADJ_SLOTS * L->W[0]);
  L->total = o;
  return DFE_OK;
}

void tile_dev(const GeomLayout& L, GeomT* T) {
  for (int s = 0; s < L.S; ++s) {
    T->rW[s] = 1.0f / static_cast<float>(L.W[s]);
    T->dw[s] = make_divisor(static_cast<float>(L.W[s] > 1 ? L.W[s] - 1 : 1));
    T->dh[s] = make_divisor(static_cast<float>(L.H[s] > 1 ? L.H[s] - 1 : 1));
  }
}

void geom_dev(const dfe_geom_args* a, const GeomLayout& L, GeomDev* D) {
  float* ws = a->workspace;
  D->B = L.B; D->S = L.S; D->ac = a->align_corners; D->mode = a->mode; D->alpha = a->alpha; D->beta = a->beta;
  for (int s = 0; s <= L.S; ++s) { D->blk_start[s] = L.blk_start[s]; D->roll_start[s] = L.roll_start[s]; D->rollb_start[s] = L.rollb_start[s]; D->fs_start[s] = L.fs_start[s]; }
  for (int s = 0; s < L.S; ++s) {
    D->H[s] = L.H[s]; D->W[s] = L.W[s]; D->N[s] = L.N[s]; D->roll_strips[s] = L.roll_strips[s]; D->rollb_strips[s] = L.rollb_strips[s];
    const long lvl = static_cast<long>(L.B) * 3 * (L.off_px[s] - L.N[0]);   // offset of level s (>=1) in a frame's block
    for (int f = 0; f < 3; ++f) {
      D->pyr[f][s] = (s == 0) ? a->img[f] : ws + L.o_pyr + f * L.pyr_plane + lvl;
      D->disp[f][s] = a->disp[f][s];
    }
    for (int d = 0; d < 2; ++d) {
      D->area[d][s] = (s == 0) ? a->img[d == 0 ? 0 : 2] : ws + L.o_area + d * L.pyr_plane + lvl;
      D->flow[d][s] = a->flow[d][s];
    }
    D->mask[s] = reinterpret_cast<unsigned char*>(ws + L.o_mask) + static_cast<long>(L.B) * L.off_px[s];
    D->yw[s] = ws + L.o_yw + 6L * L.B * L.off_px[s];
    D->wgt[s] = ws + L.o_wgt + 2L * L.B * L.off_px[s];
    D->yr[s] = ws + L.o_yr + 6L * L.B * L.off_px[s];
  }
  D->dt = L.dt;
  D->cams = reinterpret_cast<const Camera*>(ws + L.o_cams);
  D->epi = reinterpret_cast<const Epi*>(ws + L.o_epi);
}

// ---------------------------------------------------------------------- epipolar geometry
// F = K_inv^T (E K_inv), E = [t]x R (model_geometry.py:375-378, inverse_warp.py:344-364).
__device__ inline void mat3_small_bmm(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(a[i * 3 + k], b[k * 3 + j]));
      o[i * 3 + j] = acc;
    }
}

// Epi block of one (sample, direction) from the pose vector, the camera's rotation and K^-1.
__device__ inline void make_epi(const float* v, const float* R, const float* Kinv_b, Epi& e) {
  // fp32 small-bmm arithmetic of the reference (acc = 0; acc += a*b in k order, no FMA; see dfe_camera.h)
  const float Sk[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
  float E[9], M[9], Ki[9], KiT[9];
  for (int k = 0; k < 9; ++k) Ki[k] = Kinv_b[k];
  for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) KiT[r * 3 + q] = Ki[q * 3 + r];
  mat3_small_bmm(Sk, R, E);
  mat3_small_bmm(E, Ki, M);
  mat3_small_bmm(KiT, M, e.F);
  for (int k = 0; k < 9; ++k) { e.Kinv[k] = Ki[k]; e.S[k] = Sk[k]; }
}

// The camera and epipolar blocks of a forward call in ONE launch (round 2: k_prepare_cameras 8-10 us + k_prepare_epi 4 us
// + a launch gap): one thread per (b, d, s); the thread of scale 0 also forms the (b, d) epipolar block from its rotation.
// Measured and rejected in round 3: running this job as an extra block row of k_geom_pyramids (nothing there depends on
// it) -- the double-precision trigonometry lifts that kernel from 24 to 78 VGPRs and gives every one of its waves a
// 320-byte scratch frame.
struct PrepJob { const float* pose; const float* K; const float* Kinv; Camera* cams; Epi* epi; int B, S, mode; ScaleList downs; };

__global__ void __launch_bounds__(64) k_geom_prepare(PrepJob pj) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pj.B * 2 * pj.S) return;
  const int s = idx % pj.S, bd = idx / pj.S, b = bd >> 1;
  const float* pv = pj.pose + static_cast<long>(bd) * 6;
  Camera c;
  make_camera(pv, pj.K + b * 9, pj.downs.v[s], c);
  pj.cams[idx] = c;
  if (s == 0 && pj.mode == 0) {
    Epi e;
    make_epi(pv, c.R, pj.Kinv + b * 9, e);
    pj.epi[bd] = e;
  }
}

// ---------------------------------------------------------------------- pyramids
// One thread per output pixel of one (frame, scale) job; for the left / right frames the same thread also
// produces the area (box-mean) level, whose window overlaps the bilinear taps.  Horizontally adjacent source
// pixels are fetched with dword-aligned 8-byte loads (memory-instruction count is what these kernels pay for).
struct PyrJob { const float* in; float* out_bilinear; float* out_area; int outH, outW; };
struct PyrJobs { PyrJob j[3 * (DFE_MAX_SCALES - 1)]; int n, planes, inH, inW; };

__device__ __forceinline__ void load2(const float* __restrict__ row, int x0, int x1, float& v0, float& v1) {
  if (x1 == x0 + 1) { const PairF p = *reinterpret_cast<const PairF*>(row + x0); v0 = p.a; v1 = p.b; }
  else { v0 = row[x0]; v1 = row[x1]; }
}

__global__ void k_geom_pyramids(PyrJobs jobs) {
  const PyrJob jb = jobs.j[blockIdx.y];
  const long n = static_cast<long>(jobs.planes) * jb.outH * jb.outW;
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;     // (an XCD-swizzled block order was measured: 19.5 -> 30 us)
  if (i >= n) return;
  const int ox = static_cast<int>(i % jb.outW), oy = static_cast<int>((i / jb.outW) % jb.outH);
  const long pl = i / (static_cast<long>(jb.outW) * jb.outH);
  const int inH = jobs.inH, inW = jobs.inW;
  const float* src = jb.in + pl * inH * inW;
  if (jb.out_bilinear) {
    // F.interpolate(bilinear, align_corners=False), ATen's contraction (lerp2_aten, dfe_device.h)
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bilinear_src(oy, static_cast<float>(inH) / jb.outH, inH, y0, y1, ly0, ly1);
    bilinear_src(ox, static_cast<float>(inW) / jb.outW, inW, x0, x1, lx0, lx1);
    float v00, v01, v10, v11;
    load2(src + static_cast<long>(y0) * inW, x0, x1, v00, v01);
    load2(src + static_cast<long>(y1) * inW, x0, x1, v10, v11);
    jb.out_bilinear[i] = lerp2_aten_sel(aten_small_resize(jb.outH, jb.outW), v00, v01, v10, v11, lx0, lx1, ly0, ly1);
  }
  if (jb.out_area) {
    // adaptive_avg_pool2d window [floor(o*in/out), ceil((o+1)*in/out)), row-major sequential sum / count
    const int ys = (oy * inH) / jb.outH, ye = ((oy + 1) * inH + jb.outH - 1) / jb.outH;
    const int xs = (ox * inW) / jb.outW, xe = ((ox + 1) * inW + jb.outW - 1) / jb.outW;
    float sum = 0.0f;
    for (int yy = ys; yy < ye; ++yy) {
      const float* row = src + static_cast<long>(yy) * inW;
      int xx = xs;
      for (; xx + 1 < xe; xx += 2) { const PairF p = *reinterpret_cast<const PairF*>(row + xx); sum += p.a; sum += p.b; }
      if (xx < xe) sum += row[xx];
    }
    jb.out_area[i] = sum / static_cast<float>((ye - ys) * (xe - xs));
  }
}

// Levels 1 and 2 of an exact power-of-two pyramid (H, W multiples of 4: every KITTI training size) from ONE read of the
// frame: a thread owns a 4x4 input block (four 16-byte loads per plane), and writes the 2x2 level-1 outputs and the level-2
// output it covers, bilinear and (source frames) area.  The per-(frame, scale) jobs of k_geom_pyramids read the frame once
// per scale and once more for the level-2 box mean (PMC: 1.5x the algorithmic bytes).  Same taps, weights and association
// orders as the generic kernel (bilinear_src gives i0 = 2o, l = 0.5 at 1/2 and i0 = 4o + 1, l = 0.5 at 1/4; box means are
// row-major sequential sums): bit-identical outputs.  grid: x = blocks over planes * (H/4) * (W/4), y = frame.
struct Pyr12Job { const float* in[3]; float* b1[3]; float* b2[3]; float* a1[3]; float* a2[3]; int planes, H, W; };

__global__ void __launch_bounds__(256) k_geom_pyramids12(Pyr12Job jb) {
  const int f = blockIdx.y;
  const int H2 = jb.H / 4, W2 = jb.W / 4, H1 = jb.H / 2, W1 = jb.W / 2;
  const long n = static_cast<long>(jb.planes) * H2 * W2;
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ox = static_cast<int>(i % W2), oy = static_cast<int>((i / W2) % H2);
  const long pl = i / (static_cast<long>(W2) * H2);
  const float* src = jb.in[f] + pl * jb.H * jb.W + static_cast<long>(4 * oy) * jb.W + 4 * ox;
  float v[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 q = *reinterpret_cast<const float4*>(src + static_cast<long>(r) * jb.W);
    v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w;
  }
  const bool small1 = aten_small_resize(H1, W1), small2 = aten_small_resize(H2, W2);
  float* o1 = jb.b1[f] + pl * H1 * W1 + static_cast<long>(2 * oy) * W1 + 2 * ox;
  float* a1 = jb.a1[f] ? jb.a1[f] + pl * H1 * W1 + static_cast<long>(2 * oy) * W1 + 2 * ox : nullptr;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float bl[2], ar[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float v00 = v[2 * r][2 * c], v01 = v[2 * r][2 * c + 1], v10 = v[2 * r + 1][2 * c], v11 = v[2 * r + 1][2 * c + 1];
      bl[c] = lerp2_aten_sel(small1, v00, v01, v10, v11, 0.5f, 0.5f, 0.5f, 0.5f);
      float sum = 0.0f;
      sum += v00; sum += v01; sum += v10; sum += v11;
      ar[c] = sum / 4.0f;
    }
    *reinterpret_cast<PairF*>(o1 + static_cast<long>(r) * W1) = PairF{bl[0], bl[1]};
    if (a1) *reinterpret_cast<PairF*>(a1 + static_cast<long>(r) * W1) = PairF{ar[0], ar[1]};
  }
  jb.b2[f][pl * H2 * W2 + static_cast<long>(oy) * W2 + ox] = lerp2_aten_sel(small2, v[1][1], v[1][2], v[2][1], v[2][2], 0.5f, 0.5f, 0.5f, 0.5f);
  if (jb.a2[f]) {
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += v[r][c];
    jb.a2[f][pl * H2 * W2 + static_cast<long>(oy) * W2 + ox] = sum / 16.0f;
  }
}

// Box means with large windows (non-divisible sizes at coarse scales: up to (2^s + 1)^2 inputs per output).  A
// thread-per-output loop pays one memory round trip per pair of inputs; here one wave stages the window in LDS
// with row-coalesced loads and lane 0 adds it in ATen's row-major order (the mean stays bit-identical to
// adaptive_avg_pool2d's).  grid.x = output elements, grid.y = job; block = one wave.
constexpr int AREA_LDS = 1280;
constexpr int AREA_COARSE_MIN = 256;   // windows above this many inputs take the wave path

__global__ void __launch_bounds__(64) k_geom_area_coarse(PyrJobs jobs) {
  __shared__ __attribute__((aligned(16))) float buf[AREA_LDS];
  const PyrJob jb = jobs.j[blockIdx.y];
  const long n = static_cast<long>(jobs.planes) * jb.outH * jb.outW;
  const long i = blockIdx.x;
  if (i >= n) return;
  const int lane = threadIdx.x;
  const int ox = static_cast<int>(i % jb.outW), oy = static_cast<int>((i / jb.outW) % jb.outH);
  const long pl = i / (static_cast<long>(jb.outW) * jb.outH);
  const int inH = jobs.inH, inW = jobs.inW;
  const float* src = jb.in + pl * inH * inW;
  const int ys = (oy * inH) / jb.outH, ye = ((oy + 1) * inH + jb.outH - 1) / jb.outH;
  const int xs = (ox * inW) / jb.outW, xe = ((ox + 1) * inW + jb.outW - 1) / jb.outW;
  const int kw = xe - xs;
  float sum = 0.0f;
  if (kw > AREA_LDS) {        // wider than the staging buffer (W > 2048 * W_s): plain ordered loop
    if (lane == 0)
      for (int yy = ys; yy < ye; ++yy)
        for (int xx = xs; xx < xe; ++xx) sum += src[static_cast<long>(yy) * inW + xx];
  } else {
    const int rows_per = AREA_LDS / kw;
    for (int y0 = ys; y0 < ye; y0 += rows_per) {
      const int nr = min(rows_per, ye - y0);
      for (int c = lane; c < kw; c += 64) {
        const float* col = src + static_cast<long>(y0) * inW + xs + c;
        for (int r = 0; r < nr; r += 8) {       // eight independent row loads in flight
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = col[static_cast<long>(min(r + u, nr - 1)) * inW];
#pragma unroll
          for (int u = 0; u < 8; ++u) if (r + u < nr) buf[(r + u) * kw + c] = v[u];
        }
      }
      __syncthreads();
      if (lane == 0) {
        const int cnt = nr * kw;
        int k = 0;
        for (; k + 16 <= cnt; k += 16) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = buf[k + u];
#pragma unroll
          for (int u = 0; u < 16; ++u) sum += v[u];
        }
        for (; k < cnt; ++k) sum += buf[k];
      }
      __syncthreads();
    }
  }
  if (lane == 0) jb.out_area[i] = sum / static_cast<float>((ye - ys) * kw);
}

// ---------------------------------------------------------------------- pointwise forward
// Bound by instruction issue: ~860 VALU and 45 vector-memory instructions per pixel (24 of them the bilinear
// gathers), texture addresser and VALU each ~55 % busy (profiles/r02_pmc_point_kernels.md).  What the measurements
// selected: one pixel per thread in 256-thread blocks, one dword-aligned 8-byte load per footprint row, correctly
// rounded 3/5-instruction division / square-root sequences (loss_stack_exact.h), host-side divisors, integer bounds
// tests and 32-bit byte offsets from block-uniform bases.  Measured and rejected in round 2:
// RGBA-texel zero-bordered source planes with wave-persistent accumulation (fewer, wider gathers: the addresser cost
// follows bytes, VGPRs double -> 54-60 us vs 48), 32x8 LDS-staged source tiles with a +-6 px halo around the
// tile-centre flow (hit rate too low on rough flow fields: 59 us), LDS-transposed streamed loads / stores (53 us).
// Round 3 (profiles/r03_point_fwd_ablation.md): what the 24 gathers cost is set by how many 128-byte lines a wave's
// addresses touch (20 us of 50 with per-pixel random flows, 0.7 us with coherent ones), not by their count.  Two adjacent
// pixels per thread with 8-byte streamed accesses (21 -> 10.5 streamed vector-memory instructions per pixel) was measured
// and rejected: the two interleaved chains need 155 VGPRs (3 waves per SIMD): 59.9 vs 49.8 us in the loss_stack loop,
// 40-41 vs 37.7 us inside the train step.
// DFE_ABL: compile-time ablation switches of k_geom_point_fwd for the cost study of profiles/r03_point_fwd_ablation.md
// (tools/ablate_point_fwd.sh builds one library per value; 0 = the shipped kernel, no code depends on it then).
//   1 no block reductions   2 gathers at the pixel's own position (coherent)   4 no stores   8 rigid branch without gathers
//  16 flow-warp branch without gathers   32 no epipolar / flow-consistency terms   64 no projection arithmetic
// 128 no streamed source-pyramid loads
#ifndef DFE_ABL
#define DFE_ABL 0
#endif

struct PointCtx {
  int b, s, H, W, ac;
  unsigned N4, p4;
  float alpha, beta;
  const float *srcL, *srcR, *areaL, *areaR;   // block-uniform plane bases of this sample
  const float *dispL, *dispR;                  // DT: the source frames' disparity planes of this sample and scale
  const Camera* cam;                           // cams[(b*2+0)*S + s], direction stride S
  int cam_stride;
  const Epi* epi;                              // epi[b*2]
  Divisor dw, dh;
};

struct PixIn { float i0, i1, i2, fu[2], fv[2], dsp, sl[3], sr[3]; };

// DT: also the two depth terms the reference keeps commented (model_geometry.py:889-891,897-899): yr = rigid
// reconstruction x texture-gated mask (for the SSIM launch over it) and dc = clamp(|cd - pd| / |cd + pd|, 0, 1) x mask
// with cd = the projection's Z and pd = the source disparity sampled at the rigid coordinate, clamp(min=1e-3)
// (inverse_warp.py:259-262; the reference passes its disparities as "depth", model_geometry.py:807-810).
template <bool DT>
__device__ __forceinline__ void point_pixel(const PointCtx& c, int px, int py, const PixIn& in, float (&yw)[2][3],
                                            unsigned& bits, float (&acc)[PT_COUNT], float (&yr)[2][3], float (&dc)[2]) {
  const int H = c.H, W = c.W;
  float wv[2][3], dif[2];
  bool valid[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    float ix, iy;
    flow_coords_d(px, py, in.fu[d], in.fv[d], H, W, c.ac, c.dw, c.dh, ix, iy);
    FastTap t = make_fast_tap(ix, iy, H, W);
    if (DFE_ABL & 2) { t.o0 = min(c.p4, c.N4 - 8u); t.o1 = t.o0; }
    const float keep = (fast_cover(t) < 0.9999f) ? 0.0f : 1.0f;
    const float* src = d == 0 ? c.srcL : c.srcR;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (DFE_ABL & 16) wv[d][ch] = (d == 0 ? in.sl[ch] : in.sr[ch]) * (keep + t.wa0);
      else wv[d][ch] = fast_sample(reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + ch * c.N4), t) * keep;
    }
    valid[d] = !(wv[d][0] == 0.0f && wv[d][1] == 0.0f && wv[d][2] == 0.0f);
    dif[d] = mean3_abs_diff(in.i0, in.i1, in.i2, wv[d][0], wv[d][1], wv[d][2]);
  }
  bool occ[2];
  occ_decide(dif[0], dif[1], occ[0], occ[1]);
  bits = 0;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const Camera& cam = c.cam[d * c.cam_stride];
    Proj pr;
    if (DFE_ABL & 64) { pr.U = static_cast<float>(px) + in.dsp; pr.V = static_cast<float>(py) + in.dsp * cam.b[0]; pr.Z = 1.0f; }
    else pr = project_fast(cam, px, py, in.dsp);
    const float ru = pr.U - static_cast<float>(px), rv = pr.V - static_cast<float>(py);
    const float du = fabsf(ru - in.fu[d]), dv = fabsf(rv - in.fv[d]);
    const bool dyna = dyna_decide(in.fu[d], in.fv[d], ru, rv, du, dv, c.alpha, c.beta);
    float xn, yn; bool lx, ly;
    rigid_grid_d(pr, c.dw, c.dh, xn, yn, lx, ly);
    FastTap t = make_fast_tap(unnormalize(xn, W, c.ac), unnormalize(yn, H, c.ac), H, W);
    if (DFE_ABL & 2) { t.o0 = min(c.p4, c.N4 - 8u); t.o1 = t.o0; }
    const float* ar = d == 0 ? c.areaL : c.areaR;
    const float* sp = d == 0 ? in.sl : in.sr;
    float rec[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (DFE_ABL & 8) rec[ch] = sp[ch] * (t.wa0 + t.wb1);
      else rec[ch] = fast_sample(reinterpret_cast<const float*>(reinterpret_cast<const char*>(ar) + ch * c.N4), t);
    }
    const float e_rec = mean3_abs_diff(in.i0, in.i1, in.i2, rec[0], rec[1], rec[2]);
    const float e_src = mean3_abs_diff(in.i0, in.i1, in.i2, sp[0], sp[1], sp[2]);
    const bool tex = e_rec < e_src;
    const float vo = (valid[d] && occ[d]) ? 1.0f : 0.0f;
    const float m_rig = dyna ? vo : 0.0f, m_dyn = dyna ? 0.0f : vo;
    const float m_tex = tex ? m_rig : 0.0f;
    if (DT) {
      yr[d][0] = rec[0] * m_tex; yr[d][1] = rec[1] * m_tex; yr[d][2] = rec[2] * m_tex;
      const float v = fast_sample(d == 0 ? c.dispL : c.dispR, t);
      const float pd = (v >= 1e-3f || v != v) ? v : 1e-3f;
      const float q = fabsf(pr.Z - pd) / fabsf(pr.Z + pd);
      dc[d] = fminf(fmaxf(q, 0.0f), 1.0f) * m_tex;
    }
    const float l1_rec = (fabsf(in.i0 - rec[0]) + fabsf(in.i1 - rec[1])) + fabsf(in.i2 - rec[2]);
    const float l1_wrp = (fabsf(in.i0 - wv[d][0]) + fabsf(in.i1 - wv[d][1])) + fabsf(in.i2 - wv[d][2]);
    float* a = acc + d * PT_PER_DIR;
    a[PT_M_TEX] += m_tex;       a[PT_L1_DEPTH] += l1_rec * m_tex;
    a[PT_M_RIG] += m_rig;       a[PT_L1_RIG] += l1_wrp * m_rig;
    a[PT_M_DYN] += m_dyn;       a[PT_L1_DYN] += l1_wrp * m_dyn;
    a[PT_M_VO] += vo;
    if (c.s == 0 && !(DFE_ABL & 32)) {
      a[PT_FDIFF] += (du + dv) * m_rig;
      const Epi& e = c.epi[d];
      const float x1 = static_cast<float>(px), y1 = static_cast<float>(py);
      // f_mat.bmm(p1) is a large bmm: fma(F2, 1, fma(F1, y, F0 * x)) (MKL sgemm order, see project())
      const float l0 = __fmaf_rn(e.F[1], y1, e.F[0] * x1) + e.F[2];
      const float l1 = __fmaf_rn(e.F[4], y1, e.F[3] * x1) + e.F[5];
      const float l2 = __fmaf_rn(e.F[7], y1, e.F[6] * x1) + e.F[8];
      // feeds a loss value only (no mask): 1-ulp sqrt / reciprocal instead of the IEEE sequences
      const float div = __builtin_amdgcn_sqrtf(l0 * l0 + l1 * l1) + 1e-6f;
      a[PT_EPI] += fabsf(((x1 + in.fu[d]) * l0 + (y1 + in.fv[d]) * l1) + l2) * __builtin_amdgcn_rcpf(div);
    }
    bits |= (valid[d] ? (DFE_MASK_VALID_BWD << d) : 0u) | (occ[d] ? (DFE_MASK_OCC_BWD << d) : 0u) |
            (dyna ? (DFE_MASK_DYNA_BWD << d) : 0u) | (tex ? (DFE_MASK_TEX_BWD << d) : 0u);
    yw[d][0] = wv[d][0] * vo; yw[d][1] = wv[d][1] * vo; yw[d][2] = wv[d][2] * vo;
  }
  // flow consistency (model_geometry.py:195-210): |unit(fwd) + unit(bwd)| on (1 - occ_fwd)
  // loss-only: 1-ulp sqrt / reciprocal
  if (DFE_ABL & 32) return;
  const float rf = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(in.fu[1] * in.fu[1] + in.fv[1] * in.fv[1]) + 1e-12f);
  const float rb = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(in.fu[0] * in.fu[0] + in.fv[0] * in.fv[0]) + 1e-12f);
  const float inv = occ[1] ? 0.0f : 1.0f;
  acc[PT_INV] += inv;
  acc[PT_CONSIS] += (fabsf(in.fu[1] * rf + in.fu[0] * rb) + fabsf(in.fv[1] * rf + in.fv[0] * rb)) * inv;
}

// Block sums of k_geom_point_fwd's PT_COUNT per-thread values (one pixel per thread).  Nine of them are {0,1} mask
// indicators: their sums are wave population counts (v_cmp + s_bcnt1, exact) instead of float butterflies; the eleven
// float entries take the DPP tree of block_sum.  smem: 11 * 4 * nwaves floats + 9 * nwaves counts.
__device__ __forceinline__ void point_block_sums(float (&acc)[PT_COUNT], float* smem, float* out) {
  constexpr int NF = 11, NC = 9;
  constexpr int FI[NF] = {PT_L1_DEPTH, PT_L1_RIG, PT_L1_DYN, PT_FDIFF, PT_EPI,
                          PT_PER_DIR + PT_L1_DEPTH, PT_PER_DIR + PT_L1_RIG, PT_PER_DIR + PT_L1_DYN, PT_PER_DIR + PT_FDIFF,
                          PT_PER_DIR + PT_EPI, PT_CONSIS};
  constexpr int CI[NC] = {PT_M_TEX, PT_M_RIG, PT_M_DYN, PT_M_VO, PT_PER_DIR + PT_M_TEX, PT_PER_DIR + PT_M_RIG,
                          PT_PER_DIR + PT_M_DYN, PT_PER_DIR + PT_M_VO, PT_INV};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = GS_BLOCK / 64;
  float f[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) f[i] = acc[FI[i]];
#pragma unroll
  for (int i = 0; i < NF; ++i) f[i] = dpp_add<0xB1>(f[i]);
#pragma unroll
  for (int i = 0; i < NF; ++i) f[i] = dpp_add<0x4E>(f[i]);
#pragma unroll
  for (int i = 0; i < NF; ++i) f[i] = dpp_add<0x141>(f[i]);
#pragma unroll
  for (int i = 0; i < NF; ++i) f[i] = dpp_add<0x140>(f[i]);
  if ((lane & 15) == 0) {
    const int slot = wave * 4 + (lane >> 4);
#pragma unroll
    for (int i = 0; i < NF; ++i) smem[slot * NF + i] = f[i];
  }
  float* cnt = smem + NF * 4 * NW;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const unsigned long long m = __ballot(acc[CI[i]] != 0.0f);
    if (lane == 0) cnt[wave * NC + i] = static_cast<float>(__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < NF) {
    float sum = 0.0f;
    for (int w = 0; w < 4 * NW; ++w) sum += smem[w * NF + threadIdx.x];
    int dst = 0;
#pragma unroll
    for (int i = 0; i < NF; ++i) dst = (static_cast<int>(threadIdx.x) == i) ? FI[i] : dst;
    out[dst] = sum;
  } else if (threadIdx.x < NF + NC) {
    const int t = threadIdx.x - NF;
    float sum = 0.0f;
    for (int w = 0; w < NW; ++w) sum += cnt[w * NC + t];
    int dst = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) dst = (t == i) ? CI[i] : dst;
    out[dst] = sum;
  }
  __syncthreads();
}

template <bool DT>
// Register budget: left alone the compiler squeezes the kernel into 66 VGPRs (7 waves per SIMD); with the budget of 5
// waves it takes 75 (6 waves) and schedules the loads further ahead: 51.8 -> 48.3 us in the loss_stack loop, 0.326 ->
// 0.330 of the roofline inside the train step (round 3).  8 waves (64 VGPRs) spill: 62.8 us.
#ifndef DFE_PT_WPE
#define DFE_PT_WPE 5
#endif
#define DFE_PT_ATTR __attribute__((amdgpu_waves_per_eu(DFE_PT_WPE, DFE_PT_WPE)))
__global__ void __launch_bounds__(GS_BLOCK) DFE_PT_ATTR k_geom_point_fwd(GeomDev D, GeomT T, float* __restrict__ part, float* __restrict__ part2) {
  __shared__ float red[PT_COUNT * 4 * (GS_BLOCK / 64)];
  const unsigned nblk_total = D.blk_start[D.S];
  const unsigned blk = xcd_swizzle(blockIdx.x, nblk_total);
  const int b = blockIdx.y;
  const int s = find_scale(D.blk_start, D.S, blk);
  const int H = D.H[s], W = D.W[s], N = D.N[s];
  const unsigned p = (blk - D.blk_start[s]) * GS_BLOCK + threadIdx.x;
  float acc[PT_COUNT];
#pragma unroll
  for (int i = 0; i < PT_COUNT; ++i) acc[i] = 0.0f;
  float dc[2] = {0.0f, 0.0f};
  if (p < static_cast<unsigned>(N)) {
    unsigned px, py;
    tile_pixel<DFE_PT_TILE>(p, W, H, T.rW[s], px, py);      // a wave covers a 16 x 4 pixel tile (loss_stack_exact.h)
    const unsigned pm = py * static_cast<unsigned>(W) + px;   // the pixel this thread computes (index in memory)
    const unsigned p4 = pm * 4u, N4 = static_cast<unsigned>(N) * 4u;
    PointCtx c;
    c.b = b; c.s = s; c.H = H; c.W = W; c.ac = D.ac; c.N4 = N4; c.p4 = p4; c.alpha = D.alpha; c.beta = D.beta;
    c.srcL = D.pyr[0][s] + static_cast<long>(b) * 3 * N; c.srcR = D.pyr[2][s] + static_cast<long>(b) * 3 * N;
    c.areaL = D.area[0][s] + static_cast<long>(b) * 3 * N; c.areaR = D.area[1][s] + static_cast<long>(b) * 3 * N;
    c.cam = D.cams + (b * 2) * D.S + s; c.cam_stride = D.S; c.epi = D.epi + b * 2;
    c.dw = T.dw[s]; c.dh = T.dh[s];
    c.dispL = D.disp[0][s] + static_cast<long>(b) * N; c.dispR = D.disp[2][s] + static_cast<long>(b) * N;
    const float* it = D.pyr[1][s] + static_cast<long>(b) * 3 * N;
    const float* flb = D.flow[0][s] + static_cast<long>(b) * 2 * N;
    const float* flf = D.flow[1][s] + static_cast<long>(b) * 2 * N;
    PixIn in;
    in.i0 = ldb(it, p4); in.i1 = ldb(it, p4 + N4); in.i2 = ldb(it, p4 + 2 * N4);
    in.fu[0] = ldb(flb, p4); in.fv[0] = ldb(flb, p4 + N4); in.fu[1] = ldb(flf, p4); in.fv[1] = ldb(flf, p4 + N4);
    in.dsp = ldb(D.disp[1][s] + static_cast<long>(b) * N, p4);
    if (DFE_ABL & 128) { in.sl[0] = in.i1; in.sl[1] = in.i2; in.sl[2] = in.i0; in.sr[0] = in.i2; in.sr[1] = in.i0; in.sr[2] = in.i1; }
    else {
      in.sl[0] = ldb(c.srcL, p4); in.sl[1] = ldb(c.srcL, p4 + N4); in.sl[2] = ldb(c.srcL, p4 + 2 * N4);
      in.sr[0] = ldb(c.srcR, p4); in.sr[1] = ldb(c.srcR, p4 + N4); in.sr[2] = ldb(c.srcR, p4 + 2 * N4);
    }
    float yw[2][3], yr[2][3];
    unsigned bits;
    point_pixel<DT>(c, static_cast<int>(px), static_cast<int>(py), in, yw, bits, acc, yr, dc);
    if (DFE_ABL & 4) { if (yw[0][0] + yw[0][1] + yw[0][2] + yw[1][0] + yw[1][1] + yw[1][2] == 1234.5f) D.mask[s][0] = 1; }
    else
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      float* ywp = D.yw[s] + (static_cast<long>(d) * D.B + b) * 3 * N;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) stb(ywp, p4 + ch * N4, yw[d][ch]);
      if (DT && (D.dt & DFE_DEPTH_TERM_SSIM)) {
        float* yrp = D.yr[s] + (static_cast<long>(d) * D.B + b) * 3 * N;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) stb(yrp, p4 + ch * N4, yr[d][ch]);
      }
    }
    if (!(DFE_ABL & 4)) (D.mask[s] + static_cast<long>(b) * N)[pm] = static_cast<unsigned char>(bits);
    else if (bits == 0x12345u) D.mask[s][1] = 1;
  }
  if (DFE_ABL & 1) {
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < PT_COUNT; ++i) tot += acc[i];
    if (tot == 1234.5f) part[0] = tot;
    return;
  }
  point_block_sums(acc, red, part + (static_cast<long>(b) * nblk_total + blk) * PT_COUNT);
  if (DT) block_sum<2>(dc, red, part2 + (static_cast<long>(b) * nblk_total + blk) * 2);
}

// ---------------------------------------------------------------------- depth-only pointwise forward
// Model_depth loss stack (model_depth.py:296-323): rigid recon of both sources, mask = inverse_warp2
// validity * texture mask, masked-L1 sums.  One pixel per thread over the 1-px block table.
// DT: + the commented terms of Model_depth.forward (model_depth.py:326-335): SSIM over the rigid reconstructions on the
// validity x texture mask, and compute_consis_loss (model_depth.py:154-163) -- here WITHOUT a mask (plain mean).
template <bool DT>
__global__ void __launch_bounds__(GS_BLOCK) k_depth_point_fwd(GeomDev D, float* __restrict__ part, float* __restrict__ part2) {
  __shared__ float red[PT_COUNT * 4 * (GS_BLOCK / 64)];
  const unsigned nblk_total = D.blk_start[D.S];
  const unsigned blk = xcd_swizzle(blockIdx.x, nblk_total);
  const int b = blockIdx.y;
  const int s = find_scale(D.blk_start, D.S, blk);
  const int H = D.H[s], W = D.W[s], N = D.N[s];
  const unsigned pl = (blk - D.blk_start[s]) * GS_BLOCK + threadIdx.x;
  float acc[PT_COUNT];
#pragma unroll
  for (int i = 0; i < PT_COUNT; ++i) acc[i] = 0.0f;
  float dc[2] = {0.0f, 0.0f};
  if (pl < static_cast<unsigned>(N)) {
    unsigned px, py;
    tile_pixel<DFE_PT_TILE>(pl, W, H, px, py);               // wave footprint: loss_stack_exact.h
    const unsigned p = py * static_cast<unsigned>(W) + px;
    const unsigned p4 = p * 4u, N4 = static_cast<unsigned>(N) * 4u;
    const float* it = D.pyr[1][s] + static_cast<long>(b) * 3 * N;
    const float i0 = ldb(it, p4), i1 = ldb(it, p4 + N4), i2 = ldb(it, p4 + 2 * N4);
    const float dsp = ldb(D.disp[1][s] + static_cast<long>(b) * N, p4);
    const Divisor dw = make_divisor(static_cast<float>(W - 1)), dh = make_divisor(static_cast<float>(H - 1));
    unsigned bits = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const Camera& cam = D.cams[(b * 2 + d) * D.S + s];
      const Proj pr = project(cam, px, py, dsp);
      float xn, yn; bool lx, ly;
      rigid_grid_d(pr, dw, dh, xn, yn, lx, ly);
      const bool valid = fmaxf(fabsf(xn), fabsf(yn)) <= 1.0f;
      const FastTap t = make_fast_tap(unnormalize(xn, W, D.ac), unnormalize(yn, H, D.ac), H, W);
      const float* ar = D.area[d][s] + static_cast<long>(b) * 3 * N;
      float rec[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) rec[ch] = fast_sample(reinterpret_cast<const float*>(reinterpret_cast<const char*>(ar) + ch * N4), t);
      const float* sp = D.pyr[d == 0 ? 0 : 2][s] + static_cast<long>(b) * 3 * N;
      const bool tex = mean3_abs_diff(i0, i1, i2, rec[0], rec[1], rec[2]) < mean3_abs_diff(i0, i1, i2, ldb(sp, p4), ldb(sp, p4 + N4), ldb(sp, p4 + 2 * N4));
      const float m = (valid && tex) ? 1.0f : 0.0f;
      acc[d * PT_PER_DIR + PT_M_TEX] = m;
      acc[d * PT_PER_DIR + PT_L1_DEPTH] = ((fabsf(i0 - rec[0]) + fabsf(i1 - rec[1])) + fabsf(i2 - rec[2])) * m;
      bits |= (valid ? (DFE_MASK_VALID_BWD << d) : 0u) | (tex ? (DFE_MASK_TEX_BWD << d) : 0u);
      if (DT) {
        if (D.dt & DFE_DEPTH_TERM_SSIM) {
          float* yrp = D.yr[s] + (static_cast<long>(d) * D.B + b) * 3 * N;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) stb(yrp, p4 + ch * N4, rec[ch] * m);
        }
        const float v = fast_sample(D.disp[d == 0 ? 0 : 2][s] + static_cast<long>(b) * N, t);
        const float pd = (v >= 1e-3f || v != v) ? v : 1e-3f;
        dc[d] = fminf(fmaxf(fabsf(pr.Z - pd) / fabsf(pr.Z + pd), 0.0f), 1.0f);
      }
    }
    (D.mask[s] + static_cast<long>(b) * N)[p] = static_cast<unsigned char>(bits);
  }
  block_sum<PT_COUNT>(acc, red, part + (static_cast<long>(b) * nblk_total + blk) * PT_COUNT);
  if (DT) block_sum<2>(dc, red, part2 + (static_cast<long>(b) * nblk_total + blk) * 2);
}

// ---------------------------------------------------------------------- flow-only pointwise forward
// Model_flow loss stack (model_flow.py:105-138,209-255): flow warps of the box-mean pyramids, validity, soft
// Gaussian occlusion weights 2 exp(-(w - 0.5)^2 / 0.03) * valid (detached), weighted 1-channel L1 sums, flow
// consistency on (1 - weight_fwd).  Writes the float weights and the weighted warped images.
__global__ void __launch_bounds__(GS_BLOCK) k_flow_point_fwd(GeomDev D, float* __restrict__ part) {
  __shared__ float red[PT_COUNT * 4 * (GS_BLOCK / 64)];
  const unsigned nblk_total = D.blk_start[D.S];
  const unsigned blk = xcd_swizzle(blockIdx.x, nblk_total);
  const int b = blockIdx.y;
  const int s = find_scale(D.blk_start, D.S, blk);
  const int H = D.H[s], W = D.W[s], N = D.N[s];
  const unsigned pl = (blk - D.blk_start[s]) * GS_BLOCK + threadIdx.x;
  float acc[PT_COUNT];
#pragma unroll
  for (int i = 0; i < PT_COUNT; ++i) acc[i] = 0.0f;
  if (pl < static_cast<unsigned>(N)) {
    unsigned px, py;
    tile_pixel<DFE_PT_TILE>(pl, W, H, px, py);               // wave footprint: loss_stack_exact.h
    const unsigned p = py * static_cast<unsigned>(W) + px;
    const unsigned p4 = p * 4u, N4 = static_cast<unsigned>(N) * 4u;
    const float* it = D.pyr[1][s] + static_cast<long>(b) * 3 * N;
    const float i0 = ldb(it, p4), i1 = ldb(it, p4 + N4), i2 = ldb(it, p4 + 2 * N4);
    const Divisor dw = make_divisor(static_cast<float>(W > 1 ? W - 1 : 1)), dh = make_divisor(static_cast<float>(H > 1 ? H - 1 : 1));
    float fu[2], fv[2], wv[2][3], dif[2], vld[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float* fl = D.flow[d][s] + static_cast<long>(b) * 2 * N;
      fu[d] = ldb(fl, p4); fv[d] = ldb(fl, p4 + N4);
      float ix, iy;
      flow_coords_d(px, py, fu[d], fv[d], H, W, D.ac, dw, dh, ix, iy);
      const FastTap t = make_fast_tap(ix, iy, H, W);
      const float keep = (fast_cover(t) < 0.9999f) ? 0.0f : 1.0f;
      const float* src = D.pyr[d == 0 ? 0 : 2][s] + static_cast<long>(b) * 3 * N;
#pragma unroll
      for (int c = 0; c < 3; ++c) wv[d][c] = fast_sample(reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + c * N4), t) * keep;
      vld[d] = (wv[d][0] == 0.0f && wv[d][1] == 0.0f && wv[d][2] == 0.0f) ? 0.0f : 1.0f;
      dif[d] = mean3_abs_diff(i0, i1, i2, wv[d][0], wv[d][1], wv[d][2]);
    }
    float sw[2];
    occ_weights(dif[0], dif[1], sw[0], sw[1]);
    float wgt[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float c = sw[d] - 0.5f;
      wgt[d] = (2.0f * expf(-(c * c) / 0.03f)) * vld[d];
      acc[d * PT_PER_DIR + PT_M_VO] = wgt[d];
      acc[d * PT_PER_DIR + PT_L1_RIG] = dif[d] * wgt[d];
      (D.wgt[s] + (static_cast<long>(d) * D.B + b) * N)[p] = wgt[d];
      float* yw = D.yw[s] + (static_cast<long>(d) * D.B + b) * 3 * N;
      stb(yw, p4, wv[d][0] * wgt[d]); stb(yw, p4 + N4, wv[d][1] * wgt[d]); stb(yw, p4 + 2 * N4, wv[d][2] * wgt[d]);
    }
    const float nf = l2norm2(fu[1], fv[1]), nb = l2norm2(fu[0], fv[0]);
    const float inv = 1.0f - wgt[1];
    acc[PT_INV] = inv;
    acc[PT_CONSIS] = (fabsf(fu[1] / nf + fu[0] / nb) + fabsf(fv[1] / nf + fv[0] / nb)) * inv;
  }
  block_sum<PT_COUNT>(acc, red, part + (static_cast<long>(b) * nblk_total + blk) * PT_COUNT);
}

// ---------------------------------------------------------------------- SSIM forward (stage P), rolling window
// One wave owns a strip of 62 columns (lanes 1..62; lanes 0 and 63 are the halo) and marches down RS_ROWS
// rows (+1 halo row on each side).  Horizontal 3-sums come from DPP wave shifts, the vertical 3-row window
// lives in registers: no LDS, no barriers, and the loads of the next row are independent of the arithmetic of
// the current one.  grid: x = units (strip x row block) over all scales, y = b*2 + d; block = one wave.
__device__ __forceinline__ float ssim_out(const RowSums& r0, const RowSums& r1, const RowSums& r2) {
  float v = 0.0f;
  const float r9 = 1.0f / 9.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mx = ((r0.v[c * 5] + r1.v[c * 5]) + r2.v[c * 5]) * r9, my = ((r0.v[c * 5 + 1] + r1.v[c * 5 + 1]) + r2.v[c * 5 + 1]) * r9;
    const float exx = ((r0.v[c * 5 + 2] + r1.v[c * 5 + 2]) + r2.v[c * 5 + 2]) * r9, eyy = ((r0.v[c * 5 + 3] + r1.v[c * 5 + 3]) + r2.v[c * 5 + 3]) * r9;
    const float exy = ((r0.v[c * 5 + 4] + r1.v[c * 5 + 4]) + r2.v[c * 5 + 4]) * r9;
    v += fminf(fmaxf((1.0f - ssim_from_means(mx, my, exx, eyy, exy)) / 2.0f, 0.0f), 1.0f);
  }
  return v;
}

// rigid = 1: the depth-SSIM term (dfe_geom_args.depth_terms): rigid reconstructions, mask = valid & occ & dyna & texture
__global__ void __launch_bounds__(64) k_geom_ssim_fwd_roll(GeomDev D, float* __restrict__ spart, int rigid) {
  __shared__ float red[4];
  const unsigned nunit_total = D.roll_start[D.S];
  const unsigned unit = xcd_swizzle(blockIdx.x, nunit_total);
  const int b = blockIdx.y >> 1, d = blockIdx.y & 1;
  const int s = find_scale(D.roll_start, D.S, unit);
  const int H = D.H[s], W = D.W[s], N = D.N[s];
  const int u = unit - D.roll_start[s];
  const int strip = u % D.roll_strips[s], rb = u / D.roll_strips[s];
  const int x = strip * RS_COLS + static_cast<int>(threadIdx.x) - 1, y0 = rb * RS_ROWS;
  const float* it = D.pyr[1][s] + static_cast<long>(b) * 3 * N;
  const float* yw = (rigid ? D.yr[s] : D.yw[s]) + (static_cast<long>(d) * D.B + b) * 3 * N;
  const unsigned char* mk = D.mode == 2 ? reinterpret_cast<const unsigned char*>(D.wgt[s] + (static_cast<long>(d) * D.B + b) * N)
                                        : D.mask[s] + static_cast<long>(b) * N;
  const unsigned need = D.mode == 2 ? 0u : (rigid ? rigid_ssim_mask(D.mode) : DFE_MASK_VALID_BWD | DFE_MASK_OCC_BWD) << d;
  const bool lane_ok = threadIdx.x >= 1 && threadIdx.x <= RS_COLS && x < W;
  float acc = 0.0f;
  const auto SB = SSIM_SOURCE(it, yw, mk, need, x, H, W, N);
  RowRaw w0 = ssim_load(SB, y0 - 1);
  RowRaw w1 = ssim_load(SB, y0);
  RowRaw w2 = ssim_load(SB, y0 + 1);
  RowSums ra = ssim_hsum(w0), rb0 = ssim_hsum(w1);
  // rows are consumed three at a time so that the rolling window is addressed statically (registers);
  // w2 always holds the raw values of row y+1, and two further rows are loading
  for (int y = y0; y < y0 + RS_ROWS; y += 3) {
    const RowRaw n2 = ssim_load(SB, y + 2);
    const RowSums rc = ssim_hsum(w2);
    if (lane_ok && y < H && y < y0 + RS_ROWS) acc += ssim_out(ra, rb0, rc);
    const RowRaw n3 = ssim_load(SB, y + 3);
    ra = ssim_hsum(n2);
    if (lane_ok && y + 1 < H && y + 1 < y0 + RS_ROWS) acc += ssim_out(rb0, rc, ra);
    w2 = ssim_load(SB, y + 4);
    rb0 = ssim_hsum(n3);
    if (lane_ok && y + 2 < H && y + 2 < y0 + RS_ROWS) acc += ssim_out(rc, ra, rb0);
  }
  float vv[1] = {acc};
  block_sum<1>(vv, red, spart + static_cast<long>(blockIdx.y) * nunit_total + unit);
}

// ---------------------------------------------------------------------- smoothness forward
// Second-order flow smoothness on flow/20 (model_geometry.py:254-279), both directions in one pass.
// Rolling-window wave kernel like the SSIM ones: a wave owns 62 columns (lanes 0..61 produce output, the
// x+1 / x+2 neighbours come from DPP wave shifts) and marches down RS_ROWS rows with a 3-row register window
// for the vertical stencil; the target image rows are loaded once for both flow directions and the next
// row is prefetched.  grid: x = units (strip x row block) over all scales, y = b; block = one wave.
// x-term of row r at this lane (needs lanes l+1, l+2) and y-term of rows (r0, r1, r2); acc = {bwd x, bwd y, fwd x, fwd y}
__device__ __forceinline__ void fs_terms(const FRow& r0, const FRow& r1, const FRow& r2, bool x_ok, bool y_ok, float (&acc)[4]) {
  float i1[3], i2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { i1[c] = wave_shl1(r0.i[c]); i2[c] = wave_shl1(i1[c]); }
  const float wx = expf(-10.0f * mean3_abs_diff(i2[0], i2[1], i2[2], i1[0], i1[1], i1[2]));
  const float wy = expf(-10.0f * mean3_abs_diff(r2.i[0], r2.i[1], r2.i[2], r1.i[0], r1.i[1], r1.i[2]));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a1 = wave_shl1(r0.f[k]), a2 = wave_shl1(a1);
    const float tx = wx * fabsf((a2 - a1) - (a1 - r0.f[k]));
    const float ty = wy * fabsf((r2.f[k] - r1.f[k]) - (r1.f[k] - r0.f[k]));
    if (x_ok) acc[(k >> 1) * 2] += tx;
    if (y_ok) acc[(k >> 1) * 2 + 1] += ty;
  }
}

__global__ void __launch_bounds__(64) k_geom_flow_smooth_fwd(GeomDev D, float* __restrict__ fpart) {
  __shared__ float red[4 * 4];
  const unsigned nunit_total = D.fs_start[D.S];
  // XCD-aware unit order: neighbouring strips / row blocks re-read each other's halo rows and share 128-byte lines
  // (62-column strips are not line-aligned); dealt round-robin over the 8 XCDs they missed in 8 different L2s and the
  // fabric saw 2.3x the algorithmic bytes (profiles/r03_fetch_calib.md: the same shape re-fetches 1.9x)
  const unsigned unit = xcd_swizzle(blockIdx.x, nunit_total);
  const int b = blockIdx.y;
  const int s = find_scale(D.fs_start, D.S, unit);
  const int H = D.H[s], W = D.W[s], N = D.N[s];
  const int u = unit - D.fs_start[s];
  const int strip = u % D.roll_strips[s], rb = u / D.roll_strips[s];
  const int x = strip * RS_COLS + static_cast<int>(threadIdx.x), y0 = rb * FS_ROWS, yend = min(y0 + FS_ROWS, H);
  const float* it = D.pyr[1][s] + static_cast<long>(b) * 3 * N;
  const float* fb = D.flow[0][s] + static_cast<long>(b) * 2 * N;
  const float* ff = D.flow[1][s] + static_cast<long>(b) * 2 * N;
  const bool lane_ok = threadIdx.x < RS_COLS && x < W;
  const bool x_ok = lane_ok && x + 2 < W;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  FRow ra = fs_load(it, fb, ff, y0, x, H, W, N), rb1 = fs_load(it, fb, ff, y0 + 1, x, H, W, N);
  FRow rc = fs_load(it, fb, ff, y0 + 2, x, H, W, N), nx = fs_load(it, fb, ff, y0 + 3, x, H, W, N);
  for (int y = y0; y < yend; y += 3) {
    fs_terms(ra, rb1, rc, x_ok && y < yend, lane_ok && y < yend && y + 2 < H, acc);
    ra = nx; nx = fs_load(it, fb, ff, y + 4, x, H, W, N);
    fs_terms(rb1, rc, ra, x_ok && y + 1 < yend, lane_ok && y + 1 < yend && y + 3 < H, acc);
    rb1 = nx; nx = fs_load(it, fb, ff, y + 5, x, H, W, N);
    fs_terms(rc, ra, rb1, x_ok && y + 2 < yend, lane_ok && y + 2 < yend && y + 4 < H, acc);
    rc = nx; nx = fs_load(it, fb, ff, y + 6, x, H, W, N);
  }
  // partial layout: [(d*B + b)][unit][2]
  float* o0 = fpart + (static_cast<long>(0 * D.B + b) * nunit_total + unit) * 2;
  float* o1 = fpart + (static_cast<long>(1 * D.B + b) * nunit_total + unit) * 2;
  __shared__ float outv[4];
  block_sum<4>(acc, red, outv);
  if (threadIdx.x == 0) { o0[0] = outv[0]; o0[1] = outv[1]; o1[0] = outv[2]; o1[1] = outv[3]; }
}

// First-order edge-aware disparity smoothness at full resolution, all scales fused
// (model_geometry.py:225-252).  Rolling wave kernel: a wave owns 62 full-res columns (lanes 0..61 produce
// output; x+1 comes from a DPP wave shift) and marches down DSM_ROWS rows; the up-sampled disparity of every
// coarser scale is evaluated row by row from a cache of horizontally interpolated low-res rows (loss_stack_exact.h),
// so a row costs 4 streamed loads plus ~1 pair of low-res loads per scale instead of ~36 gathers.
// grid: x = units (strip x row block), y = f*B + b over the 3 frames; block = one wave.  NS = number of scales.
template <int NS>
__global__ void __launch_bounds__(64) k_geom_disp_smooth_fwd(GeomDev D, float* __restrict__ dpart, int strips) {
  __shared__ float red[2 * 4];
  const int f = blockIdx.y / D.B, b = blockIdx.y - f * D.B;
  const int H = D.H[0], W = D.W[0], N = D.N[0];
  const unsigned unit = xcd_swizzle(blockIdx.x, gridDim.x);      // neighbouring strips / row blocks on one XCD (see k_geom_flow_smooth_fwd)
  const int strip = unit % strips, rb = unit / strips;
  const int x = strip * RS_COLS + static_cast<int>(threadIdx.x), xc = min(x, W - 1);
  const int y0 = rb * DSM_ROWS, yend = min(y0 + DSM_ROWS, H);
  const float* im = D.pyr[f][0] + static_cast<long>(b) * 3 * N;
  const float* d0 = D.disp[f][0] + static_cast<long>(b) * N;
  const bool lane_ok = threadIdx.x < RS_COLS && x < W, hx = lane_ok && x + 1 < W;
  UpMap mp[NS > 1 ? NS - 1 : 1];
  UpCache ch[NS > 1 ? NS - 1 : 1];
  const bool small_out = aten_small_resize(H, W);   // tiny full-resolution images only (see up_row)
  const float* dps[NS > 1 ? NS - 1 : 1];
  float rhs[NS > 1 ? NS - 1 : 1];
#pragma unroll
  for (int s = 1; s < NS; ++s) {
    bilinear_src(xc, static_cast<float>(D.W[s]) / W, D.W[s], mp[s - 1].x0, mp[s - 1].x1, mp[s - 1].l0, mp[s - 1].l1);
    ch[s - 1].r0 = -1; ch[s - 1].r1 = -1; ch[s - 1].h0 = 0.0f; ch[s - 1].h1 = 0.0f;
    dps[s - 1] = D.disp[f][s] + static_cast<long>(b) * D.N[s];
    rhs[s - 1] = static_cast<float>(D.H[s]) / H;
  }
  float acc[2] = {0.0f, 0.0f};
  // current row
  int q = y0 * W + xc;
  float c0 = im[q], c1 = im[q + N], c2 = im[q + 2 * N];
  float u[NS];
  u[0] = d0[q];
#pragma unroll
  for (int s = 1; s < NS; ++s) u[s] = up_row(dps[s - 1], D.H[s], D.W[s], rhs[s - 1], y0, mp[s - 1], ch[s - 1], small_out);
  // prefetch of the next row's streamed values
  int qn = min(y0 + 1, H - 1) * W + xc;
  float n0 = im[qn], n1 = im[qn + N], n2 = im[qn + 2 * N], nd = d0[qn];
  for (int y = y0; y < yend; ++y) {
    const float e0 = n0, e1 = n1, e2 = n2, ed = nd;          // row y+1
    const int qf = min(y + 2, H - 1) * W + xc;                // issue the loads of row y+2
    n0 = im[qf]; n1 = im[qf + N]; n2 = im[qf + 2 * N]; nd = d0[qf];
    const bool hy = lane_ok && y + 1 < H;
    float wx = expf(-mean3_abs_diff(c0, c1, c2, wave_shl1(c0), wave_shl1(c1), wave_shl1(c2)));
    float wy = expf(-mean3_abs_diff(c0, c1, c2, e0, e1, e2));
    float un[NS];
    un[0] = ed;
#pragma unroll
    for (int s = 1; s < NS; ++s) un[s] = up_row(dps[s - 1], D.H[s], D.W[s], rhs[s - 1], min(y + 1, H - 1), mp[s - 1], ch[s - 1], small_out);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float ux = wave_shl1(u[s]);
      if (hx) acc[0] += fabsf(u[s] - ux) * wx;
      if (hy) acc[1] += fabsf(u[s] - un[s]) * wy;
      u[s] = un[s];
    }
    c0 = e0; c1 = e1; c2 = e2;
  }
  block_sum<2>(acc, red, dpart + (static_cast<long>(blockIdx.y) * gridDim.x + unit) * 2);
}

// ---------------------------------------------------------------------- finalize
// (Measured and rejected in round 3: ONE block per sample that walks the scales and assembles in place -- 23 us against
// 7 + 9 us for the two launches below: the per-scale passes are latency chains that the (scale, sample) grid runs side by side.)
// k_geom_reduce_fwd: one 256-thread block per (scale, sample) plus one per sample for the disparity-smoothness
// sums.  Phase 1: thread t accumulates the partial rows k = t (mod 256) of every column in double; phase 2: one
// thread per column adds the 256 per-thread sums in thread order.  Both orders are fixed -> bitwise
// reproducible.  k_geom_assemble_fwd (one thread per sample) then forms the eight loss values and the
// normalisers the backward needs.
__global__ void __launch_bounds__(256) k_geom_reduce_fwd(GeomDev D, const float* __restrict__ part,
                                    const float* __restrict__ spart, const float* __restrict__ fpart,
                                    const float* __restrict__ dpart, int ndunit, float* __restrict__ sums,
                                    float* __restrict__ dsum) {
  __shared__ double lds[256][SUM_COUNT + 1];
  __shared__ double parts[8][SUM_COUNT + 1];
  const int s = blockIdx.x, b = blockIdx.y, S = D.S, B = D.B, t = threadIdx.x;
  if (s < S) {
    double a[SUM_COUNT];
#pragma unroll
    for (int i = 0; i < SUM_COUNT; ++i) a[i] = 0.0;
    for (int k = D.blk_start[s] + t; k < D.blk_start[s + 1]; k += 256) {
      const float* r = part + (static_cast<long>(b) * D.blk_start[S] + k) * PT_COUNT;
#pragma unroll
      for (int i = 0; i < PT_COUNT; ++i) a[i] += r[i];
    }
    if (D.mode != 1) {
      for (int k = D.fs_start[s] + t; k < D.fs_start[s + 1]; k += 256) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const float* q = fpart + (static_cast<long>(d * B + b) * D.fs_start[S] + k) * 2;
          a[SUM_FS + 2 * d] += q[0]; a[SUM_FS + 2 * d + 1] += q[1];
        }
      }
      for (int k = D.roll_start[s] + t; k < D.roll_start[s + 1]; k += 256) {
        a[SUM_SSIM] += spart[static_cast<long>(b * 2) * D.roll_start[S] + k];
        a[SUM_SSIM + 1] += spart[static_cast<long>(b * 2 + 1) * D.roll_start[S] + k];
      }
    }
#pragma unroll
    for (int i = 0; i < SUM_COUNT; ++i) lds[t][i] = a[i];
    __syncthreads();
    // column sums in two fixed-order stages (as k_geom_pose_finalize): 8 x 26 threads add 32 per-thread sums each, then 26 threads add
    // the 8 parts -- one thread per column walking all 256 was a chain of 256 dependent double adds, ~10 us of a launch nothing overlaps
    if (t < 8 * SUM_COUNT) {
      const int c = t % SUM_COUNT, pt = t / SUM_COUNT;
      double v = 0.0;
      for (int k = pt * 32; k < pt * 32 + 32; ++k) v += lds[k][c];
      parts[pt][c] = v;
    }
    __syncthreads();
    if (t < SUM_COUNT) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += parts[k][t];
      sums[(static_cast<long>(b) * S + s) * SUM_COUNT + t] = static_cast<float>(v);
    }
  } else {
    double a[6] = {0, 0, 0, 0, 0, 0};
    if (D.mode != 2) for (int k = t; k < ndunit; k += 256)
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const float* q = dpart + (static_cast<long>(f * B + b) * ndunit + k) * 2;
        a[f * 2] += q[0]; a[f * 2 + 1] += q[1];
      }
#pragma unroll
    for (int i = 0; i < 6; ++i) lds[t][i] = a[i];
    __syncthreads();
    if (t < 8 * 6) {
      const int c = t % 6, pt = t / 6;
      double v = 0.0;
      for (int k = pt * 32; k < pt * 32 + 32; ++k) v += lds[k][c];
      parts[pt][c] = v;
    }
    __syncthreads();
    if (t < 6) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += parts[k][t];
      dsum[((t >> 1) * B + b) * 2 + (t & 1)] = static_cast<float>(v);
    }
  }
}

// depth terms: fixed-order sums of the consistency block partials and the rigid-SSIM strip partials of one
// (sample, scale): sums2[(b*S+s)*4 + {0,1}] = SSIM bwd / fwd, + {2,3} = consistency bwd / fwd
__global__ void __launch_bounds__(256) k_geom_reduce_dt(GeomDev D, const float* __restrict__ part2,
                                                        const float* __restrict__ spart2, float* __restrict__ sums2) {
  __shared__ double lds[256][5];
  const int s = blockIdx.x, b = blockIdx.y, S = D.S, t = threadIdx.x;
  double a[4] = {0, 0, 0, 0};
  if (D.dt & DFE_DEPTH_TERM_SSIM)
    for (int k = D.roll_start[s] + t; k < D.roll_start[s + 1]; k += 256) {
      a[0] += spart2[static_cast<long>(b * 2) * D.roll_start[S] + k];
      a[1] += spart2[static_cast<long>(b * 2 + 1) * D.roll_start[S] + k];
    }
  if (D.dt & DFE_DEPTH_TERM_CONSIS)
    for (int k = D.blk_start[s] + t; k < D.blk_start[s + 1]; k += 256) {
      const float* r = part2 + (static_cast<long>(b) * D.blk_start[S] + k) * 2;
      a[2] += r[0]; a[3] += r[1];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) lds[t][i] = a[i];
  __syncthreads();
  __shared__ double parts[8][4];
  if (t < 8 * 4) {
    const int c = t % 4, pt = t / 4;
    double v = 0.0;
    for (int k = pt * 32; k < pt * 32 + 32; ++k) v += lds[k][c];
    parts[pt][c] = v;
  }
  __syncthreads();
  if (t < 4) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += parts[k][t];
    sums2[(static_cast<long>(b) * S + s) * 4 + t] = static_cast<float>(v);
  }
}

__global__ void k_geom_assemble_fwd(GeomDev D, const float* __restrict__ sums, const float* __restrict__ dsum,
                                    const float* __restrict__ sums2, float* __restrict__ coef, float* __restrict__ losses) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, S = D.S, B = D.B;
  if (b >= B) return;
  const double eps = 1e-12;
  double l_dp = 0, l_fp = 0, l_fs = 0, l_sm = 0, l_fc = 0, l_dfc = 0, l_epi = 0, l_dss = 0, l_dcs = 0;
  for (int s = 0; s < S; ++s) {
    const double N = D.N[s], H = D.H[s], W = D.W[s];
    const float* sm = sums + (static_cast<long>(b) * S + s) * SUM_COUNT;
    float* cf = coef + (static_cast<long>(b) * S + s) * CF_COUNT;
    for (int d = 0; d < 2; ++d) {
      const float* a = sm + d * PT_PER_DIR;
      const double n_tex = a[PT_M_TEX] / N + eps, n_rig = a[PT_M_RIG] / N + eps, n_dyn = a[PT_M_DYN] / N + eps,
                   n_vo = a[PT_M_VO] / N + eps;
      l_dp += (a[PT_L1_DEPTH] / (3.0 * N)) / n_tex;
      if (sums2) {   // same texture-gated mask and normaliser as the depth pixel term
        const float* q = sums2 + (static_cast<long>(b) * S + s) * 4;
        l_dss += (q[d] / (3.0 * N)) / n_tex;
        l_dcs += (D.mode == 1) ? q[2 + d] / N : (q[2 + d] / N) / n_tex;   // Model_depth's consistency term carries no mask
      }
      if (D.mode == 2) l_fp += (a[PT_L1_RIG] / N) / n_vo;   // 1-channel diff broadcast over 3 channels (model_flow.py:94-103)
      else l_fp += (a[PT_L1_RIG] / (3.0 * N)) / n_rig + 2.0 * (a[PT_L1_DYN] / (3.0 * N)) / n_dyn;
      l_fs += (sm[SUM_SSIM + d] / (3.0 * N)) / n_vo;
      l_sm += (sm[SUM_FS + 2 * d] / (2.0 * H * (W - 2.0)) + sm[SUM_FS + 2 * d + 1] / (2.0 * (H - 2.0) * W)) / 2.0;
      cf[d * CF_PER_DIR + CF_DEPTH] = static_cast<float>(1.0 / (3.0 * N * n_tex));
      cf[d * CF_PER_DIR + CF_RIG] = static_cast<float>(1.0 / (3.0 * N * (D.mode == 2 ? n_vo : n_rig)));
      cf[d * CF_PER_DIR + CF_DYN] = static_cast<float>(2.0 / (3.0 * N * n_dyn));
      cf[d * CF_PER_DIR + CF_VO] = static_cast<float>(1.0 / (3.0 * N * n_vo));
      cf[d * CF_PER_DIR + CF_FD] = static_cast<float>(1.0 / (2.0 * N * n_rig));
      if (s == 0) {
        l_dfc += (a[PT_FDIFF] / (2.0 * N)) / n_rig;
        l_epi += a[PT_EPI] / N;
      }
    }
    const double n_inv = sm[PT_INV] / N + eps;
    l_fc += (sm[PT_CONSIS] / (2.0 * N)) / n_inv;
    cf[CF_CONSIS] = static_cast<float>(1.0 / (2.0 * N * n_inv));
  }
  double l_ds = 0;
  {
    const double H = D.H[0], W = D.W[0];
    for (int f = 0; f < 3; ++f) l_ds += dsum[(f * B + b) * 2] / (H * (W - 1.0)) + dsum[(f * B + b) * 2 + 1] / ((H - 1.0) * W);
  }
  losses[DFE_LOSS_DEPTH_PIXEL * B + b] = static_cast<float>(l_dp);
  losses[DFE_LOSS_DEPTH_SMOOTH * B + b] = static_cast<float>(l_ds);
  losses[DFE_LOSS_FLOW_PIXEL * B + b] = static_cast<float>(l_fp);
  losses[DFE_LOSS_FLOW_SSIM * B + b] = static_cast<float>(l_fs);
  losses[DFE_LOSS_FLOW_SMOOTH * B + b] = static_cast<float>(l_sm);
  losses[DFE_LOSS_FLOW_CONSIS * B + b] = static_cast<float>(l_fc);
  losses[DFE_LOSS_DEPTH_FLOW_CONSIS * B + b] = static_cast<float>(l_dfc);
  losses[DFE_LOSS_EPIPOLAR * B + b] = static_cast<float>(l_epi);
  losses[DFE_LOSS_DEPTH_SSIM * B + b] = static_cast<float>(l_dss);
  losses[DFE_LOSS_DEPTH_CONSIS * B + b] = static_cast<float>(l_dcs);
}

}  // namespace dfe

using namespace dfe;

static int check_args(const dfe_geom_args* a, const GeomLayout& L, bool bwd) {
  if (!a->workspace) return DFE_ERR_NULL;
  if (a->mode != 2 && (!a->pose || !a->K)) return DFE_ERR_NULL;
  if (a->mode == 0 && !a->K_inv) return DFE_ERR_NULL;
  if (a->workspace_floats < L.total) return DFE_ERR_WORKSPACE;
  for (int f = 0; f < 3; ++f) {
    if (!a->img[f]) return DFE_ERR_NULL;
    if (a->mode != 2) for (int s = 0; s < L.S; ++s) if (!a->disp[f][s]) return DFE_ERR_NULL;
  }
  if (a->mode != 1) for (int d = 0; d < 2; ++d) for (int s = 0; s < L.S; ++s) if (!a->flow[d][s]) return DFE_ERR_NULL;
  if (!bwd && !a->losses) return DFE_ERR_NULL;
  if (bwd && !a->grad_losses) return DFE_ERR_NULL;
  if (a->B > 21845) return DFE_ERR_DIMS;   // grid.y carries up to 3*B (frames x samples) <= 65535
  return DFE_OK;
}

#define DFE_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return DFE_ERR_LAUNCH; } while (0)

extern "C" {

long dfe_geom_workspace_floats(const dfe_geom_args* args) {
  GeomLayout L;
  int rc = geom_layout(args, &L);
  return rc == DFE_OK ? L.total : static_cast<long>(rc);
}

long dfe_geom_maskpack_offset_bytes(const dfe_geom_args* args, int scale) {
  GeomLayout L;
  int rc = geom_layout(args, &L);
  if (rc != DFE_OK) return rc;
  if (scale < 0 || scale >= L.S) return DFE_ERR_DIMS;
  return L.o_mask * 4 + static_cast<long>(L.B) * L.off_px[scale];
}

static int geom_fwd_impl(const dfe_geom_args* a, void* stream, hipEvent_t* ev) {
  GeomLayout L;
  int rc = geom_layout(a, &L);
  if (rc != DFE_OK) return rc;
  rc = check_args(a, L, false);
  if (rc != DFE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* ws = a->workspace;
  GeomDev D;
  geom_dev(a, L, &D);
  GeomT T{};
  tile_dev(L, &T);
  int seg = 0;
#define DFE_MARK() do { if (ev) (void)hipEventRecord(ev[++seg], st); } while (0)
  if (ev) (void)hipEventRecord(ev[0], st);
  // cameras: downscale = H / H_s as the reference computes it (float division of ints)
  float downs[DFE_MAX_SCALES];
  for (int s = 0; s < L.S; ++s) downs[s] = static_cast<float>(static_cast<double>(a->H) / static_cast<double>(L.H[s]));
  if (a->mode != 2) {
    PrepJob pj{a->pose, a->K, a->K_inv, reinterpret_cast<Camera*>(ws + L.o_cams), reinterpret_cast<Epi*>(ws + L.o_epi), L.B, L.S, a->mode, {}};
    for (int s = 0; s < L.S; ++s) pj.downs.v[s] = downs[s];
    k_geom_prepare<<<(L.B * 2 * L.S + 63) / 64, 64, 0, st>>>(pj);
    DFE_LAUNCH_CHECK();
  }
  DFE_MARK();
  // levels 1 and 2 in one pass over the frames when the pyramid is an exact power-of-two one (geom / depth models)
  int first_generic = 1;
  if (L.S >= 3 && a->mode != 2 && a->H % 4 == 0 && a->W % 4 == 0 && getenv("DFE_PYRAMIDS_GENERIC") == nullptr &&
      (reinterpret_cast<uintptr_t>(a->img[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->img[1]) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(a->img[2]) & 15) == 0) {
    Pyr12Job jb;
    jb.planes = L.B * 3; jb.H = a->H; jb.W = a->W;
    for (int f = 0; f < 3; ++f) {
      jb.in[f] = a->img[f];
      jb.b1[f] = const_cast<float*>(D.pyr[f][1]); jb.b2[f] = const_cast<float*>(D.pyr[f][2]);
      jb.a1[f] = f == 1 ? nullptr : const_cast<float*>(D.area[f == 0 ? 0 : 1][1]);
      jb.a2[f] = f == 1 ? nullptr : const_cast<float*>(D.area[f == 0 ? 0 : 1][2]);
    }
    const long n = static_cast<long>(jb.planes) * (a->H / 4) * (a->W / 4);
    k_geom_pyramids12<<<dim3(static_cast<unsigned>((n + 255) / 256), 3), 256, 0, st>>>(jb);
    DFE_LAUNCH_CHECK();
    first_generic = 3;
  }
  if (L.S > first_generic) {
    PyrJobs jobs;
    jobs.n = 0; jobs.planes = L.B * 3; jobs.inH = a->H; jobs.inW = a->W;
    int max_out = 0;
    for (int s = first_generic; s < L.S; ++s) {
      for (int f = 0; f < 3; ++f) {
        if (a->mode == 2)   // Model_flow: box-mean (adaptive_avg_pool2d) pyramid of all three frames, model_flow.py:58-64
          jobs.j[jobs.n++] = PyrJob{a->img[f], nullptr, const_cast<float*>(D.pyr[f][s]), L.H[s], L.W[s]};
        else
          jobs.j[jobs.n++] = PyrJob{a->img[f], const_cast<float*>(D.pyr[f][s]),
                                    f == 1 ? nullptr : const_cast<float*>(D.area[f == 0 ? 0 : 1][s]), L.H[s], L.W[s]};
      }
      if (L.N[s] > max_out) max_out = L.N[s];
    }
    // box means whose window exceeds AREA_COARSE_MIN inputs move to the wave-per-output kernel
    PyrJobs coarse;
    coarse.n = 0; coarse.planes = jobs.planes; coarse.inH = jobs.inH; coarse.inW = jobs.inW;
    int max_coarse = 0, kept = 0;
    for (int k = 0; k < jobs.n; ++k) {
      PyrJob& jb = jobs.j[k];
      const long kh = (a->H + jb.outH - 1) / jb.outH + 1, kw = (a->W + jb.outW - 1) / jb.outW + 1;
      if (jb.out_area && kh * kw > AREA_COARSE_MIN) {
        coarse.j[coarse.n++] = PyrJob{jb.in, nullptr, jb.out_area, jb.outH, jb.outW};
        if (jb.outH * jb.outW > max_coarse) max_coarse = jb.outH * jb.outW;
        jb.out_area = nullptr;
      }
      if (jb.out_bilinear || jb.out_area) jobs.j[kept++] = jb;
    }
    jobs.n = kept;
    if (jobs.n > 0) {
      dim3 g(static_cast<unsigned>((static_cast<long>(jobs.planes) * max_out + 255) / 256), jobs.n);
      k_geom_pyramids<<<g, 256, 0, st>>>(jobs);
      DFE_LAUNCH_CHECK();
    }
    if (coarse.n > 0) {
      dim3 g(static_cast<unsigned>(static_cast<long>(coarse.planes) * max_coarse), coarse.n);
      k_geom_area_coarse<<<g, 64, 0, st>>>(coarse);
      DFE_LAUNCH_CHECK();
    }
  }
  DFE_MARK();
  if (a->mode == 2) {
    // Model_flow: flow warps + soft weights + weighted L1 / SSIM / smoothness / consistency (no depth, no pose)
    k_flow_point_fwd<<<dim3(L.blk_start[L.S], L.B), GS_BLOCK, 0, st>>>(D, ws + L.o_part);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
    k_geom_ssim_fwd_roll<<<dim3(L.roll_start[L.S], L.B * 2), 64, 0, st>>>(D, ws + L.o_spart, 0);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
    k_geom_flow_smooth_fwd<<<dim3(L.fs_start[L.S], L.B), 64, 0, st>>>(D, ws + L.o_fpart);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
  } else if (a->mode == 1) {
    // Model_depth: rigid recon + validity*texture mask + masked L1 only (no flows, no SSIM, no flow terms)
    if (L.dt) k_depth_point_fwd<true><<<dim3(L.blk_start[L.S], L.B), GS_BLOCK, 0, st>>>(D, ws + L.o_part, ws + L.o_part2);
    else k_depth_point_fwd<false><<<dim3(L.blk_start[L.S], L.B), GS_BLOCK, 0, st>>>(D, ws + L.o_part, nullptr);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
    if (L.dt & DFE_DEPTH_TERM_SSIM) {
      k_geom_ssim_fwd_roll<<<dim3(L.roll_start[L.S], L.B * 2), 64, 0, st>>>(D, ws + L.o_spart2, 1);
      DFE_LAUNCH_CHECK();
    }
    DFE_MARK(); DFE_MARK();
  } else {
    if (L.dt) k_geom_point_fwd<true><<<dim3(L.blk_start[L.S], L.B), GS_BLOCK, 0, st>>>(D, T, ws + L.o_part, ws + L.o_part2);
    else k_geom_point_fwd<false><<<dim3(L.blk_start[L.S], L.B), GS_BLOCK, 0, st>>>(D, T, ws + L.o_part, nullptr);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
    k_geom_ssim_fwd_roll<<<dim3(L.roll_start[L.S], L.B * 2), 64, 0, st>>>(D, ws + L.o_spart, 0);
    DFE_LAUNCH_CHECK();
    if (L.dt & DFE_DEPTH_TERM_SSIM) {
      k_geom_ssim_fwd_roll<<<dim3(L.roll_start[L.S], L.B * 2), 64, 0, st>>>(D, ws + L.o_spart2, 1);
      DFE_LAUNCH_CHECK();
    }
    DFE_MARK();
    k_geom_flow_smooth_fwd<<<dim3(L.fs_start[L.S], L.B), 64, 0, st>>>(D, ws + L.o_fpart);
    DFE_LAUNCH_CHECK();
    DFE_MARK();
  }
  if (a->mode != 2) {
    const dim3 g(L.dsm_units, 3 * L.B);
    float* dp = ws + L.o_dpart;
    switch (L.S) {
      case 1: k_geom_disp_smooth_fwd<1><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 2: k_geom_disp_smooth_fwd<2><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 3: k_geom_disp_smooth_fwd<3><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 4: k_geom_disp_smooth_fwd<4><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 5: k_geom_disp_smooth_fwd<5><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 6: k_geom_disp_smooth_fwd<6><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      case 7: k_geom_disp_smooth_fwd<7><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
      default: k_geom_disp_smooth_fwd<8><<<g, 64, 0, st>>>(D, dp, L.dsm_strips); break;
    }
  }
  DFE_LAUNCH_CHECK();
  DFE_MARK();
  k_geom_reduce_fwd<<<dim3(L.S + 1, L.B), 256, 0, st>>>(D, ws + L.o_part, ws + L.o_spart, ws + L.o_fpart, ws + L.o_dpart,
                                                     L.dsm_units, ws + L.o_sums, ws + L.o_dsum);
  DFE_LAUNCH_CHECK();
  if (L.dt) {
    k_geom_reduce_dt<<<dim3(L.S, L.B), 256, 0, st>>>(D, ws + L.o_part2, ws + L.o_spart2, ws + L.o_sums2);
    DFE_LAUNCH_CHECK();
  }
  k_geom_assemble_fwd<<<(L.B + 63) / 64, 64, 0, st>>>(D, ws + L.o_sums, ws + L.o_dsum, L.dt ? ws + L.o_sums2 : nullptr,
                                                      ws + L.o_coef, a->losses);
  DFE_LAUNCH_CHECK();
  DFE_MARK();
#undef DFE_MARK
  return DFE_OK;
}

int dfe_geom_loss_fwd(const dfe_geom_args* a, void* stream) { return geom_fwd_impl(a, stream, nullptr); }

int dfe_geom_loss_fwd_profiled(const dfe_geom_args* a, void* stream, float* ms_host) {
  if (!ms_host) return DFE_ERR_NULL;
  hipEvent_t ev[DFE_GEOM_FWD_SEGMENTS + 1];
  for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return DFE_ERR_LAUNCH;
  int rc = geom_fwd_impl(a, stream, ev);
  if (rc == DFE_OK) {
    (void)hipEventSynchronize(ev[DFE_GEOM_FWD_SEGMENTS]);
    for (int i = 0; i < DFE_GEOM_FWD_SEGMENTS; ++i) (void)hipEventElapsedTime(&ms_host[i], ev[i], ev[i + 1]);
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

// Deferred read-out: events are recorded in stream order with no host synchronisation, so the launches can be
// timed inside a running training step; dfe_geom_timed_collect waits for the last event of that call only.
struct GeomTimed { int n; hipEvent_t ev[DFE_GEOM_FWD_SEGMENTS + 1]; };

int dfe_geom_loss_fwd_timed(const dfe_geom_args* a, void* stream, void** handle) {
  if (!handle) return DFE_ERR_NULL;
  GeomTimed* t = new GeomTimed;
  t->n = DFE_GEOM_FWD_SEGMENTS;
  for (int i = 0; i <= t->n; ++i) if (hipEventCreate(&t->ev[i]) != hipSuccess) { delete t; return DFE_ERR_LAUNCH; }
  const int rc = geom_fwd_impl(a, stream, t->ev);
  if (rc != DFE_OK) { for (int i = 0; i <= t->n; ++i) (void)hipEventDestroy(t->ev[i]); delete t; return rc; }
  *handle = t;
  return DFE_OK;
}

int dfe_geom_timed_collect(void* handle, float* ms_host) {
  if (!handle || !ms_host) return DFE_ERR_NULL;
  GeomTimed* t = static_cast<GeomTimed*>(handle);
  int rc = DFE_OK;
  if (hipEventSynchronize(t->ev[t->n]) != hipSuccess) rc = DFE_ERR_LAUNCH;
  for (int i = 0; i < t->n && rc == DFE_OK; ++i)
    if (hipEventElapsedTime(&ms_host[i], t->ev[i], t->ev[i + 1]) != hipSuccess) rc = DFE_ERR_LAUNCH;
  for (int i = 0; i <= t->n; ++i) (void)hipEventDestroy(t->ev[i]);
  delete t;
  return rc;
}

}  // extern "C"
