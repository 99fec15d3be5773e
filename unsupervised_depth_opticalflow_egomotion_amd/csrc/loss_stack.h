// Workspace layout and device-side tables of the fused loss stack (dfe_geom_loss_fwd/bwd).
#pragma once
#include "dfe_device.h"
#include "dfe_internal.h"

namespace dfe {

#ifndef DFE_GS_BLOCK
#define DFE_GS_BLOCK 256
#endif
constexpr int GS_BLOCK = DFE_GS_BLOCK;   // threads per block of the pointwise kernels (1 pixel / thread)
#ifndef DFE_RS_ROWS
#define DFE_RS_ROWS 8
#endif
#ifndef DFE_FS_ROWS
#define DFE_FS_ROWS 8
#endif
#ifndef DFE_RSB_ROWS
#define DFE_RSB_ROWS 8
#endif
constexpr int RS_COLS = 62, RS_ROWS = DFE_RS_ROWS;   // rolling SSIM forward: valid columns per wave (64 lanes - 2 halo), rows per wave
                                            // (measured: rows 4/6/8/16/32 -> 24.2/22.8/21.9/24.0/30.3 us; LDS tile kernel 33 us;
                                            // round 6 in the step: 6/8/9/12 rows -> 26.5/27.4/27.9/28.5)
constexpr int DSM_ROWS = 8;                 // rolling disparity-smoothness kernels: full-res rows per wave (62 valid columns)
constexpr int FS_ROWS = DFE_FS_ROWS;                  // rolling flow-smoothness kernels: rows per wave (62 valid columns);
                                            // measured 2/4/8 rows -> 24.9/20.6/16.3 us (per-pixel kernel: 24.3 us)
constexpr int RSB_COLS = 60, RSB_ROWS = DFE_RSB_ROWS;  // rolling SSIM backward: 2-lane halo on each side
                                            // (measured: rows 5/8/16 -> 49.2/44.4/48.0 us; LDS tile kernel 56.9 us; round 6, with the
                                            // 96-register kernel: 9/12/15 rows -> 49.4/44.8/45.6 against 46.6 in the step, and
                                            // k_geom_flow_smooth_bwd, which shares this strip table, 29.1/33.5/38.3 against 28.3)

// ---- per-block partial sums of k_geom_point_fwd (per direction d: index d*PT_PER_DIR + i)
enum { PT_M_TEX = 0, PT_L1_DEPTH, PT_M_RIG, PT_L1_RIG, PT_M_DYN, PT_L1_DYN, PT_M_VO, PT_FDIFF, PT_EPI, PT_PER_DIR };
constexpr int PT_INV = 2 * PT_PER_DIR, PT_CONSIS = PT_INV + 1, PT_COUNT = PT_CONSIS + 1;   // 20

// ---- reduced sums per (b, scale): [0,PT_COUNT) pointwise, then ssim[2], flow-smooth x/y per dir
constexpr int SUM_SSIM = PT_COUNT, SUM_FS = SUM_SSIM + 2, SUM_COUNT = SUM_FS + 4;            // 26

// ---- normalisers saved for the backward per (b, scale): per direction then shared
enum { CF_DEPTH = 0, CF_RIG, CF_DYN, CF_VO, CF_FD, CF_PER_DIR };
constexpr int CF_CONSIS = 2 * CF_PER_DIR, CF_COUNT = CF_CONSIS + 1;                          // 11

// ---- per-block partial sums of k_geom_point_bwd: per direction 12 camera sums + 9 dF sums
constexpr int PB_PER_DIR = 21, PB_COUNT = 2 * PB_PER_DIR;

struct Epi { float F[9]; float Kinv[9]; float S[9]; };   // fundamental matrix and the factors its backward needs

struct GeomLayout {
  int B, S;
  int H[DFE_MAX_SCALES], W[DFE_MAX_SCALES], N[DFE_MAX_SCALES];
  long off_px[DFE_MAX_SCALES + 1];   // prefix of N (per single image plane)
  int nblk[DFE_MAX_SCALES], blk_start[DFE_MAX_SCALES + 1];      // pointwise blocks per image
  int nblk0;                          // full-resolution blocks (disp smoothness)
  int roll_start[DFE_MAX_SCALES + 1], roll_strips[DFE_MAX_SCALES];   // rolling-SSIM units (strip x row block) per scale
  int rollb_start[DFE_MAX_SCALES + 1], rollb_strips[DFE_MAX_SCALES]; // same for the backward kernel
  int fs_start[DFE_MAX_SCALES + 1];                                  // flow-smoothness units (roll_strips x FS_ROWS blocks)
  int dsm_units, dsm_strips;                                         // disparity-smoothness units at full resolution
  // workspace offsets in floats
  long o_wgt;       // mode 2: soft occlusion weights, [scale][2][B][N_s] floats
  long o_cams, o_epi, o_pyr, o_area, o_mask, o_yw, o_part, o_spart, o_fpart, o_dpart, o_sums, o_coef, o_dsum,
      o_gw, o_gup, o_bpart, total;
  long pyr_plane;   // floats of one frame's bilinear pyramid levels >= 1 (B*3*sum_{s>=1} N_s)
  // optional depth terms (dfe_geom_args.depth_terms, mode 0): regions appended after `o_bpart` so that every other
  // offset (the mask pack's in particular) does not depend on the flag
  int dt;
  long o_yr, o_gyr, o_part2, o_spart2, o_sums2;   // masked rigid warps, their SSIM gradient, block / strip partials, sums
  long o_scq;       // depth-consistency term: fixed-point accumulators of the projected-depth scatter (dfe_scatter.h),
                    // 64-byte header + int64 [2 source frames][scale][B][N_s]
  // separable adjoint of the bilinear up-sampling at the coarse levels (ratio below 1/4, s >= adj_s0; none: adj_s0 = S):
  // column sums of the up-sampled gradients per segment of adj_L[s] rows, [3][S - adj_s0][B][adj_nseg][ADJ_SLOTS][W_0]
  int adj_mode, adj_s0, adj_nseg, adj_L[DFE_MAX_SCALES];
  long o_adjp;
};
constexpr int ADJ_SLOTS = 8;      // low-res rows a segment can touch: ceil(L * ratio) + 2 with L * ratio <= 6
constexpr int ADJ_LMAX = 32;      // rows per segment (all of a lane's loads in flight at once)

// Kernel-argument tables (passed by value).
struct GeomDev {
  int B, S, ac, mode;
  float alpha, beta;
  int H[DFE_MAX_SCALES], W[DFE_MAX_SCALES], N[DFE_MAX_SCALES];
  int blk_start[DFE_MAX_SCALES + 1];
  int roll_start[DFE_MAX_SCALES + 1], roll_strips[DFE_MAX_SCALES];   // rolling-SSIM unit table
  int rollb_start[DFE_MAX_SCALES + 1], rollb_strips[DFE_MAX_SCALES];
  int fs_start[DFE_MAX_SCALES + 1];
  const float* pyr[3][DFE_MAX_SCALES];    // bilinear pyramid per frame (level 0 = the frame itself)
  const float* area[2][DFE_MAX_SCALES];   // area pyramid of the left / right frame
  const float* disp[3][DFE_MAX_SCALES];
  const float* flow[2][DFE_MAX_SCALES];
  const Camera* cams;                      // [(b*2+d)*S + s]
  const Epi* epi;                          // [b*2+d]
  unsigned char* mask[DFE_MAX_SCALES];     // [B][N_s]
  float* yw[DFE_MAX_SCALES];               // [2][B][3][N_s] masked warped images
  float* wgt[DFE_MAX_SCALES];              // mode 2: [2][B][N_s] soft occlusion weights
  int dt;                                  // DFE_DEPTH_TERM_* bits (mode 0)
  float* yr[DFE_MAX_SCALES];               // dt & SSIM: [2][B][3][N_s] rigid reconstructions x texture-gated mask
};

int geom_layout(const dfe_geom_args* a, GeomLayout* L);
void geom_dev(const dfe_geom_args* a, const GeomLayout& L, GeomDev* D);

// ---- rolling-window SSIM helpers (forward and backward kernels)
struct RowSums { float v[15]; };   // per channel: sum x, y, xx, yy, xy over the 3 horizontal neighbours
struct RowRaw { float a[3], b[3], vo; };   // masked target / warped values of one row at this lane's column, and the mask weight

// issue the 7 loads of one row (returns zeros outside the image); kept separate from the DPP sums so that
// the loads of rows y+2.. are in flight while row y is reduced (software prefetch: ~2 waves per SIMD only)
// The per-pixel SSIM weight is either a bit test on the mask pack (modes 0/1: valid & occ of this direction) or a
// float soft weight (mode 2, Model_flow): `mk` then points at the float plane and `need` is 0.
// mask bits (direction 0) of the depth-SSIM term: Model_geometry gates it with valid & occ & dyna & texture
// (model_geometry.py:889-891), Model_depth with inverse_warp2-validity & texture (model_depth.py:326-327)
__device__ __forceinline__ unsigned rigid_ssim_mask(int mode) {
  return mode == 1 ? (DFE_MASK_VALID_BWD | DFE_MASK_TEX_BWD) : (DFE_MASK_VALID_BWD | DFE_MASK_OCC_BWD | DFE_MASK_DYNA_BWD | DFE_MASK_TEX_BWD);
}

__device__ __forceinline__ float ssim_weight_at(const unsigned char* __restrict__ mk, unsigned need, int q) {
  if (need == 0u) return reinterpret_cast<const float*>(mk)[q];
  return ((mk[q] & need) == need) ? 1.0f : 0.0f;
}

__device__ __forceinline__ RowRaw ssim_load(const float* __restrict__ it, const float* __restrict__ yw,
                                            const unsigned char* __restrict__ mk, unsigned need, int y, int x,
                                            int H, int W, int N) {
  RowRaw r;
  const bool in = y >= 0 && y < H && x >= 0 && x < W;
  const int q = in ? y * W + x : 0;
  const float wq = ssim_weight_at(mk, need, q);
  float ta[3], tb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { ta[c] = it[q + c * N]; tb[c] = yw[q + c * N]; }
  const float vo = in ? wq : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { r.a[c] = in ? ta[c] * vo : 0.0f; r.b[c] = in ? tb[c] : 0.0f; }
  r.vo = vo;
  return r;
}

// The same row through buffer loads (round 6): the three planes of the target, of the reconstruction and the mask as raw buffer
// resources (wave-uniform bases), the lane's column as a 32-bit byte offset, the row and the channel in the scalar offset.  A lane
// outside the image carries an offset past the end of the buffer and the hardware returns zeros: no 64-bit address arithmetic and
// no selects per value (ssim_load spends ~35 vector instructions per row on them, this form ~8).  Same values, same arithmetic.
constexpr unsigned SSIM_OOB = 0x7ffffff0u;      // + any scalar offset of a real tensor stays below 2^32 and past every buffer's end
struct SsimBuf {
  __amdgpu_buffer_rsrc_t it, yw, mk;
  unsigned need, N4, W4, colf;                  // colf: the lane's column as a byte offset into a float plane, SSIM_OOB outside [0, W)
  int H, W;
};
__device__ __forceinline__ SsimBuf ssim_buf(const float* it, const float* yw, const unsigned char* mk, unsigned need, int x, int H, int W, int N) {
  SsimBuf S;
  S.it = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(it), 0, 12 * N, 0x00020000);
  S.yw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(yw), 0, 12 * N, 0x00020000);
  S.mk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(mk), 0, need == 0u ? 4 * N : N, 0x00020000);
  S.need = need; S.N4 = 4u * N; S.W4 = 4u * W; S.H = H; S.W = W;
  const bool col_in = x >= 0 && x < W;
  S.colf = col_in ? 4u * static_cast<unsigned>(x) : SSIM_OOB;
  return S;
}
__device__ __forceinline__ RowRaw ssim_load(const SsimBuf& S, int y) {      // y is wave-uniform
  RowRaw r;
  const bool row_in = y >= 0 && y < S.H;
  const unsigned vf = row_in ? S.colf : SSIM_OOB, rowf = row_in ? static_cast<unsigned>(y) * S.W4 : 0u;
  float ta[3], tb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    ta[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(S.it, vf, rowf + c * S.N4, 0));
    tb[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(S.yw, vf, rowf + c * S.N4, 0));
  }
  float vo;
  if (S.need == 0u) vo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(S.mk, vf, rowf, 0));
  else {
    const unsigned vb = vf >> 2;                    // the byte plane's offset; SSIM_OOB >> 2 is still past its end
    const unsigned bits = __builtin_amdgcn_raw_buffer_load_b8(S.mk, vb, row_in ? static_cast<unsigned>(y) * static_cast<unsigned>(S.W) : 0u, 0);
    vo = ((bits & S.need) == S.need) ? 1.0f : 0.0f;      // an out-of-range lane reads 0: need != 0, so vo = 0
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { r.a[c] = ta[c] * vo; r.b[c] = tb[c]; }
  r.vo = vo;
  return r;
}

// the pointer form behind the same interface (DFE_SSIM_BUF=0: the A/B build)
struct SsimPtr { const float* it; const float* yw; const unsigned char* mk; unsigned need; int x, H, W, N; };
__device__ __forceinline__ RowRaw ssim_load(const SsimPtr& S, int y) { return ssim_load(S.it, S.yw, S.mk, S.need, y, S.x, S.H, S.W, S.N); }
#ifndef DFE_SSIM_BUF
#define DFE_SSIM_BUF 1
#endif
#if DFE_SSIM_BUF
#define SSIM_SOURCE(it, yw, mk, need, x, H, W, N) ssim_buf(it, yw, mk, need, x, H, W, N)
#else
#define SSIM_SOURCE(it, yw, mk, need, x, H, W, N) SsimPtr{it, yw, mk, need, x, H, W, N}
#endif

__device__ __forceinline__ RowSums ssim_hsum(const RowRaw& w) {
  RowSums r;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    r.v[c * 5 + 0] = wave_nbr_sum(w.a[c]);
    r.v[c * 5 + 1] = wave_nbr_sum(w.b[c]);
    r.v[c * 5 + 2] = wave_nbr_sum(w.a[c] * w.a[c]);
    r.v[c * 5 + 3] = wave_nbr_sum(w.b[c] * w.b[c]);
    r.v[c * 5 + 4] = wave_nbr_sum(w.a[c] * w.b[c]);
  }
  return r;
}

// ---- rolling disparity-smoothness helpers: bilinear up-sampling (ATen, align_corners=False) of a low-res
// plane evaluated row by row.  UpMap is this lane's horizontal tap (fixed for the whole march); UpCache holds the
// horizontally interpolated values of the two low-res rows the current full-res row needs, and is advanced
// (uniformly across the wave) as the wave marches down -- so a full-res row costs ~1 pair of loads per scale.
struct UpMap { int x0, x1; float l0, l1; };
struct UpCache { int r0, r1; float h0, h1; };

__device__ __forceinline__ float up_hrow(const float* __restrict__ dp, int Ws, int r, const UpMap& m) {
  const float* row = dp + r * Ws;
  return lerp_aten(row[m.x0], row[m.x1], m.l0, m.l1);
}

// `small`: the full-resolution output has H + W <= 128 (tiny images only) -> ATen's non-separable association
// (lerp2_aten_small), evaluated directly from the four taps; wave-uniform branch.
__device__ __forceinline__ float up_row(const float* __restrict__ dp, int Hs, int Ws, float rh, int y, const UpMap& m, UpCache& c,
                                        bool small = false) {
  int a0, a1; float ly0, ly1;
  bilinear_src(y, rh, Hs, a0, a1, ly0, ly1);          // wave-uniform
  if (small) {
    const float* r0 = dp + a0 * Ws;
    const float* r1 = dp + a1 * Ws;
    return lerp2_aten_small(r0[m.x0], r0[m.x1], r1[m.x0], r1[m.x1], m.l0, m.l1, ly0, ly1);
  }
  float n0, n1;
  if (a0 == c.r0) n0 = c.h0; else if (a0 == c.r1) n0 = c.h1; else n0 = up_hrow(dp, Ws, a0, m);
  if (a1 == c.r1) n1 = c.h1; else if (a1 == c.r0) n1 = c.h0; else if (a1 == a0) n1 = n0; else n1 = up_hrow(dp, Ws, a1, m);
  c.r0 = a0; c.r1 = a1; c.h0 = n0; c.h1 = n1;
  return lerp_aten(n0, n1, ly0, ly1);
}

// ---- rolling flow-smoothness helpers
struct FRow { float i[3]; float f[4]; };   // image row values and flow/20 (bwd u, v, fwd u, v) at this lane's column

__device__ __forceinline__ FRow fs_load(const float* __restrict__ it, const float* __restrict__ fb, const float* __restrict__ ff,
                                        int y, int x, int H, int W, int N) {
  FRow r;
  const bool in = y >= 0 && y < H && x >= 0 && x < W;
  const int q = in ? y * W + x : 0;
  const Divisor D20{20.0f, 1.0f / 20.0f};
  const float t0 = it[q], t1 = it[q + N], t2 = it[q + 2 * N];
  const float u0 = fb[q], v0 = fb[q + N], u1 = ff[q], v1 = ff[q + N];
  r.i[0] = t0; r.i[1] = t1; r.i[2] = t2;
  r.f[0] = div_exact(u0, D20); r.f[1] = div_exact(v0, D20); r.f[2] = div_exact(u1, D20); r.f[3] = div_exact(v1, D20);
  return r;
}

__device__ __forceinline__ int find_scale(const int* starts, int S, int idx) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < DFE_MAX_SCALES; ++k) s += (k < S && idx >= starts[k]) ? 1 : 0;
  return s;
}

}  // namespace dfe
