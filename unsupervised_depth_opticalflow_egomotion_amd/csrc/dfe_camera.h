// Camera block arithmetic shared by k_prepare_cameras (ops_basic.hip) and the fused loss stack's prepare job
// (loss_stack_fwd.hip): the reference's own fp32 association orders for R, K_s [R | t] and K_s^-1.
#pragma once
#include "dfe_device.h"
#include "dfe_internal.h"

namespace dfe {

// ====================================================================== cameras
__device__ inline void mat3_mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__device__ inline void euler_mats(double rx, double ry, double rz, double* X, double* Y, double* Z,
                                  double* dX, double* dY, double* dZ) {
  double cx = cos(rx), sx = sin(rx), cy = cos(ry), sy = sin(ry), cz = cos(rz), sz = sin(rz);
  double x[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  double y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  double z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double dx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
  double dy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
  double dz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
  for (int i = 0; i < 9; ++i) { X[i] = x[i]; Y[i] = y[i]; Z[i] = z[i]; dX[i] = dx[i]; dY[i] = dy[i]; dZ[i] = dz[i]; }
}

// ---- the reference's own fp32 arithmetic for the camera block (probed bit for bit against ATen CPU, torch 2.10):
//  * xmat @ ymat @ zmat, intrinsics @ pose_mat and the 3x3 products of the essential / fundamental matrix are
//    "small" bmm calls (contraction * rows * cols < 400): ATen's scalar loop acc = 0; acc += a[k] * b[k][j]
//    in k order, every product and sum rounded to fp32, no FMA (aten/src/ATen/native/LinearAlgebra.cpp
//    baddbmm_cpu_kernel);
//  * intrinsics.inverse() (inverse_warp.py:284,329) = LAPACK getrf + getrs on the transposed storage: partial
//    pivoting, first column scaled by the reciprocal of the pivot, second column divided, FMA Schur updates,
//    reciprocal diagonal in the triangular solve.  Bit-identical for intrinsics that need no row exchange
//    (|fx| >= |cx|, |fy| >= |cy|: every pinhole camera with a field of view below 90 degrees, KITTI: 0.58 W vs
//    0.5 W); within a few ulp otherwise (the exchange path of MKL was not pinned);
//  * cos / sin: ATen uses MKL VML (HA mode, <= 0.6 ulp, proprietary); here the correctly rounded value
//    (double evaluation rounded once).  The two agree whenever the exact value is not within 0.1 ulp of a
//    rounding boundary; the parity tests use poses for which that holds (synthetic.robust_pose).
__device__ inline void mat3_mul_f32(const float* a, const float* b, float* o, int ncol = 3, int ldb = 3) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < ncol; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < 3; ++k) acc = __fadd_rn(acc, __fmul_rn(a[i * 3 + k], b[k * ldb + j]));
      o[i * ncol + j] = acc;
    }
}

__device__ inline void inverse3_lapack_f32(const float* A, float* X) {
  float M[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 3 + j] = A[j * 3 + i];   // M = A^T
  int perm[3] = {0, 1, 2};
  for (int j = 0; j < 2; ++j) {
    int p = j;
    for (int i = j + 1; i < 3; ++i) if (fabsf(M[i * 3 + j]) > fabsf(M[p * 3 + j])) p = i;
    if (p != j) {
      for (int k = 0; k < 3; ++k) { const float t = M[j * 3 + k]; M[j * 3 + k] = M[p * 3 + k]; M[p * 3 + k] = t; }
      const int t = perm[j]; perm[j] = perm[p]; perm[p] = t;
    }
    if (j == 0) {
      const float r = __fdiv_rn(1.0f, M[0]);
      M[3] = __fmul_rn(M[3], r); M[6] = __fmul_rn(M[6], r);
    } else {
      M[7] = __fdiv_rn(M[7], M[4]);
    }
    for (int i = j + 1; i < 3; ++i)
      for (int k = j + 1; k < 3; ++k) M[i * 3 + k] = __fmaf_rn(-M[i * 3 + j], M[j * 3 + k], M[i * 3 + k]);
  }
  // A = U^T L^T P: U^T Y = I (forward), L^T W = Y (backward, unit diagonal), X = P^T W
  float Y[9], Wm[9];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) {
      float s = (i == c) ? 1.0f : 0.0f;
      for (int k = 0; k < i; ++k) s = __fsub_rn(s, __fmul_rn(M[k * 3 + i], Y[k * 3 + c]));
      Y[i * 3 + c] = __fmul_rn(s, __fdiv_rn(1.0f, M[i * 3 + i]));
    }
  for (int c = 0; c < 3; ++c)
    for (int i = 2; i >= 0; --i) {
      float s = Y[i * 3 + c];
      for (int k = i + 1; k < 3; ++k) s = __fsub_rn(s, __fmul_rn(M[k * 3 + i], Wm[k * 3 + c]));
      Wm[i * 3 + c] = s;
    }
  for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) X[perm[i] * 3 + c] = Wm[i * 3 + c];
}

// correctly rounded fp32 cos / sin (double evaluation, one rounding)
__device__ inline float cos_cr(float x) { return static_cast<float>(cos(static_cast<double>(x))); }
__device__ inline float sin_cr(float x) { return static_cast<float>(sin(static_cast<double>(x))); }

// R = Rx Ry Rz exactly as euler2mat evaluates it (inverse_warp.py:110-145)
__device__ inline void euler_rotation_f32(float rx, float ry, float rz, float* R) {
  const float cx = cos_cr(rx), sx = sin_cr(rx), cy = cos_cr(ry), sy = sin_cr(ry), cz = cos_cr(rz), sz = sin_cr(rz);
  const float x[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  const float y[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  const float z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  float xy[9];
  mat3_mul_f32(x, y, xy);
  mat3_mul_f32(xy, z, R);
}

// Camera block of one (sample, direction, scale): pv = the 6-vector (tx,ty,tz,rx,ry,rz), Kb = the sample's K [9],
// down = H / H_s as the reference computes it (model_geometry.py:92-93, inverse_warp.py:284-289).
__device__ inline void make_camera(const float* pv, const float* Kb, float down, Camera& c) {
  // the reference divides in fp32: intrinsics[:,0:2]/downscale
  for (int i = 0; i < 9; ++i) {
    const float k = Kb[i];
    c.K[i] = (i < 6) ? __fdiv_rn(k, down) : k;
  }
  euler_rotation_f32(pv[3], pv[4], pv[5], c.R);
  // proj = K_s @ [R | t]  (inverse_warp.py:289): A = proj[:, :3], b = proj[:, 3]
  float T34[12], P[12];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T34[i * 4 + j] = c.R[i * 3 + j]; T34[i * 4 + 3] = pv[i]; }
  mat3_mul_f32(c.K, T34, P, 4, 4);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c.A[i * 3 + j] = P[i * 4 + j]; c.b[i] = P[i * 4 + 3]; }
  inverse3_lapack_f32(c.K, c.kinv);
  // dR/d(rx) = dX Y Z ; dR/d(ry) = X dY Z ; dR/d(rz) = X Y dZ  (backward only: double, rounded once)
  double X[9], Y[9], Z[9], dX[9], dY[9], dZ[9], XY[9], T1[9], T2[9];
  euler_mats(pv[3], pv[4], pv[5], X, Y, Z, dX, dY, dZ);
  mat3_mul(X, Y, XY);
  mat3_mul(dX, Y, T1); mat3_mul(T1, Z, T2);
  for (int i = 0; i < 9; ++i) c.dR[i] = static_cast<float>(T2[i]);
  mat3_mul(X, dY, T1); mat3_mul(T1, Z, T2);
  for (int i = 0; i < 9; ++i) c.dR[9 + i] = static_cast<float>(T2[i]);
  mat3_mul(XY, dZ, T2);
  for (int i = 0; i < 9; ++i) c.dR[18 + i] = static_cast<float>(T2[i]);
}

}  // namespace dfe
