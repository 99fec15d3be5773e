// Internal (non-ABI) declarations shared by the translation units of libdfe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/dfe_hip.h"

namespace dfe {
struct ScaleList { float v[DFE_MAX_SCALES]; };
struct IntList { int v[DFE_MAX_SCALES]; };

// ops_corr.hip: the PWC cost volume and its gradients (LDS-staged).  out / gout: 81 planes per sample with batch stride
// obs / gbs (the planes may be a slice of a wider tensor); add1 (batch stride abs1) is added to g1 when given.
// tail / flow (optional): planes 81 ... of the same tensor as `out` (a PWC level's x) receive f1 and the two flow planes
int launch_corr_fwd(const float* f1, const float* f2, float* out, long obs, float* tail, const float* flow, int B, int C, int H, int W,
                    hipStream_t st);
int launch_corr_bwd(const float* f1, const float* f2, const float* gout, long gbs, const float* add1, long abs1, float* g1,
                    float* g2, unsigned* amax2, int B, int C, int H, int W, hipStream_t st);
}  // namespace dfe
