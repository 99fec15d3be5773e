#!/bin/bash
# Wave-footprint experiment of the pointwise kernels (loss_stack_exact.h tile_pixel): builds libdfe_hip.so variants with
# k_geom_point_fwd (-DDFE_PT_TILE=w) / k_geom_point_bwd (-DDFE_PB_TILE=w) covering w x (64/w) pixel tiles per wave
# (0 = 64 consecutive pixels of a row) into scratch/abl/ (CPU; hipcc cross-compiles).  On the GPU box:
#   python tools/ls_segments.py [--batch 16 --height 375 --width 1242 --scales 6] --lib scratch/abl/libdfe_hip_tile<w>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/unsupervised_depth_opticalflow_egomotion_amd/csrc
mkdir -p $R/scratch/abl
OBJS=$(ls $C/*.o | grep -v "loss_stack_fwd.o\|loss_stack_bwd.o")
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
for k in ${1:-0 8 16 32}; do
  ( /opt/rocm/bin/hipcc $FL -DDFE_PT_TILE=$k -c $C/loss_stack_fwd.hip -o $R/scratch/abl/fwd_tile$k.o && \
    /opt/rocm/bin/hipcc $FL -DDFE_PB_TILE=$k -c $C/loss_stack_bwd.hip -o $R/scratch/abl/bwd_tile$k.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $R/scratch/abl/fwd_tile$k.o $R/scratch/abl/bwd_tile$k.o -o $R/scratch/abl/libdfe_hip_tile$k.so && echo built tile$k ) &
done
wait
