#!/bin/bash
# Builds one library per DFE_ABL value (loss_stack_fwd.hip ablation switches) into scratch/abl/ (CPU, hipcc cross-compiles).
# usage: bash tools/ablate_point_fwd.sh "0 1 2 4 ..."     then on the GPU box: python tools/ls_segments.py --lib scratch/abl/libdfe_hip_abl<k>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/unsupervised_depth_opticalflow_egomotion_amd/csrc
mkdir -p $R/scratch/abl
OBJS=$(ls $C/*.o | grep -v loss_stack_fwd.o)
for k in ${1:-0 1 2 4 8 16 24 32 64 128 253}; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -DDFE_ABL=$k -c $C/loss_stack_fwd.hip -o $R/scratch/abl/fwd_$k.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $R/scratch/abl/fwd_$k.o -o $R/scratch/abl/libdfe_hip_abl$k.so && echo built $k ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
