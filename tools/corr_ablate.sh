#!/bin/bash
# Builds one library per DFE_CORR_ABL value (ops_corr.hip ablation switches: 1 no stores, 2 no FMAs, 4 no staging loads)
# into scratch/abl/ (CPU, hipcc cross-compiles).  usage: bash tools/corr_ablate.sh "0 1 2 4"; on the GPU box:
#   python tools/corr_bench.py --only-corr --lib scratch/abl/libdfe_hip_corr<k>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/unsupervised_depth_opticalflow_egomotion_amd/csrc
mkdir -p $R/scratch/abl
OBJS=$(ls $C/*.o | grep -v ops_corr.o)
for k in ${1:-0 1 2 4}; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -DDFE_CORR_ABL=$k -c $C/ops_corr.hip -o $R/scratch/abl/corr_$k.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $R/scratch/abl/corr_$k.o -o $R/scratch/abl/libdfe_hip_corr$k.so && echo built $k ) &
done
wait
