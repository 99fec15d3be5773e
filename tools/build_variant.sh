#!/bin/bash
# A/B builds of the fused loss stack: relinks the library with loss_stack_fwd/bwd.hip compiled under extra -D flags; select it with DFE_HIP_LIB.
# usage: tools/build_variant.sh <name> <extra -D flags...>  -> scratch/libs/libdfe_hip_<name>.so (only the loss-stack objects are rebuilt)
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/unsupervised_depth_opticalflow_egomotion_amd/csrc
mkdir -p $ROOT/scratch/libs/$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
for f in loss_stack_fwd loss_stack_bwd; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $ROOT/scratch/libs/$name/$f.o &
done
wait
objs=$(ls *.o | grep -v "^loss_stack_" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $ROOT/scratch/libs/$name/loss_stack_fwd.o $ROOT/scratch/libs/$name/loss_stack_bwd.o -o $ROOT/scratch/libs/libdfe_hip_$name.so
rm -rf $ROOT/scratch/libs/$name
echo built $name
