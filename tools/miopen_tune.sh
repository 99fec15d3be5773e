#!/bin/bash
# Auto-tunes MIOpen for the convolution problems of the training step (and of the depth / flow modes) on the GPU box and
# leaves the user databases under gpurun_out/miopen_db (gpurun merges them back); copy the *.udb.txt / *.ufdb.txt files
# into unsupervised_depth_opticalflow_egomotion_amd/miopen_db/ to ship them (miopen_tuning.py).
# usage (through gpurun): bash tools/miopen_tune.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/miopen_db
cp -n unsupervised_depth_opticalflow_egomotion_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null   # extend what is shipped
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
for mode in geom depth flow; do
  t0=$(date +%s)
  MIOPEN_FIND_ENFORCE=3 timeout 1500 python bench.py --mode $mode --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune_$mode.log 2>&1
  echo "tuning $mode: exit $? after $(( $(date +%s) - t0 )) s"
done
wc -l gpurun_out/miopen_db/*.txt
