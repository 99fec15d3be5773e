#!/bin/bash
# Runs ON THE GPU BOX: measured HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and durations of
# this build's net-glue kernels inside the train step -> gpurun_out/<tag>_glue_traffic.md   (usage: bash tools/glue_traffic.sh r03)
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --net-streams 1 > /dev/null 2>&1     # MIOpen find phase, unprofiled
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/gt_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/gt_$C -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --net-streams 1 > /tmp/gt_$C.log 2>&1
  cp "$(find /tmp/gt_$C -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_glue_$C.csv
done
python3 $R/tools/glue_table.py $TAG | tee $OUT/${TAG}_glue_traffic.md
