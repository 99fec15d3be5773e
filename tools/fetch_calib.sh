#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of tools/ubench/fetch_calib (known byte counts) -> gpurun_out/<tag>_fetch_calib.md
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/ubench/fetch_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/ubench/fetch_calib.hip -o $R/tools/ubench/fetch_calib
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fc_$C -o t -- $R/tools/ubench/fetch_calib > /tmp/fc_$C.log 2>&1
  cp "$(find /tmp/fc_$C -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_fetch_calib_$C.csv
done
python3 $R/tools/fetch_calib.py $TAG | tee $OUT/${TAG}_fetch_calib.md
