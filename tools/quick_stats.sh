#!/bin/bash
# kernel-trace stats of a short train-step run (steady state: MIOpen's find phase runs in a separate warm-up process
# first so that its search kernels stay out of the trace); prints this build's kernels matching [pattern] and the
# per-step breakdown.  usage: bash tools/quick_stats.sh [pattern]
export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/qs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/qs.log 2>&1
f=$(find /tmp/qs -name '*kernel_stats.csv' | head -1)
mkdir -p gpurun_out; cp $f gpurun_out/quick_kernel_stats.csv
python tools/stats_md.py $f 0 | grep -E "${1:-.}"
python tools/step_breakdown.py $f 13
tail -1 /tmp/qs.log | cut -c1-160
