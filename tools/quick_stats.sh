#!/bin/bash
# kernel-trace stats of a short train-step run; prints this build's kernels (usage: bash tools/quick_stats.sh [pattern])
export TMPDIR=/tmp
rm -rf /tmp/qs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/qs.log 2>&1
f=$(find /tmp/qs -name '*kernel_stats.csv' | head -1)
python tools/stats_md.py $f 12 | grep -E "${1:-.}"
tail -1 /tmp/qs.log | cut -c1-160
