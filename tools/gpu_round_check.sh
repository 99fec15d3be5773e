#!/bin/bash
# Full GPU check of one round: the -m gpu suite, smoke(), a train-step kernel trace, the default bench line.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round_check.sh r02
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/${tag}_pytest_gpu.log
tail -5 $out/${tag}_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; tail -2 $out/${tag}_smoke.log
# MIOpen's find phase first, unprofiled, in its own process: under the profiler its timing-based choices are perturbed
# (and cached for later runs on the box), and its search kernels would fill the trace
timeout 600 python bench.py --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/prof_ts
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ts -o ts -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_train_step_prof_bench.log 2>&1
f=$(find /tmp/prof_ts -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/${tag}_train_step_kernel_stats.csv && python tools/stats_md.py $out/${tag}_train_step_kernel_stats.csv 25 > $out/${tag}_train_step_kernel_stats.md
tail -1 $out/${tag}_train_step_prof_bench.log
timeout 900 python bench.py > $out/${tag}_bench_default.log 2>&1; tail -1 $out/${tag}_bench_default.log
