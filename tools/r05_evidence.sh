#!/bin/bash
# Round-5 evidence run ON THE GPU BOX (through gpurun): everything DESIGN.md / profiles/ of the round quote, into gpurun_out/r05_*.
# usage: bash tools/r05_evidence.sh [skip_tests]
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
if [ -z "$1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $out/r05_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/r05_pytest_gpu.log; tail -3 $out/r05_pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r05_smoke.log 2>&1; tail -1 $out/r05_smoke.log
fi
# fused loss stack: per-kernel roofline table at the headline shape (also refreshes profiles/pmc_point_fwd_traffic.json's source hash)
bash tools/pmc_loss_stack.sh r05 > /dev/null 2>&1
cp $out/r05_pmc_point_fwd_traffic.json profiles/pmc_point_fwd_traffic.json 2>/dev/null
# the default bench line (with the CPU baseline), twice more without it, the forced-DDP line
timeout 900 python bench.py > $out/r05_bench_default.log 2>&1; grep '^{"metric"' $out/r05_bench_default.log > $out/r05_bench_line.json; cut -c1-260 $out/r05_bench_line.json
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | cut -c1-200; done
timeout 600 python bench.py --force-ddp --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $out/r05_bench_force_ddp_line.json; cut -c1-200 $out/r05_bench_force_ddp_line.json
for m in depth flow; do timeout 600 python bench.py --mode $m --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | cut -c1-200; done
for b in 1 2; do timeout 600 python bench.py --batch $b --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b host floor: ms_per_step', d['ms_per_step'], 'host_enqueue_ms', d['host_enqueue_ms'])"; done
python tools/instep_roofline.py $out/r05_bench_line.json > $out/r05_instep_roofline.md
# A/B of the round's switches in the step (alternating, same box)
for i in 1 2; do for v in 0 1; do DFE_WINO_WGRAD=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('DFE_WINO_WGRAD=$v', d['ms_per_step'])"; done; done > $out/r05_wgrad_step_ab.txt 2>&1; cat $out/r05_wgrad_step_ab.txt
for i in 1 2; do for v in 0 1; do DFE_SCONV=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('DFE_SCONV=$v', d['ms_per_step'])"; done; done > $out/r05_sconv_step_ab.txt 2>&1; cat $out/r05_sconv_step_ab.txt
# steady-state kernel trace of the train step
timeout 600 python bench.py --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/prof_ts
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ts -o ts -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r05_train_step_prof_bench.log 2>&1
f=$(find /tmp/prof_ts -name '*kernel_stats.csv' | head -1); t=$(find /tmp/prof_ts -name '*kernel_trace.csv' | head -1)
cp $f $out/r05_train_step_kernel_stats.csv; python tools/stats_md.py $f 25 > $out/r05_train_step_kernel_stats.md
python tools/stream_kernels.py $t 4 45 15 > $out/r05_stream_kernels.txt; python tools/step_breakdown.py $f 13 > $out/r05_step_breakdown.txt 2>&1
# the convolution kernels of this build against MIOpen + what their waves wait for
python tools/wgrad_bench.py --iters 20 > $out/r05_wgrad_bench.md 2>&1
python tools/wino_bench.py > $out/r05_wino_bench.md 2>&1
python tools/sconv_bench.py > $out/r05_sconv_bench.md 2>&1
bash tools/pmc_wino.sh r05
./tools/ubench/mfma_loop > $out/r05_mfma_loop.txt 2>&1
./tools/ubench/mfma_chain > $out/r05_mfma_chain.txt 2>&1
./tools/ubench/mfma_feed > $out/r05_mfma_feed.txt 2>&1
# PWC-side kernels: algorithmic-byte table
python tools/corr_bench.py --check > $out/r05_pwc_roofline_table.md 2>&1
ls $out | grep r05 | tr '\n' ' '
