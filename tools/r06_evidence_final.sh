#!/bin/bash
# Round-6 evidence refresh ON THE GPU BOX after the last kernel changes (SSIM / point backward, hardware-queue default): the parts of
# tools/r06_evidence.sh whose subjects changed.  The Winograd tables, their counters and the finisher proxy are unchanged kernels.
# usage: bash tools/r06_evidence_final.sh [skip_tests]
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
if [ -z "$1" ]; then
  timeout 3000 python -m pytest tests -m gpu -q > $out/r06_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/r06_pytest_gpu.log; tail -3 $out/r06_pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r06_smoke.log 2>&1; tail -1 $out/r06_smoke.log
fi
bash tools/pmc_loss_stack.sh r06 > /dev/null 2>&1
cp $out/r06_pmc_point_fwd_traffic.json profiles/pmc_point_fwd_traffic.json 2>/dev/null
bash tools/pmc_loss_stack.sh r06_config5_b2 2 375 1242 6 > /dev/null 2>&1
bash tools/pmc_loss_stack.sh r06_config5_b16 16 375 1242 6 > /dev/null 2>&1
timeout 900 python bench.py > $out/r06_bench_default.log 2>&1; grep '^{"metric"' $out/r06_bench_default.log > $out/r06_bench_line.json; cut -c1-260 $out/r06_bench_line.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > $out/r06_bench_driver_cmd_line.json; cut -c1-200 $out/r06_bench_driver_cmd_line.json
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | cut -c1-200; done
timeout 600 python bench.py --force-ddp --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $out/r06_bench_force_ddp_line.json; cut -c1-200 $out/r06_bench_force_ddp_line.json
for m in depth flow; do timeout 600 python bench.py --mode $m --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | cut -c1-200; done
{ for b in 1 2 4; do for g in "" "--graph"; do timeout 600 python bench.py --batch $b $g --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b ${g:-eager}: ms_per_step', d['ms_per_step'], 'host_enqueue_ms', d['host_enqueue_ms'])"; done; done; } > $out/r06_graph_mode.txt 2>&1; cat $out/r06_graph_mode.txt
python tools/instep_roofline.py $out/r06_bench_line.json > $out/r06_instep_roofline.md
rm -rf /tmp/prof_ts
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ts -o ts -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r06_train_step_prof_bench.log 2>&1
f=$(find /tmp/prof_ts -name '*kernel_stats.csv' | head -1); t=$(find /tmp/prof_ts -name '*kernel_trace.csv' | head -1)
cp $f $out/r06_train_step_kernel_stats.csv; python tools/stats_md.py $f 25 > $out/r06_train_step_kernel_stats.md
python tools/stream_kernels.py $t 4 45 15 > $out/r06_stream_kernels.txt; python tools/step_breakdown.py $f 13 > $out/r06_step_breakdown.txt 2>&1
python tools/parity_report.py > $out/r06_parity_report.txt 2>&1; tail -5 $out/r06_parity_report.txt
ls $out | grep r06 | tr '\n' ' '
