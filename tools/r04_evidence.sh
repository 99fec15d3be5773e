#!/bin/bash
# Round-4 evidence run ON THE GPU BOX (through gpurun): everything the DESIGN / profiles of the round quote, into gpurun_out/r04_*.
# usage: bash tools/r04_evidence.sh [skip_tests]
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
if [ -z "$1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $out/r04_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/r04_pytest_gpu.log; tail -3 $out/r04_pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r04_smoke.log 2>&1; tail -1 $out/r04_smoke.log
fi
# the default bench line (with the CPU baseline), twice more without it, the forced-DDP line and the opt-in bf16 line
timeout 900 python bench.py > $out/r04_bench_default.log 2>&1; grep '^{"metric"' $out/r04_bench_default.log > $out/r04_bench_line.json; cut -c1-200 $out/r04_bench_line.json
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | cut -c1-180; done
timeout 600 python bench.py --force-ddp --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $out/r04_bench_force_ddp_line.json; cut -c1-160 $out/r04_bench_force_ddp_line.json
timeout 600 python bench.py --amp bf16 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $out/r04_bench_bf16_line.json; cut -c1-200 $out/r04_bench_bf16_line.json
python tools/instep_roofline.py $out/r04_bench_line.json > $out/r04_instep_roofline.md
# steady-state kernel trace of the train step
rm -rf /tmp/prof_ts
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ts -o ts -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/r04_train_step_prof_bench.log 2>&1
f=$(find /tmp/prof_ts -name '*kernel_stats.csv' | head -1); t=$(find /tmp/prof_ts -name '*kernel_trace.csv' | head -1)
cp $f $out/r04_train_step_kernel_stats.csv; python tools/stats_md.py $f 25 > $out/r04_train_step_kernel_stats.md
python tools/stream_kernels.py $t 4 45 15 > $out/r04_stream_kernels.txt; python tools/step_breakdown.py $f 13 > $out/r04_step_breakdown.txt 2>&1
# PWC-side kernels: algorithmic-byte table + PMC traffic
python tools/corr_bench.py --check > $out/r04_pwc_roofline_table.md 2>&1
bash tools/pmc_corr.sh r04
# fused Winograd-on-MFMA convolution against MIOpen on the nets' large layers
python tools/wino_bench.py > $out/r04_wino_bench.md 2>&1
# small-plane convolutions: HIP-event time per call against MIOpen + kernel durations by grid
python tools/planeconv_bench.py --levels 6,5 > $out/r04_planeconv_bench.md 2>&1
rm -rf /tmp/pcv; rocprofv3 --kernel-trace --output-format csv -d /tmp/pcv -o t -- python3 tools/planeconv_bench.py --levels 6,5 --iters 10 --reps 1 > /dev/null 2>&1
python tools/trace_by_grid.py $(find /tmp/pcv -name '*kernel_trace.csv' | head -1) "planeconv|miopenSp3|igemm|batched_transpose|SubTensor" > $out/r04_planeconv_kernels.txt 2>&1
# fused loss stack: per-kernel roofline tables at the headline shape and at configs[4] (B = 2, B = 16)
bash tools/pmc_loss_stack.sh r04 > /dev/null 2>&1
bash tools/pmc_loss_stack.sh r04_config5_b2 2 375 1242 6 > /dev/null 2>&1
bash tools/pmc_loss_stack.sh r04_config5_b16 16 375 1242 6 > /dev/null 2>&1
ls $out | grep r04 | tr '\n' ' '
