#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + separate --pmc passes of the loss_stack workload,
# then tools/roofline_table.py merges them with the algorithmic byte models into profiles-ready files under gpurun_out/.
# usage: bash tools/pmc_loss_stack.sh <tag> [B H W S]      (writes gpurun_out/<tag>_*; default 4 256 832 3)
TAG=${1:-r03}
B=${2:-4}; H=${3:-256}; W=${4:-832}; S=${5:-3}
SHAPE="--batch $B --height $H --width $W --scales $S"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
CMD="python3 $R/bench.py --workload loss_stack $SHAPE --steps 30 --warmup 5 --no-cpu-baseline"
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o t -- $CMD > /tmp/ks.log 2>&1
cp /tmp/ks/t_kernel_stats.csv $OUT/${TAG}_loss_stack_kernel_stats.csv
grep '^{"metric"' /tmp/ks.log > $OUT/${TAG}_loss_stack_bench_line.json
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm$i -o t -- python3 $R/bench.py --workload loss_stack $SHAPE --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pm$i.log 2>&1
  f=$(find /tmp/pm$i -name "*counter_collection.csv" | head -1)
  cp "$f" $OUT/${TAG}_pmc_pass$i.csv
done
cd $R && python3 tools/roofline_table.py $TAG $B $H $W $S
