#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): separate rocprofv3 --pmc passes over tools/wino_bench.py, merged per kernel and grid size
# by tools/pmc_by_kernel.py: what the Winograd kernels' waves wait for.   usage: bash tools/pmc_wino.sh <tag> [bench script] [kernel pattern] [name]
# (the strided convolutions: bash tools/pmc_wino.sh r05 sconv_bench.py k_sconv sconv)
TAG=${1:-r04}; BENCH=${2:-wino_bench.py}; PAT=${3:-k_wino_fwd16|k_wino_wgrad2}; NAME=${4:-wino}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
i=0; FILES=""
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pw$i -o t -- python3 $R/tools/$BENCH --iters 3 $PMC_BENCH_ARGS > /tmp/pw$i.log 2>&1
  f=$(find /tmp/pw$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/${TAG}_${NAME}_pmc_pass$i.csv; FILES="$FILES $OUT/${TAG}_${NAME}_pmc_pass$i.csv"; else tail -5 /tmp/pw$i.log; fi
done
python3 $R/tools/pmc_by_kernel.py "$PAT" $FILES > $OUT/${TAG}_${NAME}_pmc.txt
