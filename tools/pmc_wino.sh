#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): separate rocprofv3 --pmc passes over tools/wino_bench.py, merged per kernel and grid size
# by tools/pmc_by_kernel.py: what the Winograd kernels' waves wait for.   usage: bash tools/pmc_wino.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
i=0; FILES=""
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pw$i -o t -- python3 $R/tools/wino_bench.py --iters 3 > /tmp/pw$i.log 2>&1
  f=$(find /tmp/pw$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/${TAG}_wino_pmc_pass$i.csv; FILES="$FILES $OUT/${TAG}_wino_pmc_pass$i.csv"; else tail -5 /tmp/pw$i.log; fi
done
python3 $R/tools/pmc_by_kernel.py "k_wino_fwd16|k_wino_wgrad2" $FILES > $OUT/${TAG}_wino_pmc.txt
