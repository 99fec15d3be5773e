#!/bin/bash
# usage: bash scratch/pmc_one.sh <tag> <pattern>  (env passes through)
TAG=$1; PAT=$2; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT; i=0; FILES=""
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/po$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/po$i -o t -- python3 $R/tools/wino_one.py $ONE_ARGS > /tmp/po$i.log 2>&1
  f=$(find /tmp/po$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" /tmp/po_pass$i.csv; FILES="$FILES /tmp/po_pass$i.csv"; else tail -5 /tmp/po$i.log; fi
done
python3 $R/tools/pmc_by_kernel.py "$PAT" $FILES > $OUT/${TAG}_pmc.txt
