#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): separate rocprofv3 --pmc passes over tools/corr_bench.py --only-corr, merged per
# kernel and grid size by tools/pmc_by_kernel.py.   usage: bash tools/pmc_corr.sh <tag>   -> gpurun_out/<tag>_corr_pmc.txt
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
i=0; FILES=""
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pc$i -o t -- python3 $R/tools/corr_bench.py --only-corr --iters 5 > /tmp/pc$i.log 2>&1
  f=$(find /tmp/pc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/${TAG}_corr_pmc_pass$i.csv; FILES="$FILES $OUT/${TAG}_corr_pmc_pass$i.csv"; else tail -5 /tmp/pc$i.log; fi
done
python3 $R/tools/pmc_by_kernel.py "k_corr" $FILES > $OUT/${TAG}_corr_pmc.txt
