#!/bin/bash
# Transformed filters kept across steps (ops.WinoWeightCache) against the per-call transform: the Winograd tests, then the
# train step alternating DFE_WINO_CACHE=0 / 1 on the same box.  Run from the repository root on a GPU box.
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_hip_wino.py -x -q 2>&1 | tail -3 > gpurun_out/wc_tests.txt
tail -2 gpurun_out/wc_tests.txt
for i in 1 2 3; do
  for c in 0 1; do
    DFE_WINO_CACHE=$c DFE_WINO_CACHE_STATS=1 timeout 150 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/wc_b${c}_$i.json 2> gpurun_out/wc_b${c}_$i.err
    python -c "import json,sys;d=json.loads(open('gpurun_out/wc_b${c}_$i.json').read().strip().splitlines()[-1]);print('cache=$c', d['ms_per_step'])"
    grep "wino weight cache" gpurun_out/wc_b${c}_$i.err
  done
done
