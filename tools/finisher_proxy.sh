#!/bin/bash
# Upper bound of what removing the small finishing launches could buy, measured BEFORE building anything:
# a copy of csrc/ under /tmp whose launchers skip them (DFE_PROXY_SKIP=1: k_wino_sum, k_planeconv_finish, k_bias_grad_final(_multi),
# k_glue_bias_final, k_head_final; DFE_PROXY_SKIP_BIG=1: the weight gradients' split sums) -- results are garbage, timing is valid
# (Adam bounds every update by lr).  Alternating runs on one box.   usage (on the GPU box): bash tools/finisher_proxy.sh
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pf; mkdir -p /tmp/pf/pkg /tmp/pf/include
cp -r $R/unsupervised_depth_opticalflow_egomotion_amd/csrc /tmp/pf/pkg/csrc; cp $R/include/dfe_hip.h /tmp/pf/include/; rm -f /tmp/pf/pkg/csrc/*.o
python3 - <<'PY'
import re, glob
small = [r'k_glue_bias_final<<<', r'k_head_final<\d, \w+><<<', r'k_bias_grad_final<<<', r'k_bias_grad_final_multi<<<', r'k_planeconv_finish<<<', r'k_wino_sum<<<']
for f in glob.glob('/tmp/pf/pkg/csrc/*.hip'):
    s = open(f).read(); o = s
    for p in small:
        s = re.sub(r'(?<![\w_])(' + p + ')', r'if (!getenv("DFE_PROXY_SKIP")) \1', s)
    s = s.replace('if (p.S > 64) k_wgrad_sum<32>', 'if (getenv("DFE_PROXY_SKIP_BIG")) {} else if (p.S > 64) k_wgrad_sum<32>')
    s = s.replace('if (ns > 64) k_wgrad_sum<32>', 'if (getenv("DFE_PROXY_SKIP_BIG")) {} else if (ns > 64) k_wgrad_sum<32>')
    if s != o:
        open(f, 'w').write(('#include <cstdlib>\n' if '#include <cstdlib>' not in s else '') + s)
PY
(cd /tmp/pf/pkg/csrc && make -j8 > /tmp/pf/build.log 2>&1) || { tail -5 /tmp/pf/build.log; exit 1; }
L=/tmp/pf/pkg/libdfe_hip.so
p() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', j['ms_per_step'], 'host_enqueue_ms', j.get('host_enqueue_ms'))"; }
cd $R
for i in 1 2; do
  python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | p "all launches          "
  DFE_PROXY_SKIP=1 DFE_HIP_LIB=$L python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | p "without small finishers"
  DFE_PROXY_SKIP_BIG=1 DFE_HIP_LIB=$L python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | p "without wgrad split sums"
done
