#!/bin/bash
# Proxy for a fused point + SSIM forward kernel ON THE GPU BOX: what the two kernels cost
#   (a) at 1.19x the pixels -- the halo a 32x16 block tile would recompute (width 992 against 832);
#   (b) with the SSIM kernels' loads taken away (a scratch build: the operands come out of the pixel index; arithmetic, DPP
#       window sums and stores stay) -- what a fused kernel that reads its window from LDS would still pay.
# A fused kernel costs at least (point at 1.19x) + (SSIM without loads); compare with the pair today.  usage: bash tools/fuse_proxy.sh
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/csrc_abl /tmp/incl_abl; mkdir -p /tmp/abl/pkg /tmp/abl/include
cp -r $R/unsupervised_depth_opticalflow_egomotion_amd/csrc /tmp/abl/pkg/csrc; cp $R/include/dfe_hip.h /tmp/abl/include/; rm -f /tmp/abl/pkg/csrc/*.o
python3 - <<'PY'
p = '/tmp/abl/pkg/csrc/loss_stack.h'
s = open(p).read()
old = '''  const float wq = ssim_weight_at(mk, need, q);
  float ta[3], tb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { ta[c] = it[q + c * N]; tb[c] = yw[q + c * N]; }'''
new = '''  const float wq = (q & 7) ? 1.0f : 0.0f;
  float ta[3], tb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { ta[c] = __int_as_float(0x3f000000 + ((q * 3 + c) & 0xffff)); tb[c] = __int_as_float(0x3f100000 + ((q * 5 + c) & 0xffff)); }'''
assert old in s
open(p, 'w').write(s.replace(old, new))
PY
(cd /tmp/abl/pkg/csrc && sed -i 's#\.\./\.\./include/dfe_hip.h#../../include/dfe_hip.h#' Makefile dfe_internal.h && make -j8 > /tmp/abl/build.log 2>&1) || { tail -5 /tmp/abl/build.log; exit 1; }
cd /tmp
run() {   # tag, width, library
  rm -rf /tmp/fp_$1
  DFE_HIP_LIB=$3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$1 -o t -- python3 $R/bench.py --workload loss_stack --width $2 --steps 60 --warmup 5 --no-cpu-baseline > /tmp/fp_$1.log 2>&1
  f=$(find /tmp/fp_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1 (B = 4, 256 x $2, S = 3; rocprofv3 kernel averages over 65 launches)"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_geom_point_fwd', 'k_geom_ssim_fwd_roll', 'k_geom_point_bwd', 'k_geom_ssim_bwd_roll')):
        print("   %-34s calls %5s  avg %7.1f us" % (n.split('(')[0].replace('void ', '').replace('dfe::', ''), r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
run base 832 $R/unsupervised_depth_opticalflow_egomotion_amd/libdfe_hip.so
run wide_1.19x 992 $R/unsupervised_depth_opticalflow_egomotion_amd/libdfe_hip.so
run ssim_without_loads 832 /tmp/abl/pkg/libdfe_hip.so
run base_again 832 $R/unsupervised_depth_opticalflow_egomotion_amd/libdfe_hip.so
