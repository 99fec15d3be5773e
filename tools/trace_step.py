"""Analyse a rocprofv3 kernel trace: pick one steady-state step (between two launches of a marker kernel) and
print the GPU-busy breakdown by kernel family with full names."""
import csv, sys, collections, re
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_geom_point_fwd"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if marker in r[2]]
pairs = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1) if idx[i + 1] - idx[i] > 500]
a, b = pairs[-2]
step = rows[a:b]
wall = (step[-1][1] - step[0][0]) / 1e6
busy = sum(e - s for s, e, _ in step) / 1e6
gaps = sum(max(0, step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)) / 1e6
print("wall %.2f ms  busy %.2f ms  gaps %.2f ms  kernels %d" % (wall, busy, gaps, len(step)))
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in step:
    n = re.sub(r"\s+", " ", n)[:int(sys.argv[3]) if len(sys.argv) > 3 else 150]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print("%7.3f ms %5d  %s" % (t, c, n))

# family breakdown
fam = collections.OrderedDict([("this build (dfe::)", 0.0), ("MIOpen conv / gemm", 0.0), ("MIOpen layout transposes", 0.0), ("MIOpen batch norm + aux", 0.0), ("ATen", 0.0), ("other", 0.0)])
cnt = dict((k, 0) for k in fam)
for s_, e_, n in step:
    t = (e_ - s_) / 1e6
    if "dfe::" in n: k = "this build (dfe::)"
    elif "batched_transpose" in n: k = "MIOpen layout transposes"
    elif "BatchNorm" in n or "SubTensorOp" in n or "Im2d2Col" in n or "Col2Im" in n: k = "MIOpen batch norm + aux"
    elif n.startswith("miopen") or "igemm" in n or n.startswith("Cijk") or "naive_conv" in n or "ck::" in n or "gemm" in n.lower() or "conv" in n.lower() and "at::" not in n: k = "MIOpen conv / gemm"
    elif "at::" in n: k = "ATen"
    else: k = "other"
    fam[k] += t; cnt[k] += 1
print()
for k, v in fam.items():
    print("%-28s %7.2f ms %5.1f %%  (%d kernels)" % (k, v, 100 * v / busy, cnt[k]))
mine = collections.defaultdict(lambda: [0, 0.0])
for s_, e_, n in step:
    if "dfe::" in n:
        nm = re.sub(r"\(.*", "", n.replace("void ", "").replace("dfe::", ""))
        mine[nm][0] += 1; mine[nm][1] += (e_ - s_) / 1e3
print()
for n, (c, t) in sorted(mine.items(), key=lambda kv: -kv[1][1]):
    print("%9.1f us %4d  %s" % (t, c, n))
