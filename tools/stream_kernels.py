#!/usr/bin/env python3
"""What runs on each HIP stream during one training step (rocprofv3 kernel trace): per stream the kernels grouped by
name with count, total and mean duration, plus the share of launches shorter than a threshold -- what a fused
small-plane kernel or a merged launch could remove from the step's long pole (the flow stream).
    python tools/stream_kernels.py <kernel_trace.csv> [skip_steps=4] [top=40] [short_us=15]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], (r.get("Queue_Id"), r.get("Stream_Id")),
                 r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
short = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0
marks = [s for s, e, n, q, g, w in rows if "k_geom_point_fwd" in n]
a, b = marks[skip], marks[skip + 1]
step = [r for r in rows if a <= r[0] < b]
print("step wall %.2f ms, %d kernels" % ((b - a) / 1e6, len(step)))


def short_name(n):
    n = re.sub(r"\s+", " ", n).replace("void ", "").replace("dfe::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:90]


byq = collections.defaultdict(list)
for s, e, n, q, g, w in step:
    byq[q].append((s, e, n, g, w))
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, *_ in kv[1])):
    busy = sum(e - s for s, e, *_ in lst) / 1e3
    sh = [(e - s) / 1e3 for s, e, *_ in lst if (e - s) / 1e3 < short]
    print("\n=== stream %s: %d kernels, busy %.2f ms; %d launches < %.0f us = %.2f ms" % (q, len(lst), busy / 1e3, len(sh), short, sum(sh) / 1e3))
    agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for s, e, n, g, w in lst:
        k = short_name(n)
        d = (e - s) / 1e3
        agg[k][0] += 1; agg[k][1] += d
        if d < short:
            agg[k][2] += 1; agg[k][3] += d
    for k, (c, t, cs, ts) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%8.1f us %4d x %7.1f | short: %3d = %6.1f us | %s" % (t, c, t / c, cs, ts, k))
