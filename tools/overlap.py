#!/usr/bin/env python3
"""How busy and how concurrent the GPU is inside one training step: union of the kernel intervals, sum of their durations
and the time spent with 1 / 2 / 3+ kernels in flight, between consecutive k_geom_point_fwd launches of a rocprofv3
kernel trace.   python tools/overlap.py <kernel_trace.csv> [skip_steps]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
marks = [s for s, e, n in rows if "k_geom_point_fwd" in n]
if len(marks) < skip + 2:
    sys.exit("not enough steps in the trace")
steps = list(zip(marks[skip:-1], marks[skip + 1:]))
tot = {"wall": 0, "busy": 0, "sum": 0, "c1": 0, "c2": 0, "c3": 0, "n": 0}
for a, b in steps:
    ev = []
    for s, e, n in rows:
        if e <= a or s >= b:
            continue
        s, e = max(s, a), min(e, b)
        ev.append((s, 1)); ev.append((e, -1))
        tot["sum"] += e - s
        tot["n"] += 1
    ev.sort()
    depth, last = 0, a
    for t, d in ev:
        if depth >= 1:
            tot["busy"] += t - last
            tot["c%d" % min(depth, 3)] += t - last
        last = t
        depth += d
    tot["wall"] += b - a
n = len(steps)
print("steps %d | wall %.2f ms/step | GPU busy (union) %.2f ms = %.1f %% | sum of kernel durations %.2f ms | kernels/step %.0f" % (
    n, tot["wall"] / n / 1e6, tot["busy"] / n / 1e6, 100.0 * tot["busy"] / tot["wall"], tot["sum"] / n / 1e6, tot["n"] / n))
print("time with 1 kernel in flight %.2f ms, 2: %.2f ms, 3+: %.2f ms (per step)" % (tot["c1"] / n / 1e6, tot["c2"] / n / 1e6, tot["c3"] / n / 1e6))
