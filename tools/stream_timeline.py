#!/usr/bin/env python3
"""Per-stream timeline of one training step from a rocprofv3 kernel trace: for every HIP stream (queue) the span it is
active in, its busy time, and a coarse picture of who runs when -- to see which branch is the step's critical path.
    python tools/stream_timeline.py <kernel_trace.csv> [skip_steps] [bin_ms]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    q = r.get("Stream_Id") or r.get("Queue_Id")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], (r.get("Queue_Id"), r.get("Stream_Id"))))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
binms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
marks = [s for s, e, n, q in rows if "k_geom_point_fwd" in n]
a, b = marks[skip], marks[skip + 1]
step = [(s, e, n, q) for s, e, n, q in rows if s >= a and s < b]
print("step wall %.2f ms, %d kernels, starts at k_geom_point_fwd (loss stack fwd -> backward -> Adam -> next forward)" % ((b - a) / 1e6, len(step)))
byq = collections.defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, n in lst)
    print("queue/stream %s: %4d kernels, active %.2f .. %.2f ms, busy %.2f ms" % (q, len(lst), (lst[0][0] - a) / 1e6, (max(e for s, e, n in lst) - a) / 1e6, busy / 1e6))
nb = int((b - a) / 1e6 / binms) + 1
qs = sorted(byq, key=lambda q: byq[q][0][0])
print("\nbusy fraction per %.1f ms bin (columns = streams in order of first use; last column = no kernel in flight)" % binms)
for k in range(nb):
    lo, hi = a + k * binms * 1e6, a + (k + 1) * binms * 1e6
    cells = []
    for q in qs:
        t = sum(max(0, min(e, hi) - max(s, lo)) for s, e, n in byq[q])
        cells.append("%4.0f%%" % (100 * t / (hi - lo)))
    ev = []
    for s, e, n, q in step:
        if e > lo and s < hi:
            ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
    ev.sort(); d = 0; last = lo; idle = 0
    for t, dd in ev:
        if d == 0: idle += t - last
        last = t; d += dd
    idle += hi - last if d == 0 else 0
    # the dominant kernel family in the bin
    fam = collections.Counter()
    for s, e, n, q in step:
        ov = max(0, min(e, hi) - max(s, lo))
        if ov: fam[n.split("(")[0].split("<")[0][:28]] += ov
    top = ", ".join("%s %.0f%%" % (n, 100 * v / (hi - lo)) for n, v in fam.most_common(2))
    print("%5.1f ms | %s | idle %3.0f%% | %s" % (k * binms, " ".join(cells), 100 * idle / (hi - lo), top))
