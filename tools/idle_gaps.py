"""kernel_trace.csv of a train-step run -> GPU busy / idle time per step between consecutive k_geom_point_fwd launches.
usage: python tools/idle_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_geom_point_fwd" in r[2]]
spans, busys, gaps_big = [], [], []
for a, b in zip(marks[-8:-1], marks[-7:]):
    seg = rows[a:b]
    span = rows[b][0] - rows[a][0]
    busy, cur_end = 0, seg[0][0]
    for s, e, _ in seg:
        s2 = max(s, cur_end)
        if e > s2:
            busy += e - s2
        gap = s - cur_end
        if gap > 20000:
            gaps_big.append((gap / 1e3, _[:60]))
        cur_end = max(cur_end, e)
    spans.append(span / 1e6); busys.append(busy / 1e6)
print("steps measured:", len(spans))
print("span ms  :", ["%.2f" % v for v in spans])
print("busy ms  :", ["%.2f" % v for v in busys])
print("idle frac:", ["%.3f" % (1 - b / s) for s, b in zip(spans, busys)])
gaps_big.sort(reverse=True)
print("largest gaps (us, next kernel):", gaps_big[:12])
