"""kernel_stats.csv -> kernels sorted by calls per step (tiny, frequent launches are what is left to collapse).
usage: python tools/small_kernels.py <kernel_stats.csv> <steps>"""
import csv, sys
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r["Calls"]))
tot = 0
for r in rows[:45]:
    c = int(r["Calls"]) / steps; t = float(r["TotalDurationNs"]) / 1e6 / steps
    print("%6.1f calls  %7.3f ms  avg %6.1f us  %s" % (c, t, float(r["AverageNs"]) / 1e3, r["Name"][:120]))
