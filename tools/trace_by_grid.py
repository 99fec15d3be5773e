"""rocprofv3 kernel_trace.csv -> average duration per (kernel, grid size): tells the layers of one kernel apart.
usage: python tools/trace_by_grid.py <kernel_trace.csv> [name filter regex]"""
import collections, csv, re, sys
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if pat and not pat.search(n):
        continue
    short = re.sub(r"\(.*", "", n).replace("void ", "")[:60]
    grid = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    acc[(short, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items()):
    v = sorted(v)
    print("%-60s grid %-22s n=%4d  median %.1f us  min %.1f" % (k, "x".join(g), len(v), v[len(v) // 2], v[0]))
