"""rocprofv3 kernel_stats.csv -> markdown: the top-N rows by total time, then every kernel of this build (dfe::)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2])
def short(n):
    n = re.sub(r"\s+", " ", n)
    n = re.sub(r"\(.*", "", n) if "dfe::" in n else n
    return n.replace("void ", "")[:110]
def table(rs):
    print("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
    for r in rs:
        print("| %s | %s | %.2f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
if top:
    print("Top %d kernels of the whole run (includes MIOpen's first-call search kernels of the warm-up):\n" % top)
    table(rows[:top])
    print()
print("Kernels of this build:\n")
table([r for r in rows if "dfe::" in r["Name"]])
