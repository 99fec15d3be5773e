"""kernel_stats.csv of a train-step trace -> per-step time by owner, plus the largest ATen / other kernels.
usage: python tools/step_breakdown.py <kernel_stats.csv> <steps traced>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
cat, det = {}, {}
for r in rows:
    n = r["Name"]; t = float(r["TotalDurationNs"]) / 1e6 / steps
    if "dfe::" in n:
        c = "dfe loss stack" if ("geom" in n or "prepare" in n) else ("dfe pwc ops" if ("corr" in n or "warp_flow" in n or "pwc" in n) else "dfe net glue")
    elif "transpose" in n: c = "miopen transpose"
    elif "SubTensor" in n or "OpTensor" in n: c = "miopen tensor ops"
    elif any(k in n for k in ("igemm", "Sp3Asm", "Conv", "Cijk", "gemm", "xdlops", "conv")): c = "miopen conv/gemm"
    elif "at::" in n or "elementwise" in n: c = "aten"
    else: c = "other"
    cat[c] = cat.get(c, 0) + t
    det.setdefault(c, []).append((t, n[:110], int(r["Calls"]) / steps))
tot = sum(cat.values())
for c, t in sorted(cat.items(), key=lambda x: -x[1]):
    print("%-20s %7.3f ms %5.1f%%" % (c, t, 100 * t / tot))
print("total %.3f ms" % tot)
for c in ("aten", "other", "miopen tensor ops"):
    print("--", c)
    for t, n, k in sorted(det.get(c, []), reverse=True)[:12]:
        print("  %.3f ms  %5.1f calls  %s" % (t, k, n))
