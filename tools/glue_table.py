#!/usr/bin/env python3
"""Per-kernel measured traffic of the net-glue kernels (everything of this build outside the fused loss stack) in the
train step: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (gfx950: FETCH_SIZE counts half of a coalesced read,
profiles/r03_fetch_calib.md) over the launch duration -- how close each glue pass runs to the HBM roofline.  Input: the
two counter CSVs of tools/glue_traffic.sh (counter collection serialises the launches, so durations are per kernel)."""
import collections, csv, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
LOSS_STACK = ("k_geom_", "k_depth_point", "k_flow_point", "k_prepare")


def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n.replace("dfe::", "").strip()


agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": [], "us": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(OUT, "%s_glue_%s.csv" % (TAG, c)))):
        if "dfe::" not in r["Kernel_Name"] or r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        if k.startswith(LOSS_STACK):
            continue
        agg[k][c].append(float(r["Counter_Value"]))
        if c == "FETCH_SIZE":
            agg[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = 5.0   # 2 warm-up + 3 timed steps per pass
rows = []
for k, v in agg.items():
    if not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
        continue
    n = len(v["FETCH_SIZE"])
    mb = (2 * sum(v["FETCH_SIZE"]) / n + sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])) * 1024 / 1e6
    us = sum(v["us"]) / len(v["us"])
    rows.append((us * n / steps, k, n / steps, us, mb, mb / us * 1e3 if us else 0.0))
rows.sort(reverse=True)
print("# %s: net-glue kernels of the train step (B=4, 256x832): measured fabric traffic per launch and achieved rate\n" % TAG)
print("PMC bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; durations under counter collection (launches serialised); HBM peak 8 TB/s, plain copy 6.3 TB/s.\n")
print("| kernel | launches / step | avg us | us / step | PMC MB / launch | GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for tot, k, n, us, mb, gbs in rows[:28]:
    print("| %s | %.0f | %.1f | %.0f | %.1f | %.0f | %.2f |" % (k, n, us, tot, mb, gbs, gbs / 8000.0))
print("\nSum over all %d glue kernels: %.2f ms per step." % (len(rows), sum(r[0] for r in rows) / 1e3))
