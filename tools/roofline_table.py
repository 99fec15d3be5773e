#!/usr/bin/env python3
"""Merge rocprofv3 kernel stats + PMC passes (tools/pmc_loss_stack.sh) with the algorithmic byte models of the fused
loss stack into <tag>_pmc_loss_stack.json and <tag>_roofline_table.md (under gpurun_out/; copy into profiles/).

Byte models (fp32, B=4, 256x832, S=3 unless overridden): the reads and writes a kernel cannot avoid, per launch.
PMC traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE is in KB and on gfx950 counts half of a
coalesced read stream (MI355X_MICROARCH.md, HBM section); dword-per-lane streams are uncalibrated (treat as +-2x on the
read side); the 90 MB working set sits in the 256 MB Infinity Cache, so fabric-side counters also see cache hits."""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import byte_models  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
B, H, W, S = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (4, 256, 832, 3)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
MODEL = byte_models.models(B, H, W, S)   # kernel -> (bytes per launch, formula)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0].replace("dfe::", "").strip()


stats = {}
p = os.path.join(OUT, TAG + "_loss_stack_kernel_stats.csv")
for r in csv.DictReader(open(p)):
    if "dfe::" in r["Name"]:
        k = short(r["Name"])
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        a = stats.setdefault(k, [0, 0.0]); a[0] += calls; a[1] += tot
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, 10):
    f = os.path.join(OUT, "%s_pmc_pass%d.csv" % (TAG, i))
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if "dfe::" in r["Kernel_Name"]:
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "workload": {"batch": B, "height": H, "width": W, "scales": S}, "kernels": {}}
rows = []
for k in sorted(stats, key=lambda q: -stats[q][1]):
    calls, tot = stats[k]
    us = tot / calls / 1e3
    c = {n: sum(v) / len(v) for n, v in pmc.get(k, {}).items()}
    e = {"avg_us": round(us, 2), "calls": calls}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = round(c["FETCH_SIZE"], 1), round(c["WRITE_SIZE"], 1)
        e["pmc_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        e["waves"] = int(c["SQ_WAVES"])
        for n, key in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave"), ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave"),
                       ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave"), ("SQ_INSTS_LDS", "lds_per_wave")):
            if n in c:
                e[key] = round(c[n] / c["SQ_WAVES"], 1)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        for n, key in (("SQ_WAIT_ANY", "frac_wait_mem"), ("SQ_WAIT_INST_ANY", "frac_wait_issue"), ("SQ_ACTIVE_INST_ANY", "frac_active")):
            if n in c:
                e[key] = round(c[n] / c["SQ_WAVE_CYCLES"], 3)
    if "TA_BUSY_avr" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
        e["ta_busy_frac"] = round(c["TA_BUSY_avr"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if "TCC_HIT_sum" in c and (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0)):
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
    if k in MODEL:
        e["algorithmic_bytes"], e["byte_model"] = MODEL[k]
        e["achieved_GBs"] = round(MODEL[k][0] / (us * 1e-6) / 1e9, 1)
        e["frac_of_8TBs"] = round(e["achieved_GBs"] / 8000.0, 3)
        if "pmc_bytes" in e:
            e["pmc_over_algorithmic"] = round(e["pmc_bytes"] / MODEL[k][0], 2)
    res["kernels"][k] = e
    rows.append((k, e))
json.dump(res, open(os.path.join(OUT, TAG + "_pmc_loss_stack.json"), "w"), indent=1)
with open(os.path.join(OUT, TAG + "_roofline_table.md"), "w") as fh:
    fh.write("# %s: fused loss stack, per-kernel roofline table (loss_stack workload, B=%d %dx%d S=%d, idle-GPU loop)\n\n" % (TAG, B, H, W, S))
    fh.write("`rocprofv3 --kernel-trace --stats` + separate `--pmc` passes (tools/pmc_loss_stack.sh); peak 8 TB/s (spec).\n"
             "PMC bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction; calibration of the other access shapes: "
             "profiles/r03_fetch_calib.md).  Working set of the stack: %.0f MB (Infinity Cache: 256 MiB).\n\n" % (sum(v[0] for v in MODEL.values()) / 1e6 / 2))
    fh.write("| kernel | avg us | algorithmic MB | achieved GB/s | frac of 8 TB/s | PMC MB | PMC / alg | VALU / wave | VMEM rd+wr / wave | TA busy | wait-mem | wait-issue |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, e in rows:
        fh.write("| %s | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
            k, e["avg_us"], "%.1f" % (e["algorithmic_bytes"] / 1e6) if "algorithmic_bytes" in e else "-",
            e.get("achieved_GBs", "-"), e.get("frac_of_8TBs", "-"), "%.1f" % (e["pmc_bytes"] / 1e6) if "pmc_bytes" in e else "-",
            e.get("pmc_over_algorithmic", "-"), e.get("valu_per_wave", "-"),
            "%s+%s" % (e.get("vmem_rd_per_wave", "-"), e.get("vmem_wr_per_wave", "-")), e.get("ta_busy_frac", "-"),
            e.get("frac_wait_mem", "-"), e.get("frac_wait_issue", "-")))
    fh.write("\nByte models:\n\n")
    for k, (bts, why) in MODEL.items():
        if k in stats:
            fh.write("* `%s`: %.1f MB = %s\n" % (k, bts / 1e6, why))
# the traffic figure bench.py may quote for k_geom_point_fwd: tied to the kernel sources it was measured on
e = res["kernels"].get("k_geom_point_fwd", {})
if "pmc_bytes" in e:
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(OUT), "unsupervised_depth_opticalflow_egomotion_amd", "csrc")
    for name in ("loss_stack_fwd.hip", "loss_stack.h", "loss_stack_exact.h", "dfe_device.h"):   # == bench.KERNEL_SOURCES
        h.update(open(os.path.join(csrc, name), "rb").read())
    json.dump({"note": "HBM-side bytes per launch of k_geom_point_fwd from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, "
                       "tools/pmc_loss_stack.sh): (2*FETCH_SIZE + WRITE_SIZE)*1024. bench.py reports it as roofline.traffic only "
                       "when its workload AND the sha256 of the kernel sources match.",
               "kernel": "k_geom_point_fwd", "kernel_source_sha256": h.hexdigest(),
               "workload": {"batch": B, "height": H, "width": W, "scales": S},
               "FETCH_SIZE_KB": e["FETCH_SIZE_KB"], "WRITE_SIZE_KB": e["WRITE_SIZE_KB"], "hbm_bytes_per_launch": e["pmc_bytes"],
               "waves": e.get("waves"), "valu_per_wave": e.get("valu_per_wave"),
               "algorithmic_bytes_per_launch": e.get("algorithmic_bytes"), "source": TAG + "_pmc_loss_stack.json"},
              open(os.path.join(OUT, TAG + "_pmc_point_fwd_traffic.json"), "w"), indent=1)
print(open(os.path.join(OUT, TAG + "_roofline_table.md")).read())
