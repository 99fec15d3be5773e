"""Round-4 experiment: a copy of the shipped MIOpen user databases in which the (VALU) Winograd entries are
dropped wherever an MFMA implicit GEMM was measured within <ratio> of them, so that one stream's matrix-core kernels could
co-issue with the other stream's vector-ALU Winograd kernels.  Run the bench with MIOPEN_USER_DB_PATH=<outdir>.
Result: no gain (24.56-24.62 ms at ratio 1.06 against 24.62-24.77; worse above).

    python tools/miopen_mix_db.py <ratio> <outdir>"""
import os, shutil, sys
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unsupervised_depth_opticalflow_egomotion_amd", "miopen_db")
thr, out = float(sys.argv[1]), sys.argv[2]
os.makedirs(out, exist_ok=True)
n = 0; dw = dg = 0.0
for f in os.listdir(src):
    if not f.endswith(".ufdb.txt"):
        shutil.copy(os.path.join(src, f), out); continue
    lines = []
    for l in open(os.path.join(src, f)):
        key, _, val = l.rstrip("\n").partition("=")
        ents = [e for e in val.split(";") if ":" in e]
        t = {e.split(":")[0]: float(e.split(":")[1].split(",")[0]) for e in ents}
        w = min([v for k, v in t.items() if "Winograd" in k], default=None)
        g = min([v for k, v in t.items() if "ImplicitGemm" in k], default=None)
        if key.split("-")[-1] in ("F", "B") and w is not None and g is not None and w <= min(t.values()) + 1e-12 and g / w <= thr:
            val = ";".join(e for e in ents if "Winograd" not in e); n += 1; dw += w; dg += g
        lines.append(key + "=" + val + "\n")
    open(os.path.join(out, f), "w").writelines(lines)
print("ratio %.2f: %d problems switched to implicit GEMM (%.2f -> %.2f ms, one call each)" % (thr, n, dw, dg))
