#!/usr/bin/env python3
"""Counter value / known bytes for every kernel of tools/ubench/fetch_calib (see tools/fetch_calib.sh)."""
import csv, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
GiB = 1 << 30
H, W = 256, 832
planes = (256 << 20) // (H * W)
plane_bytes = planes * H * W * 4
KNOWN = [   # (kernel substring, occurrence, counter, bytes moved, description)
    ("k_read_stream<4>", 0, "FETCH_SIZE", GiB, "coalesced read, 16 B / lane"),
    ("k_read_stream<2>", 0, "FETCH_SIZE", GiB, "coalesced read, 8 B / lane"),
    ("k_read_stream<1>", 0, "FETCH_SIZE", GiB, "coalesced read, 4 B / lane (256-B wave rows)"),
    ("k_read_strips", 0, "FETCH_SIZE", plane_bytes, "rolling-stencil shape: 62-column strips, dword / lane, 8 rows + 2 halo rows per wave (re-reads: 2/62 columns, 2/8 rows)"),
    ("k_read_pairs", 0, "FETCH_SIZE", plane_bytes, "8-byte pair gathers at the own pixel, 2 rows (each byte requested ~4x)"),
    ("k_read_pairs", 1, "FETCH_SIZE", plane_bytes, "8-byte pair gathers, footprint shifted by (3, 3)"),
    ("k_write_stream<4>", 0, "WRITE_SIZE", GiB, "coalesced write, 16 B / lane"),
    ("k_write_stream<1>", 0, "WRITE_SIZE", GiB, "coalesced write, 4 B / lane"),
    ("k_write_bytes", 0, "WRITE_SIZE", GiB // 4, "coalesced write, 1 B / lane (mask pack)"),
]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    seen = {}
    for r in csv.DictReader(open(os.path.join(OUT, "%s_fetch_calib_%s.csv" % (TAG, c)))):
        if r["Counter_Name"] != c:
            continue
        n = r["Kernel_Name"]
        k = seen.get(n, 0); seen[n] = k + 1
        vals[(n, k, c)] = (float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# %s: FETCH_SIZE / WRITE_SIZE calibration on gfx950 (1 GiB buffer = 4x the Infinity Cache, every byte moved once)\n" % TAG)
print("| kernel | access shape | known MB | counter (KB) x 1024 / known | us | GB/s |")
print("|---|---|---|---|---|---|")
for sub, occ, c, nbytes, desc in KNOWN:
    hit = [(k, v) for k, v in vals.items() if sub in k[0] and k[1] == occ and k[2] == c]
    if not hit:
        print("| %s | %s | %.0f | missing | | |" % (sub, desc, nbytes / 1e6)); continue
    (name, _, _), (v, us) = hit[0]
    print("| %s #%d | %s | %.0f | %s = %.3f | %.0f | %.0f |" % (sub, occ, desc, nbytes / 1e6, c, v * 1024 / nbytes, us, nbytes / us / 1e3))
