#!/usr/bin/env python3
"""GPU timeline of ONE training step from a rocprofv3 kernel trace (best taken with ``bench.py --graph``: a replayed hipGraph has no
per-launch host cost, so the trace shows the device's own schedule): how long the step is, how long NO kernel / exactly one / two /
three or more kernels are resident, and how much of the time the resident kernels are "wide" (>= 256 workgroups: able to fill
the chip) -- i.e. where the step is a queue of chip-filling kernels and where it is a chain of small launches.
    python tools/timeline_occupancy.py <kernel_trace.csv> [skip_steps=3]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    wx = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
    gy = int(r.get("Grid_Size_Y") or 1); gz = int(r.get("Grid_Size_Z") or 1)
    wy = int(r.get("Workgroup_Size_Y") or 1); wz = int(r.get("Workgroup_Size_Z") or 1)
    nwg = max(1, gx // max(wx, 1)) * max(1, gy // max(wy, 1)) * max(1, gz // max(wz, 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], nwg))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marks = [s for s, e, n, g in rows if "k_geom_point_fwd" in n]
a, b = marks[skip], marks[skip + 1]
step = [r for r in rows if a <= r[0] < b]
ev = []
for s, e, n, g in step:
    ev.append((s, 1, g >= 256)); ev.append((min(e, b), -1, g >= 256))
ev.sort()
t = a; cnt = wide = 0
hist = {}; wide_t = 0; narrow_only = 0
for ts, d, w in ev:
    dt = ts - t
    if dt > 0:
        hist[min(cnt, 3)] = hist.get(min(cnt, 3), 0) + dt
        if wide > 0: wide_t += dt
        elif cnt > 0: narrow_only += dt
    t = ts; cnt += d; wide += d if w else 0
tot = b - a
print("step %.2f ms, %d kernels" % (tot / 1e6, len(step)))
for k in sorted(hist):
    print("  %s resident kernels: %6.2f ms (%4.1f %%)" % (("%d" % k) if k < 3 else ">=3", hist[k] / 1e6, 100.0 * hist[k] / tot))
print("  at least one WIDE kernel (>= 256 workgroups) resident: %.2f ms (%.1f %%)" % (wide_t / 1e6, 100.0 * wide_t / tot))
print("  only narrow kernels resident: %.2f ms (%.1f %%); nothing resident: %.2f ms" % (narrow_only / 1e6, 100.0 * narrow_only / tot, hist.get(0, 0) / 1e6))
# the longest stretches without a wide kernel
t = a; cnt = wide = 0; cur = None; gaps = []
names = {}
for ts, d, w in ev:
    if wide == 0 and cur is None: cur = t
    if cur is not None and (wide > 0):
        pass
    t = ts
    prev_wide = wide
    cnt += d; wide += d if w else 0
    if prev_wide == 0 and wide > 0 and cur is not None:
        gaps.append((ts - cur, cur)); cur = None
    if prev_wide > 0 and wide == 0:
        cur = ts
gaps.sort(reverse=True)
print("  longest stretches without a wide kernel (ms, offset in step ms):", ["%.2f@%.2f" % (g / 1e6, (c - a) / 1e6) for g, c in gaps[:8]])
