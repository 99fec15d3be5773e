#!/usr/bin/env python3
"""Price the fused loss stack's launches at their IN-STEP durations: the HIP-event segment times `bench.py` records inside
its timed region (`kernel_ms` of the JSON line) against the algorithmic byte models of tools/byte_models.py -- the same
arithmetic as `roofline.frac` of the bench line, for every segment, so that profiles/ reproduces the bench number
(tools/roofline_table.py prices the idle-GPU loop with per-pixel random flows instead).

    python tools/instep_roofline.py <bench_line.json> [batch height width scales]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import byte_models  # noqa: E402

FWD = [("prep", "k_geom_prepare"), ("pyramids", "k_geom_pyramids12"), ("point_fwd", "k_geom_point_fwd"), ("ssim_fwd", "k_geom_ssim_fwd_roll"),
       ("flow_smooth", "k_geom_flow_smooth_fwd"), ("disp_smooth", "k_geom_disp_smooth_fwd"), ("reduce + assemble", None)]
BWD = [("ssim_bwd", "k_geom_ssim_bwd_roll"), ("point_bwd", "k_geom_point_bwd"), ("flow_smooth_bwd", "k_geom_flow_smooth_bwd"),
       ("disp_smooth_bwd1", "k_geom_disp_smooth_bwd1"), ("disp_smooth_bwd2", "k_geom_disp_smooth_bwd2"), ("pose_finalize", None)]


def main():
    line = None
    for l in open(sys.argv[1]):
        if l.lstrip().startswith('{"metric"'):
            line = json.loads(l)
    B, H, W, S = (int(x) for x in sys.argv[2:6]) if len(sys.argv) >= 6 else (4, 256, 832, 3)
    model = byte_models.models(B, H, W, S)
    km = line["kernel_ms"]
    print("# fused loss stack at its in-step durations (HIP events inside bench.py's timed region)\n")
    print("bench line: %s, %.3f ms/step, %d steps; roofline.frac %.4f (bound: %s, valu_frac %s)\n" % (
        line["config"]["workload"], line["ms_per_step"], line["steps"], line["roofline"]["frac"], line["roofline"]["bound"],
        line["roofline"].get("valu_frac")))
    print("| segment | kernel | in-step us | algorithmic MB | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|")
    for names, times in ((FWD, km["fwd_ms"]), (BWD, km["bwd_ms"])):
        for (seg, kern), ms in zip(names, times):
            if kern in model:
                nbytes = model[kern][0]
                print("| %s | `%s` | %.1f | %.1f | %.0f | %.3f |" % (seg, kern, ms * 1e3, nbytes / 1e6, nbytes / (ms * 1e-3) / 1e9, nbytes / (ms * 1e-3) / 8e12))
            else:
                print("| %s | %s | %.1f | - | - | - |" % (seg, "`%s`" % kern if kern else "-", ms * 1e3))
    f, b = sum(km["fwd_ms"]), sum(km["bwd_ms"])
    print("\nforward %.1f us + backward %.1f us = %.1f us per step (segments include the gaps between launches)" % (f * 1e3, b * 1e3, (f + b) * 1e3))
    pair = (model["k_geom_point_fwd"][0] + model["k_geom_ssim_fwd_roll"][0]) / ((km["fwd_ms"][2] + km["fwd_ms"][3]) * 1e-3) / 8e12
    print("warp + SSIM forward pair: %.3f of the HBM peak" % pair)


if __name__ == "__main__":
    main()
