#!/usr/bin/env python3
"""Per-layer timing of the small-plane 3x3 convolutions (csrc/ops_planeconv.hip) against the MIOpen calls they replace in
ops.DenseDecodeFn, on the decoder layers of PWC levels 6 / 5 / 4 of a 256x832 frame (B = 8 pairs).  HIP-event time per call,
median of --reps bursts of --iters calls.  MIOpen forward = convolution + dfe_bias_act_fwd2 (what the decoder runs);
planeconv forward = k_planeconv + k_planeconv_finish (bias and activation inside).

    python tools/planeconv_bench.py [--levels 6,5,4] [--iters 50] [--reps 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import convs, ops          # noqa: E402
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib       # noqa: E402

LEVEL_HW = {6: (4, 13), 5: (8, 26), 4: (16, 52), 3: (32, 104)}
LEVEL_CIN = {6: 81, 5: 211, 4: 179, 3: 147}            # 81 + pyramid channels + 2 (+ 2 up-sampled features)


def layers(lvl):
    c0 = LEVEL_CIN[lvl]
    return [(c0, 128), (128, 128), (256, 96), (224, 64), (160, 32)]


def timeit(fn, iters, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(out)[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="6,5,4")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.B
    print("| level | layer | GFLOP | fwd MIOpen+epi | fwd plane | dgrad MIOpen | dgrad plane | wgrad MIOpen | wgrad plane |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot = {}
    for lvl in [int(v) for v in a.levels.split(",")]:
        H, W = LEVEL_HW[lvl]
        sums = [0.0] * 6
        for ci, co in layers(lvl):
            if not get_lib().dfe_planeconv_supported(B, ci, co, H, W):
                continue
            x = torch.randn(B, ci, H, W, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
            bias = torch.randn(co, device=dev)
            gy = torch.randn(B, co, H, W, device=dev)
            d1 = torch.empty(B, co, H, W, device=dev)
            d2 = torch.empty(B, co + 64, H, W, device=dev)

            def mi_fwd():
                z = convs.raw_forward(x, w, 1, 1)
                ops.check(get_lib().dfe_bias_act_fwd2(ops.ptr(z), ops.ptr(bias), ops.ptr(z), z.stride(0), ops.ptr(d2), d2.stride(0),
                                                      B, co, H, W, 0.1, ops.stream_ptr()), "epi")
            t = [timeit(mi_fwd, a.iters, a.reps),
                 timeit(lambda: ops.planeconv_fwd_into(x, w, bias, 0.1, d1, 0, d2, 0), a.iters, a.reps),
                 timeit(lambda: convs.raw_backward(gy, x, w, 1, 1, 1, True, False), a.iters, a.reps),
                 timeit(lambda: ops.planeconv_backward(gy, x, w, True, False), a.iters, a.reps),
                 timeit(lambda: convs.raw_backward(gy, x, w, 1, 1, 1, False, True), a.iters, a.reps),
                 timeit(lambda: ops.planeconv_backward(gy, x, w, False, True), a.iters, a.reps)]
            gf = 2.0 * B * H * W * ci * co * 9 / 1e9
            print("| %d (%dx%d) | %d -> %d | %.2f | %s |" % (lvl, H, W, ci, co, gf, " | ".join("%.1f" % v for v in t)))
            sums = [s + v for s, v in zip(sums, t)]
        tot[lvl] = sums
        print("| %d | five layers | | %s |" % (lvl, " | ".join("%.1f" % v for v in sums)))
    for lvl, s in tot.items():
        print("level %d: MIOpen %.0f us -> planeconv %.0f us (forward + data gradient + weight gradient, five layers)"
              % (lvl, s[0] + s[2] + s[4], s[1] + s[3] + s[5]))


if __name__ == "__main__":
    main()
