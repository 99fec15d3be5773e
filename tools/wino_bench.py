#!/usr/bin/env python3
"""The fused Winograd-on-MFMA convolution (csrc/ops_wino.hip) against MIOpen on the networks' large 3x3 stride-1 layers:
HIP-event time per call of the forward pass and of the data gradient (the same kernel on the transposed filter).

    python tools/wino_bench.py [--iters 30]
"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import ops          # noqa: E402

SHAPES = [(12, 64, 64, 64, 208), (12, 128, 128, 32, 104), (12, 256, 256, 16, 52), (12, 512, 512, 8, 26), (8, 128, 128, 64, 208),
          (8, 115, 128, 64, 208), (8, 256, 96, 64, 208), (8, 224, 64, 64, 208), (8, 160, 32, 64, 208), (8, 128, 128, 32, 104),
          (12, 32, 32, 64, 208), (8, 179, 128, 16, 52)]


def ev(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


VALID = [(12, 96, 32, 130, 418), (12, 128, 64, 66, 210), (12, 256, 128, 34, 106), (12, 512, 256, 18, 54), (12, 64, 32, 130, 418)]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30); a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("| B x Ci -> Co @ H x W | GFLOP (direct) | MIOpen fwd us (TF/s) | wino fwd us (TF/s) | MIOpen dgrad us | wino dgrad us | max rel err |")
    print("|---|---|---|---|---|---|---|")
    for (B, ci, co, H, W) in SHAPES:
        x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; gy = torch.randn(B, co, H, W, device=dev)
        fl = 2.0 * B * co * ci * 9 * H * W
        t_m = ev(lambda: F.conv2d(x, w, None, 1, 1), a.iters)
        t_w = ev(lambda: ops.wino_conv3x3(x, w, 1), a.iters)
        t_md = ev(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]), a.iters)
        t_wd = ev(lambda: ops.wino_conv3x3(gy, w, 1, transposed=True), a.iters)
        ref = F.conv2d(x, w, None, 1, 1)
        err = float((ops.wino_conv3x3(x, w, 1) - ref).abs().max() / ref.abs().max())
        print("| %d x %d -> %d @ %dx%d | %.1f | %.1f (%.0f) | %.1f (%.0f) | %.1f | %.1f | %.1e |" % (
            B, ci, co, H, W, fl / 1e9, t_m, fl / t_m / 1e6, t_w, fl / t_w / 1e6, t_md, t_wd, err))


def valid():
    dev = torch.device("cuda:0")
    print()
    print("| valid (pre-padded input) B x Ci -> Co @ Hp x Wp | MIOpen fwd us | wino fwd us | MIOpen dgrad us | wino dgrad (full) us |")
    print("|---|---|---|---|---|")
    for (B, ci, co, H, W) in VALID:
        x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; gy = torch.randn(B, co, H - 2, W - 2, device=dev)
        t_m = ev(lambda: F.conv2d(x, w), 20)
        t_w = ev(lambda: ops.wino_conv3x3(x, w, 0), 20)
        t_md = ev(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]), 20)
        t_wd = ev(lambda: ops.wino_conv3x3(gy, w, 2, transposed=True), 20)
        print("| %d x %d -> %d @ %dx%d | %.1f | %.1f | %.1f | %.1f |" % (B, ci, co, H, W, t_m, t_w, t_md, t_wd))


def wgrad():
    dev = torch.device("cuda:0")
    print()
    print("| weight gradient B x Ci -> Co @ H x W (P) | MIOpen us (incl. its transposes / fill) | wino wgrad us | max rel err |")
    print("|---|---|---|---|")
    for (B, ci, co, H, W, P) in [(12, 64, 64, 64, 208, 1), (12, 128, 128, 32, 104, 1), (12, 256, 256, 16, 52, 1), (8, 128, 128, 64, 208, 1), (8, 115, 128, 64, 208, 1),
                                 (8, 256, 96, 64, 208, 1), (8, 224, 64, 64, 208, 1), (8, 160, 32, 64, 208, 1), (12, 96, 32, 130, 418, 0), (12, 128, 64, 66, 210, 0),
                                 (12, 256, 128, 34, 106, 0), (12, 32, 32, 64, 208, 1)]:
        x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev); gy = torch.randn(B, co, H + 2 * P - 2, W + 2 * P - 2, device=dev)
        t_m = ev(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [P, P], [1, 1], False, [0, 0], 1, [False, True, False]), 20)
        t_w = ev(lambda: ops.wino_wgrad3x3(x, gy, P), 20)
        ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [P, P], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        err = float((ops.wino_wgrad3x3(x, gy, P) - ref).abs().max() / ref.abs().max())
        print("| %d x %d -> %d @ %dx%d (%d) | %.1f | %.1f | %.1e |" % (B, ci, co, H, W, P, t_m, t_w, err))


if __name__ == "__main__":
    main()
    valid()
    wgrad()
