"""Per-layer rates of MIOpen's fp32 convolutions at the shapes the batched nets run.

  python tools/conv_layer_bench.py          # on an MI355X box

Part 1: 3x3 stride-1 layers of the encoder / decoder / PWC: forward, data gradient, weight gradient in TFLOP/s
(nominal 2*B*Co*Ci*9*H*W FLOP).  Part 2: the thin, wide decoder layers on pre-padded inputs, NCHW vs channels_last,
next to the time their HBM traffic alone would take."""
import time

import torch

dev = torch.device("cuda:0")
shapes = [(12, 64, 64, 64, 208), (12, 128, 128, 32, 104), (12, 256, 256, 16, 52), (12, 512, 512, 8, 26),
          (12, 512, 256, 16, 52), (12, 256, 128, 32, 104), (12, 128, 64, 64, 208), (12, 96, 32, 128, 416), (12, 16, 16, 256, 832),
          (8, 115, 128, 32, 104), (8, 128, 128, 32, 104), (8, 256, 96, 32, 104), (8, 83, 128, 64, 208), (8, 128, 128, 64, 208), (12, 32, 32, 64, 208)]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for (B, ci, co, H, W) in shapes:
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev); gy = torch.randn(B, co, H, W, device=dev)
    fl = 2.0 * B * co * ci * 9 * H * W
    tw = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    td = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    tf = t(lambda: torch.nn.functional.conv2d(x, w, None, 1, 1))
    print("B%2d %3d->%3d %3dx%3d  %6.2f GF | wrw %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | fwd %7.1f us %5.1f TF" % (
        B, ci, co, H, W, fl / 1e9, tw * 1e6, fl / tw / 1e12, td * 1e6, fl / td / 1e12, tf * 1e6, fl / tf / 1e12))

print()
thin = [(12, 16, 16, 256, 832), (12, 96, 32, 128, 416), (12, 32, 16, 128, 416), (12, 16, 1, 256, 832), (12, 32, 1, 128, 416), (12, 64, 32, 64, 208)]
for cl in (False, True):
    for (B, ci, co, H, W) in thin:
        x = torch.randn(B, ci, H + 2, W + 2, device=dev); w = torch.randn(co, ci, 3, 3, device=dev); gy = torch.randn(B, co, H, W, device=dev)
        if cl:
            x, w, gy = (a.contiguous(memory_format=torch.channels_last) for a in (x, w, gy))
        cb = lambda m: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, m)
        tw = t(lambda: cb([False, True, False])) * 1e6; td = t(lambda: cb([True, False, False])) * 1e6; tf = t(lambda: torch.nn.functional.conv2d(x, w)) * 1e6
        mb = (B * ci * (H + 2) * (W + 2) + B * co * H * W) * 4 / 1e6
        print("%s B%2d %3d->%3d %3dx%3d  %6.1f MB | fwd %6.1f us | dgrad %6.1f us | wrw %6.1f us | hbm-ideal %5.1f us" % ("NHWC" if cl else "NCHW", B, ci, co, H, W, mb, tf, td, tw, mb / 4.0))
