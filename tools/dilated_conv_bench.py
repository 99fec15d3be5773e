"""Dilated 3x3 convolutions of PWC's context network (B=8, 64x208): MIOpen on the dilated problem vs the same arithmetic as
a dense 3x3 convolution on the d*d phase images (space-to-batch).  us per call, HIP events."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
cb = torch.ops.aten.convolution_backward


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def s2b(x, d):
    B, C, H, W = x.shape
    return x.view(B, C, H // d, d, W // d, d).permute(0, 3, 5, 1, 2, 4).reshape(B * d * d, C, H // d, W // d)


def b2s(y, d, B):
    _, C, h, w = y.shape
    return y.view(B, d, d, C, h, w).permute(0, 3, 4, 1, 5, 2).reshape(B, C, h * d, w * d)


B, H, W = 8, 64, 208
for ci, co, d in [(128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16)]:
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = F.conv2d(x, w, None, 1, d, d); gy = torch.randn_like(y)
    xs = s2b(x, d).contiguous(); ys = F.conv2d(xs, w, None, 1, 1, 1); gys = s2b(gy, d).contiguous()
    err = float((b2s(ys, d, B) - y).abs().max())
    t = {}
    t["fwd"] = timeit(lambda: F.conv2d(x, w, None, 1, d, d))
    t["dgrad"] = timeit(lambda: cb(gy, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [True, False, False]))
    t["wrw"] = timeit(lambda: cb(gy, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [False, True, False]))
    t["fwd_s"] = timeit(lambda: F.conv2d(xs, w, None, 1, 1, 1))
    t["dgrad_s"] = timeit(lambda: cb(gys, xs, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    t["wrw_s"] = timeit(lambda: cb(gys, xs, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    t["s2b_x"] = timeit(lambda: s2b(x, d).contiguous())
    t["b2s_y"] = timeit(lambda: b2s(ys, d, B).contiguous())
    print("%3d->%3d dil %2d | dilated fwd %6.1f dgrad %6.1f wrw %6.1f | phase images fwd %6.1f dgrad %6.1f wrw %6.1f | copies in %5.1f out %5.1f | max diff %.1e" % (
        ci, co, d, t["fwd"], t["dgrad"], t["wrw"], t["fwd_s"], t["dgrad_s"], t["wrw_s"], t["s2b_x"], t["b2s_y"], err))
