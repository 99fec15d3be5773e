"""MIOpen convolution rates by layout and dtype (round 4): the same 3x3 layers as NCHW and as channels_last tensors with
PYTORCH_MIOPEN_SUGGEST_NHWC=1, fp32 and bf16 -- what a bf16 / NHWC network path could buy.
    python tools/nhwc_probe.py [1|0]"""
import os, sys, time
os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
cb = torch.ops.aten.convolution_backward
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
for (B, ci, co, H, W) in [(8, 128, 128, 64, 208), (12, 64, 64, 64, 208), (8, 211, 128, 8, 26)]:
    for dt in (torch.bfloat16, torch.float32):
        for cl in (False, True):
            x = torch.randn(B, ci, H, W, device=dev, dtype=dt); w = torch.randn(co, ci, 3, 3, device=dev, dtype=dt); gy = torch.randn(B, co, H, W, device=dev, dtype=dt)
            if cl:
                x, w, gy = (a.contiguous(memory_format=torch.channels_last) for a in (x, w, gy))
            y = F.conv2d(x, w, None, 1, 1)
            tf = t(lambda: F.conv2d(x, w, None, 1, 1))
            tw = t(lambda: cb(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
            td = t(lambda: cb(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
            print("B%d %d->%d %dx%d %s %s | fwd %.1f dgrad %.1f wrw %.1f us | out cl=%s" % (B, ci, co, H, W, str(dt)[6:], "NHWC" if cl else "NCHW", tf, td, tw, y.is_contiguous(memory_format=torch.channels_last)), flush=True)
