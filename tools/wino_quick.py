import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from unsupervised_depth_opticalflow_egomotion_amd import ops
SHAPES = [(12, 64, 64, 64, 208), (12, 128, 128, 32, 104), (12, 256, 256, 16, 52), (12, 512, 512, 8, 26), (8, 128, 128, 64, 208),
          (8, 256, 96, 64, 208), (8, 224, 64, 64, 208), (8, 160, 32, 64, 208), (8, 128, 128, 32, 104), (12, 32, 32, 64, 208)]
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else ""
out = []
for (B, ci, co, H, W) in SHAPES:
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; gy = torch.randn(B, co, H, W, device=dev)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    err = float((ops.wino_conv3x3(x, w, 1).double() - ref).abs().max() / ref.abs().max())
    t_w = ev(lambda: ops.wino_conv3x3(x, w, 1)); t_wd = ev(lambda: ops.wino_conv3x3(gy, w, 1, transposed=True))
    out.append("%dx%d->%d@%dx%d %.1f/%.1f e%.0e" % (B, ci, co, H, W, t_w, t_wd, err))
print(tag, " | ".join(out))
