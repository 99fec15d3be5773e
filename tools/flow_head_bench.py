import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from unsupervised_depth_opticalflow_egomotion_amd import ops
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
for (B, C, H, W) in [(8, 96, 64, 208), (8, 32, 64, 208), (8, 96, 32, 104), (8, 96, 16, 52), (8, 96, 8, 26), (8, 96, 4, 13)]:
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(2, C, 3, 3, device=dev) * 0.05; b = torch.randn(2, device=dev)
    gy = torch.randn(B, 2, H, W, device=dev)
    tf = timeit(lambda: ops.flow_head_fwd_raw(x, w, b)); tb = timeit(lambda: ops.flow_head_bwd_raw(x, w, gy))
    mf = timeit(lambda: F.conv2d(x, w, b, 1, 1)); mb = timeit(lambda: cb(gy, x, w, [2], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True]))
    print("B%d %3d->2 %3dx%3d | head kernels fwd %6.1f bwd %6.1f us | MIOpen fwd %6.1f bwd %6.1f us" % (B, C, H, W, tf, tb, mf, mb))
