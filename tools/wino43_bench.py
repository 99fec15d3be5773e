import os, sys, ctypes, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from unsupervised_depth_opticalflow_egomotion_amd import ops
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib, ptr, stream_ptr, check
lib = get_lib()
SHAPES = [(2, 64, 64, 16, 24), (12, 64, 64, 64, 208), (12, 128, 128, 32, 104), (8, 128, 128, 64, 208), (8, 256, 96, 64, 208), (8, 224, 64, 64, 208),
          (8, 160, 32, 64, 208), (12, 32, 32, 64, 208), (12, 32, 16, 128, 416), (12, 256, 256, 16, 52)]
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
dev = torch.device("cuda:0")
def w43(x, w, bias=None, slope=1.0, transposed=False):
    B, Ci, H, W = x.shape
    Co = w.shape[1] if transposed else w.shape[0]
    y = torch.empty(B, Co, H, W, device=x.device)
    wb = torch.empty(lib.dfe_wino43_weight_floats(Ci, Co), device=x.device)
    check(lib.dfe_wino43_conv3x3(ptr(x), ptr(w), ptr(bias) if bias is not None else None, float(slope), ptr(y), y.stride(0), ptr(wb), B, Ci, Co, H, W, int(transposed), stream_ptr()), "w43")
    return y
for (B, ci, co, H, W) in SHAPES:
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; gy = torch.randn(B, co, H, W, device=dev)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    y = w43(x, w)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    refd = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    errd = float((w43(gy, w, transposed=True).double() - refd).abs().max() / refd.abs().max())
    e23 = float((ops.wino_conv3x3(x, w, 1).double() - ref).abs().max() / ref.abs().max())
    t43 = ev(lambda: w43(x, w)); t23 = ev(lambda: ops.wino_conv3x3(x, w, 1))
    print("%2dx%3d->%3d@%3dx%3d  F(4,3) %.1f us  F(2,3) %.1f us  ratio %.2f | err fwd %.1e dgrad %.1e (F(2,3) %.1e)" % (B, ci, co, H, W, t43, t23, t23 / t43, err, errd, e23))
