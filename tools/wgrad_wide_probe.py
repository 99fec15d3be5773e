"""Round-4 probe: does this build's MFMA weight-gradient kernel (dfe_wgrad3x3_fwd, built for the thin
decoder layers) beat MIOpen's implicit-GEMM weight gradient INCLUDING its three layout transposes on the wide layers?
HIP-event time per call.  Result on MI355X: no (45-58 TFLOP/s against 69-97), see DESIGN.md section 9.

    python tools/wgrad_wide_probe.py
"""
import sys, os, torch, ctypes
sys.path.insert(0, os.getcwd())
from unsupervised_depth_opticalflow_egomotion_amd import ops
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib, ptr, stream_ptr, check
import torch.nn.functional as F
dev = torch.device("cuda:0"); lib = get_lib()
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) * 1e3 / n
for (B, ci, co, H, W) in [(12, 64, 64, 64, 208), (12, 128, 128, 32, 104), (12, 128, 64, 64, 208), (8, 128, 128, 64, 208), (12, 32, 32, 64, 208), (12,64,64,128,416)]:
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev); gy = torch.randn(B, co, H, W, device=dev)
    fl = 2.0 * B * co * ci * 9 * H * W
    tm = ev(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    gw_ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    n = lib.dfe_wgrad3x3_partials_floats(B, ci, co, H, W)
    if n <= 0:
        print(B, ci, co, H, W, "unsupported"); continue
    part = torch.empty(n, device=dev); gw = torch.empty_like(w)
    def mine():
        p = F.pad(x, (1, 1, 1, 1))
        check(lib.dfe_wgrad3x3_fwd(ptr(p), ptr(gy), ptr(gw), ptr(part), B, ci, co, H, W, stream_ptr()), "wg")
    tp = ev(lambda: F.pad(x, (1, 1, 1, 1)))
    t2 = ev(mine)
    err = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
    print("B%2d %3d->%3d %3dx%3d %6.2f GF | MIOpen wrw %6.1f us %5.1f TF | dfe_wgrad3x3 incl. pad %6.1f us (pad %5.1f) %5.1f TF | rel err %.1e | partials %.1f MB" % (B, ci, co, H, W, fl/1e9, tm, fl/tm/1e6, t2, tp, fl/t2/1e6, err, n*4/1e6))
