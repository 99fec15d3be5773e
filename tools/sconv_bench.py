#!/usr/bin/env python3
"""The strided weight gradient of csrc/ops_sconv.hip against MIOpen (aten, its layout transposes and fills included) on the networks'
stride-2 layers: parity against float64 aten first (odd sizes, ragged channel counts), then HIP-event time per call.
(profiles/r05_sconv_bench_all.md was made by this tool while the forward and data-gradient kernels of the family existed.)

    python tools/sconv_bench.py [--iters 20] [--check-only] [--blocks 512,768]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import ops          # noqa: E402
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib  # noqa: E402

# (B, Ci, Co, H, W, K, stride, P): the step's strided layers (12 frames through DepthNet / FeaturePyramid, 4 triplets through PoseCNN)
ONE_BY_ONE = [(12, 64, 128, 64, 208, 1, 2, 0), (12, 128, 256, 32, 104, 1, 2, 0), (12, 256, 512, 16, 52, 1, 2, 0)]
LAYERS = [(12, 3, 64, 256, 832, 7, 2, 3), (12, 64, 128, 64, 208, 3, 2, 1), (12, 128, 256, 32, 104, 3, 2, 1), (12, 256, 512, 16, 52, 3, 2, 1),
          (12, 3, 16, 256, 832, 3, 2, 1), (12, 16, 32, 128, 416, 3, 2, 1), (12, 32, 64, 64, 208, 3, 2, 1), (12, 64, 96, 32, 104, 3, 2, 1),
          (12, 96, 128, 16, 52, 3, 2, 1), (12, 128, 196, 8, 26, 3, 2, 1),
          (4, 9, 16, 256, 832, 7, 2, 3), (4, 16, 32, 128, 416, 5, 2, 2), (4, 32, 64, 64, 208, 3, 2, 1), (4, 64, 128, 32, 104, 3, 2, 1),
          (4, 128, 256, 16, 52, 3, 2, 1), (4, 256, 256, 8, 26, 3, 2, 1), (4, 256, 256, 4, 13, 3, 2, 1)] + ONE_BY_ONE

CHECKS = [(2, 16, 16, 8, 12, 3, 2, 1), (1, 3, 64, 20, 36, 7, 2, 3), (2, 9, 16, 17, 23, 7, 2, 3), (1, 16, 32, 19, 27, 5, 2, 2), (3, 40, 70, 10, 13, 3, 2, 1),
          (2, 3, 16, 9, 11, 3, 2, 1), (1, 64, 128, 64, 208, 3, 2, 1), (2, 33, 17, 7, 9, 3, 2, 1), (1, 256, 256, 4, 13, 3, 2, 1), (2, 96, 128, 16, 52, 3, 2, 1),
          (1, 17, 20, 5, 130, 3, 2, 1), (1, 3, 5, 6, 6, 3, 2, 1), (2, 16, 16, 4, 4, 3, 2, 0), (1, 20, 40, 9, 9, 3, 1, 1),
          (2, 64, 128, 9, 13, 1, 2, 0), (1, 24, 40, 8, 12, 1, 2, 0), (2, 70, 33, 13, 7, 3, 2, 1), (1, 16, 32, 21, 27, 5, 2, 2)]


def ev(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def make(shape, dev):
    B, ci, co, H, W, K, S, P = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, ci, H, W, device=dev)
    w = torch.randn(co, ci, K, K, device=dev)
    ho, wo = (H + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1
    gy = torch.randn(B, co, ho, wo, device=dev)
    return x, w, gy


def aten_bwd(x, w, gy, S, P, mask):
    return torch.ops.aten.convolution_backward(gy, x, w, None, [S, S], [P, P], [1, 1], False, [0, 0], 1, mask)


OURS = {
    "wgrad": lambda x, w, gy, K, S, P: ops.sconv_wgrad(x, gy, K, S, P),
}
ATEN = {
    "wgrad": lambda x, w, gy, K, S, P: aten_bwd(x, w, gy, S, P, [False, True, False])[1],
    "fwd": lambda x, w, gy, K, S, P: torch.nn.functional.conv2d(x, w, None, S, P),
    "dgrad": lambda x, w, gy, K, S, P: aten_bwd(x, w, gy, S, P, [True, False, False])[0],
}
SUPPORTED = {
    "wgrad": lambda shape: ops.sconv_wgrad_supported(shape[:1] + shape[1:2] + shape[3:5], shape[2], shape[5], shape[6], shape[7]),
}


def check(passes, dev):
    bad = 0
    for shape in CHECKS:
        K, S, P = shape[5:]
        x, w, gy = make(shape, dev)
        row = []
        for ps in passes:
            if not SUPPORTED[ps](shape):
                row.append("%s: unsupported" % ps)
                continue
            ref = ATEN[ps](x.double(), w.double(), gy.double(), K, S, P)
            scale = float(ref.abs().max())
            g = OURS[ps](x, w, gy, K, S, P)
            err = float((g.double() - ref).abs().max()) / scale
            rep = bool(torch.equal(g, OURS[ps](x, w, gy, K, S, P)))
            m = float((ATEN[ps](x, w, gy, K, S, P).double() - ref).abs().max()) / scale
            row.append("%s: %.1e (aten-fp32 %.1e)%s" % (ps, err, m, "" if rep else " NOT-REPRODUCIBLE"))
            bad += (not err <= 3e-5) or not rep
        print("check %-36s %s" % (shape, " | ".join(row)), flush=True)
    print("CHECK", "FAILED (%d)" % bad if bad else "ok", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--pass", dest="passes", default="wgrad")
    ap.add_argument("--blocks", default="")          # grid-size targets to sweep, e.g. "256,512,768"
    ap.add_argument("--rows", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    passes = [p for p in a.passes.split(",") if p in OURS]
    bad = check(passes, dev)
    if a.check_only:
        sys.exit(1 if bad else 0)
    lib = get_lib()
    blocks = [int(v) for v in a.blocks.split(",") if v] or [0]
    for ps in passes:
        print()
        print("%s | B x Ci -> Co @ HxW (k, stride, P) | GFLOP | MIOpen us (TF/s) | own us (TF/s) | err |" % ps)
        print("|---|---|---|---|---|")
        tot_m = tot_o = 0.0
        for shape in LAYERS:
            B, ci, co, H, W, K, S, P = shape
            x, w, gy = make(shape, dev)
            fl = 2.0 * B * co * ci * K * K * gy.shape[2] * gy.shape[3]
            t_m = ev(lambda: ATEN[ps](x, w, gy, K, S, P), a.iters)
            if not SUPPORTED[ps](shape):
                print("| %d x %d -> %d @ %dx%d (%d, %d, %d) | %.2f | %.1f (%.0f) | - | |" % (B, ci, co, H, W, K, S, P, fl / 1e9, t_m, fl / t_m / 1e6))
                tot_m += t_m; tot_o += t_m
                continue
            t_o, cell = 1e9, []
            for b in blocks:
                lib.dfe_sconv_tune(b, a.rows)
                t = ev(lambda: OURS[ps](x, w, gy, K, S, P), a.iters)
                cell.append("%.1f" % t)
                t_o = min(t_o, t)
            ref = ATEN[ps](x.double(), w.double(), gy.double(), K, S, P)
            err = float((OURS[ps](x, w, gy, K, S, P).double() - ref).abs().max() / ref.abs().max())
            tot_m += t_m; tot_o += t_o
            print("| %d x %d -> %d @ %dx%d (%d, %d, %d) | %.2f | %.1f (%.0f) | %s (%.0f) | %.1e |" % (
                B, ci, co, H, W, K, S, P, fl / 1e9, t_m, fl / t_m / 1e6, " / ".join(cell), fl / t_o / 1e6, err), flush=True)
        print("%s, sum over the listed layers: MIOpen %.0f us, own %.0f us" % (ps, tot_m, tot_o))


if __name__ == "__main__":
    main()
