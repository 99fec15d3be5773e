#!/usr/bin/env python3
"""The Winograd-domain weight gradient (csrc/ops_wino_wgrad.hip) against MIOpen's weight gradients on the networks' 3x3
stride-1 layers (profiles/r03_conv_census_tuned.txt): parity against float64 aten first, then HIP-event time per call for each
wave tile (dfe_wino_wgrad_tune).

    python tools/wgrad_bench.py [--iters 20] [--check-only] [--tiles 22,21,12]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import ops          # noqa: E402
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib  # noqa: E402

# (B, Ci, Co, H, W, P, dilation): the step's layers (B = 12 frames / 8 pairs)
LAYERS = [(12, 64, 64, 64, 208, 1, 1), (12, 128, 128, 32, 104, 1, 1), (12, 256, 256, 16, 52, 1, 1), (12, 512, 512, 8, 26, 1, 1),
          (8, 128, 128, 64, 208, 1, 1), (8, 115, 128, 64, 208, 1, 1), (8, 256, 96, 64, 208, 1, 1), (8, 224, 64, 64, 208, 1, 1),
          (8, 160, 32, 64, 208, 1, 1), (8, 34, 128, 64, 208, 1, 1), (8, 147, 128, 32, 104, 1, 1), (8, 256, 96, 32, 104, 1, 1),
          (8, 224, 64, 32, 104, 1, 1), (8, 128, 128, 32, 104, 1, 1), (12, 64, 64, 32, 104, 1, 1), (8, 179, 128, 16, 52, 1, 1),
          (12, 96, 96, 16, 52, 1, 1), (12, 32, 32, 64, 208, 1, 1), (8, 64, 32, 64, 208, 1, 1),
          (12, 128, 64, 66, 210, 0, 1), (12, 256, 128, 34, 106, 0, 1), (12, 512, 256, 18, 54, 0, 1), (12, 128, 64, 34, 106, 0, 1),
          (12, 256, 128, 18, 54, 0, 1), (12, 512, 256, 10, 28, 0, 1), (12, 96, 32, 130, 418, 0, 1), (12, 64, 32, 66, 210, 0, 1),
          (12, 32, 16, 130, 418, 0, 1), (12, 16, 16, 258, 834, 0, 1),
          (8, 128, 128, 64, 208, 2, 2), (8, 128, 128, 64, 208, 4, 4), (8, 128, 96, 64, 208, 8, 8), (8, 96, 64, 64, 208, 16, 16)]

CHECKS = [(2, 64, 64, 16, 24, 1, 1), (3, 40, 70, 10, 12, 1, 1), (2, 33, 17, 8, 6, 0, 1), (1, 3, 5, 4, 4, 1, 1), (2, 96, 32, 34, 50, 0, 1),
          (2, 128, 128, 32, 104, 1, 1), (1, 16, 16, 64, 208, 1, 1), (1, 70, 35, 7, 9, 1, 1), (2, 5, 3, 3, 3, 1, 1), (1, 65, 65, 5, 31, 0, 1),
          (2, 130, 40, 9, 13, 1, 1), (1, 32, 32, 16, 32, 2, 2), (1, 40, 36, 32, 64, 4, 4), (2, 33, 17, 8, 12, 2, 2), (1, 24, 16, 32, 48, 8, 8),
          (1, 8, 8, 64, 64, 16, 16), (1, 64, 64, 13, 27, 1, 1), (1, 64, 64, 4, 106, 0, 1)]


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
    B, ci, co, H, W, P, d = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, ci, H, W, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev)
    ho, wo = (H, W) if d > 1 else (H + 2 * P - 2, W + 2 * P - 2)
    gy = torch.randn(B, co, ho, wo, device=dev)
    return x, w, gy


def aten_wgrad(x, w, gy, P, d):
    return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [P, P], [d, d], False, [0, 0], 1, [False, True, False])[1]


def check(tiles, dev):
    lib = get_lib()
    bad = 0
    for shape in CHECKS:
        B, ci, co, H, W, P, d = shape
        x, w, gy = make(shape, dev)
        ref = aten_wgrad(x.double(), w.double(), gy.double(), P, d)
        scale = float(ref.abs().max())
        row = []
        for t in tiles:
            assert lib.dfe_wino_wgrad_tune(t, 0, 0, 0) == 0
            g = ops.wino_wgrad3x3(x, gy, P if d == 1 else 1, d)
            err = float((g.double() - ref).abs().max()) / scale
            rep = bool(torch.equal(g, ops.wino_wgrad3x3(x, gy, P if d == 1 else 1, d)))
            row.append("%d: %.1e%s" % (t, err, "" if rep else " NOT-REPRODUCIBLE"))
            bad += (err > 3e-5) or not rep
        m = float((aten_wgrad(x, w, gy, P, d).double() - ref).abs().max()) / scale
        print("check %-34s aten-fp32 %.1e | %s" % (shape, m, " | ".join(row)), flush=True)
    lib.dfe_wino_wgrad_tune(0, 0, 0, 0)
    print("CHECK", "FAILED (%d)" % bad if bad else "ok", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--tiles", default="21,12,11")
    ap.add_argument("--blocks", default="")          # e.g. "256:512,512:768": block targets (one- / two-blocks-per-CU kernels) to sweep
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tiles = [int(t) for t in a.tiles.split(",")]
    bad = check([0] + tiles, dev)
    if a.check_only:
        sys.exit(1 if bad else 0)
    lib = get_lib()
    blocks = [tuple(int(v) for v in b.split(":")) for b in a.blocks.split(",") if b] or [(0, 0)]
    print()
    print("| B x Ci -> Co @ HxW (P, dil) | GFLOP | MIOpen us (TF/s) | " + " | ".join("tile %d us (TF/s)" % t for t in tiles) + " | err |")
    print("|---|---|---|" + "---|" * (len(tiles) + 1))
    tot_m, tot_b = 0.0, 0.0
    for shape in LAYERS:
        B, ci, co, H, W, P, d = shape
        x, w, gy = make(shape, dev)
        fl = 2.0 * B * co * ci * 9 * gy.shape[2] * gy.shape[3]
        t_m = ev(lambda: aten_wgrad(x, w, gy, P, d), a.iters)
        cells, best = [], 1e9
        for t in tiles:
            if (t // 10 == 2 and co <= 16) or (t % 10 == 2 and ci <= 16):
                cells.append("-")
                continue
            tb = 1e9
            for (b1, b2) in blocks:
                lib.dfe_wino_wgrad_tune(t, b1, b2, 0)
                tb = min(tb, ev(lambda: ops.wino_wgrad3x3(x, gy, P if d == 1 else 1, d), a.iters))
            cells.append("%.1f (%.0f)" % (tb, fl / tb / 1e6))
            best = min(best, tb)
        lib.dfe_wino_wgrad_tune(0, 768, 512, 0)
        ref = aten_wgrad(x.double(), w.double(), gy.double(), P, d)
        err = float((ops.wino_wgrad3x3(x, gy, P if d == 1 else 1, d).double() - ref).abs().max() / ref.abs().max())
        tot_m += t_m; tot_b += min(best, t_m)
        print("| %d x %d -> %d @ %dx%d (%d, %d) | %.1f | %.1f (%.0f) | %s | %.1e |" % (
            B, ci, co, H, W, P, d, fl / 1e9, t_m, fl / t_m / 1e6, " | ".join(cells), err), flush=True)
    print("sum over the listed layers: MIOpen %.0f us, best-of %.0f us" % (tot_m, tot_b))


if __name__ == "__main__":
    main()
