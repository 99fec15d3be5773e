#!/usr/bin/env python3
"""PWC-side hot-path kernels (a20 corr_naive, a2 warp_flow on feature maps, the fused level input) at the five PWC
levels of a 256x832 frame, B = 8 (target, source) pairs -- the shapes of BASELINE configs[2]: HIP-event time per
launch against ALGORITHMIC bytes (correlation forward (2C + 81) H W 4 B, each gradient
(C + 81 + C) H W 4; warp_flow forward (2C + 2) H W 4, backward (3C + 2 + 2 + C) H W 4 with the feature gradient), as a
fraction of the 8 TB/s HBM peak.  Writes the markdown table of profiles/r04_pwc_roofline_table.md to stdout.

    python tools/corr_bench.py [--check] [--sweep]

--check compares every shape against a float64 torch composition on the device (max abs error printed).
--sweep tries the tuning overrides DFE_CORR_FWD / DFE_CORR_BWD (the launcher reads them per call)."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from unsupervised_depth_opticalflow_egomotion_amd import ops
from unsupervised_depth_opticalflow_egomotion_amd._lib import check, get_lib, stream_ptr

PEAK = 8.0e12
LEVELS = [(2, 32, 64, 208), (3, 64, 32, 104), (4, 96, 16, 52), (5, 128, 8, 26), (6, 196, 4, 13)]
dev = torch.device("cuda:0")
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def corr_ref64(f1, f2):
    B, C, H, W = f1.shape
    p = F.pad(f2.double(), (4, 4, 4, 4))
    a = f1.double()
    return torch.stack([(a * p[:, :, i:i + H, j:j + W]).mean(1) for i in range(9) for j in range(9)], 1)


def run_level(B, C, H, W, do_check, only_corr=False, iters=50):
    lib = get_lib()
    g = torch.Generator(device="cpu").manual_seed(C)
    f1 = torch.randn(B, C, H, W, generator=g).to(dev)
    f2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 81, H, W, generator=g).to(dev)
    # a smooth flow field (what a PWC net produces): low-resolution noise up-sampled, ~2 px amplitude
    flow = (F.interpolate(torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g), size=(H, W), mode="bilinear", align_corners=False) * 2.0).to(dev)
    out = torch.empty(B, 81, H, W, device=dev)
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    s = stream_ptr()
    fwd = lambda: check(lib.dfe_corr_fwd(ptr(f1), ptr(f2), ptr(out), B, C, H, W, 4, s), "fwd")
    bw1 = lambda: check(lib.dfe_corr_bwd(ptr(f1), ptr(f2), ptr(go), ptr(g1), None, B, C, H, W, 4, s), "bwd1")
    bw2 = lambda: check(lib.dfe_corr_bwd(ptr(f1), ptr(f2), ptr(go), None, ptr(g2), B, C, H, W, 4, s), "bwd2")
    bw12 = lambda: check(lib.dfe_corr_bwd(ptr(f1), ptr(f2), ptr(go), ptr(g1), ptr(g2), B, C, H, W, 4, s), "bwd12")
    res = {"corr_fwd": (timeit(fwd, iters), (2 * C + 81) * H * W * 4 * B), "corr_bwd_g1": (timeit(bw1, iters), (2 * C + 81) * H * W * 4 * B),
           "corr_bwd_g2": (timeit(bw2, iters), (2 * C + 81) * H * W * 4 * B),
           "corr_bwd (g1 + g2, one launch)": (timeit(bw12, iters), (4 * C + 81) * H * W * 4 * B)}
    err = None
    if do_check:
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        ref = corr_ref64(a, b)
        (ref * go.double()).sum().backward()
        fwd(); bw1(); bw2()
        torch.cuda.synchronize()
        err = (float((out.double() - ref.detach()).abs().max()), float((g1.double() - a.grad.double()).abs().max()),
               float((g2.double() - b.grad.double()).abs().max()))
    if only_corr:
        return res, err
    # feature warp (pwc_tf.py:119 et al.) and the fused level input: straight through the C ABI with caller-owned buffers (the
    # autograd wrappers add allocations and Python time that the training step hides behind its GPU work)
    HW, n = H * W, B * C * H * W
    warped, gflow, gx2 = torch.empty_like(f2), torch.empty_like(flow), torch.empty_like(f2)
    ws = torch.empty(lib.dfe_scatter_ws_bytes(n) // 4 + 4, device=dev, dtype=torch.float32)
    gy = torch.randn(B, C, H, W, generator=g).to(dev)
    wf = lambda: check(lib.dfe_warp_flow_fwd(ptr(f2), ptr(flow), ptr(warped), B, C, H, W, 0, 0, s), "wf")
    wb = lambda: check(lib.dfe_warp_flow_bwd(ptr(f2), ptr(flow), ptr(gy), ptr(gflow), ptr(gx2), ptr(ws), B, C, H, W, 0, 0, s), "wb")
    res["warp_flow_fwd"] = (timeit(wf, iters), (2 * C + 2) * HW * 4 * B)
    res["warp_flow_bwd (levels >= 16x52: header fill + max + flow gradient + inverse map (count, scan, fill) + gather; below: zero-fill + max + scatter + convert)"] = (timeit(wb, iters), (4 * C + 4) * HW * 4 * B)
    x = torch.empty(B, lib.dfe_pwc_level_channels(C), H, W, device=dev)
    gxl = torch.randn_like(x)
    g_w, g_c1, g_c2, g_fl = torch.empty_like(f1), torch.empty_like(f1), torch.empty_like(f1), torch.empty_like(flow)
    lv = lambda: check(lib.dfe_pwc_level_fwd(ptr(f1), ptr(f2), ptr(flow), ptr(warped), ptr(x), B, C, H, W, 0, s), "lvl")
    lb = lambda: check(lib.dfe_pwc_level_bwd(ptr(f1), ptr(f2), ptr(flow), ptr(warped), ptr(gxl), ptr(g_w), ptr(g_c1), ptr(g_c2), ptr(ws),
                                             ptr(g_fl), B, C, H, W, 0, s), "lvlb")
    res["pwc_level_fwd (warp + corr + c1 / flow planes: 2 launches)"] = (timeit(lv, iters), ((2 * C + 2) + (2 * C + 81) + 2 * (C + 2)) * HW * 4 * B)
    # round 6: the inverse map of the feature warp built in the forward pass (what PwcLevelInputFn runs when dL/dc2 is wanted)
    mp = torch.empty(lib.dfe_pwc_level_map_bytes(B, H, W), device=dev, dtype=torch.uint8)
    lvm = lambda: check(lib.dfe_pwc_level_fwd_map(ptr(f1), ptr(f2), ptr(flow), ptr(warped), ptr(x), ptr(mp), B, C, H, W, 0, s), "lvlm")
    lbm = lambda: check(lib.dfe_pwc_level_bwd_map(ptr(f1), ptr(f2), ptr(flow), ptr(warped), ptr(gxl), ptr(g_w), ptr(g_c1), ptr(g_c2), ptr(mp),
                                                  ptr(g_fl), B, C, H, W, 0, s), "lvlbm")
    res["pwc_level_fwd_map (planes > 1024 px: zero-fill + warp with tap count + scan + fill + corr; smaller: warp + one-launch map + corr)"] = (timeit(lvm, iters), ((2 * C + 2) + (2 * C + 81) + 2 * (C + 2)) * HW * 4 * B)
    lvm()
    res["pwc_level_bwd_map (corr both gradients + flow gradient + gather: 3 launches at every level)"] = (timeit(lbm, iters), (2 * (2 * C + 81) + (C + 2) + (4 * C + 4)) * HW * 4 * B)
    res["pwc_level_bwd (levels >= 16x52: header fill + corr both gradients + flow gradient + inverse map (3 launches) + gather; below: zero-fill + corr + scatter + convert)"] = (timeit(lb, iters), (2 * (2 * C + 81) + (C + 2) + (4 * C + 4)) * HW * 4 * B)
    return res, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only-corr", action="store_true")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--lib", default="", help="another build of libdfe_hip.so (tools/corr_ablate.sh) instead of the in-tree one")
    a = ap.parse_args()
    if a.lib:
        from unsupervised_depth_opticalflow_egomotion_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    B = a.batch
    if a.sweep:
        lib = get_lib()
        s = stream_ptr()
        for lvl, C, H, W in LEVELS:
            f1 = torch.randn(B, C, H, W, device=dev); f2 = torch.randn(B, C, H, W, device=dev)
            go = torch.randn(B, 81, H, W, device=dev); out = torch.empty(B, 81, H, W, device=dev)
            g1 = torch.empty_like(f1); g2 = torch.empty_like(f1)
            best = []
            for dyg in (9, 5, 3):
                os.environ["DFE_CORR_DYG"] = str(dyg)
                for th in (1, 2, 4, 8):
                    for ks in (0, 1, 2, 4):
                        for m in (0, 8, 16, 32):
                            os.environ["DFE_CORR_FWD"] = "%d,%d,%d" % (th, ks, m)
                            try:
                                t = timeit(lambda: check(lib.dfe_corr_fwd(ptr(f1), ptr(f2), ptr(out), B, C, H, W, 4, s), "fwd"), n=20, warm=2)
                            except Exception:
                                continue
                            best.append((t, th, ks, m, dyg))
            os.environ.pop("DFE_CORR_DYG", None)
            os.environ.pop("DFE_CORR_FWD", None)
            best.sort()
            print("level %d fwd  best (us, TH, KS, CC, DYG):" % lvl, ["%.1f %d %d %d %d" % b for b in best[:8]], flush=True)
            best = []
            for th in (1, 2, 4, 8):
                for ncg in (1, 2, 3, 4, 6):
                    for isp in (1, 2, 3, 5, 9):
                        os.environ["DFE_CORR_BWD"] = "%d,%d,%d" % (th, ncg, isp)
                        try:
                            t0 = timeit(lambda: check(lib.dfe_corr_bwd(ptr(f1), ptr(f2), ptr(go), ptr(g1), ptr(g2), B, C, H, W, 4, s), "b"), n=20, warm=2)
                            t1 = 0.0
                        except Exception:
                            continue
                        best.append((t0 + t1, t0, t1, th, ncg, isp))
            os.environ.pop("DFE_CORR_BWD", None)
            best.sort()
            print("level %d bwd  best (g1+g2 us, g1, g2, TH, NCG, IS):" % lvl, ["%.1f %.1f %.1f %d %d %d" % b for b in best[:6]], flush=True)
        return
    print("# PWC-side hot-path kernels against ALGORITHMIC bytes (tools/corr_bench.py%s, MI355X)\n" % (" --check" if a.check else ""))
    print("HIP-event time per launch (50 launches back to back, so a launch's floor of ~5 us is in every figure) against the bytes the")
    print("operator cannot avoid: correlation forward (2C + 81) H W 4 B, both gradients (4C + 81) H W 4; feature warp forward (2C + 2) H W 4,")
    print("backward (4C + 4) H W 4; the fused level input = warp + correlation + the c1 / flow planes.  The five PWC levels of a 256x832")
    print("frame, B = %d (target, source) pairs = BASELINE configs[2]; flows are smooth (low-resolution noise up-sampled, ~2 px).\n" % B)
    print("| level (C, H x W), B = %d | kernel / operator | algorithmic MB | us | alg GB/s | frac of 8 TB/s |" % B)
    print("|---|---|---|---|---|---|")
    tot = {}
    for lvl, C, H, W in LEVELS:
        res, err = run_level(B, C, H, W, a.check, a.only_corr, a.iters)
        for k, (us, nbytes) in res.items():
            print("| %d (%d, %dx%d) | %s | %.2f | %.1f | %.0f | %.3f |" % (lvl, C, H, W, k, nbytes / 1e6, us, nbytes / us / 1e3, nbytes / (us * 1e-6) / PEAK))
            tot[k] = tot.get(k, 0.0) + us
        if err:
            print("<!-- level %d max abs error vs float64: out %.2e g1 %.2e g2 %.2e -->" % ((lvl,) + err))
    print("\nsum over the five levels (us): " + ", ".join("%s %.1f" % kv for kv in tot.items()))
    if a.only_corr:
        return
    print("correlation per training step (one PWC pass over the 2B pairs): fwd %.1f + bwd %.1f us"
          % (tot["corr_fwd"], tot["corr_bwd (g1 + g2, one launch)"]))
    g = lambda pre: sum(v for k, v in tot.items() if k.startswith(pre + " "))      # noqa: E731
    print("fused level input over the five levels, map built in the forward pass (what the training step runs): fwd %.1f + bwd %.1f = %.1f us"
          % (g("pwc_level_fwd_map"), g("pwc_level_bwd_map"), g("pwc_level_fwd_map") + g("pwc_level_bwd_map")))
    print("the same with the map built per backward call (round 5; DFE_PWC_LEVEL_MAP=0):                           fwd %.1f + bwd %.1f = %.1f us"
          % (g("pwc_level_fwd"), g("pwc_level_bwd"), g("pwc_level_fwd") + g("pwc_level_bwd")))


if __name__ == "__main__":
    main()
