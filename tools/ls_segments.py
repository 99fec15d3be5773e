#!/usr/bin/env python3
"""Per-launch durations of the fused loss stack (HIP events between the launches, dfe_geom_loss_*_timed) on synthetic
net outputs: python tools/ls_segments.py [--batch 4 --height 256 --width 832 --scales 3 --iters 200]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import synthetic, loss_stack as LS
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--height", type=int, default=256)
ap.add_argument("--width", type=int, default=832); ap.add_argument("--scales", type=int, default=3)
ap.add_argument("--iters", type=int, default=200); ap.add_argument("--tag", default="")
ap.add_argument("--flow-noise", type=float, default=0.3, help="white noise (px at scale 0) added to the synthetic flows; 0 = smooth rigid flows")
ap.add_argument("--lib", default="", help="another build of libdfe_hip.so (tools/ablate_point_fwd.sh) instead of the in-tree one")
a = ap.parse_args()
if a.lib:
    from unsupervised_depth_opticalflow_egomotion_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda:0")
inp = synthetic.make_loss_stack_inputs(a.batch, a.height, a.width, a.scales, seed=1234, num_flow_scales=max(a.scales, 4) if a.scales > 3 else None, flow_noise=a.flow_noise)
g = lambda x, grad=False: torch.from_numpy(np.ascontiguousarray(x)).to(dev).requires_grad_(grad)
imgs = [g(x) for x in inp.imgs]; disps = [[g(x, True) for x in l] for l in inp.disps]; pose = g(inp.pose, True)
fb = [g(x, True) for x in inp.flows_bwd]; ff = [g(x, True) for x in inp.flows_fwd]; K, Ki = g(inp.K), g(inp.K_inv)
def step():
    lp = LS.geom_loss_stack(imgs[0], imgs[1], imgs[2], disps[0], disps[1], disps[2], pose, fb, ff, K, Ki, num_scales=a.scales)
    sum(v.mean() for v in lp.values()).backward()
for _ in range(20): step()
torch.cuda.synchronize(); LS.timing_begin()
for _ in range(a.iters): step()
torch.cuda.synchronize(); f, b = LS.timing_collect()
names_f = ["prep", "pyramids", "point_fwd", "ssim_fwd", "flow_smooth", "disp_smooth", "reduce"]
names_b = ["ssim_bwd", "point_bwd", "flow_smooth_bwd", "disp_smooth_bwd1", "bwd2", "pose_finalize"]
print(a.tag, "fwd us:", " ".join("%s=%.1f" % (n, 1e3 * v) for n, v in zip(names_f, f.mean(0))), "| sum %.1f" % (1e3 * f.mean(0).sum()))
print(a.tag, "bwd us:", " ".join("%s=%.1f" % (n, 1e3 * v) for n, v in zip(names_b, b.mean(0))), "| sum %.1f" % (1e3 * b.mean(0).sum()))
