"""Where the step's time goes per stream, measured WITHOUT a profiler (HIP events on the three streams): forward branch
spans, the loss stack, the backward spans per stream, the optimiser."""
import os, sys, types
sys.path.insert(0, os.getcwd())
import torch
import bench
from unsupervised_depth_opticalflow_egomotion_amd import models
from unsupervised_depth_opticalflow_egomotion_amd.train_step import total_loss

args = types.SimpleNamespace(scales=3, height=256, width=832, batch=4, mode="geom")
dev = torch.device("cuda:0")
wl = bench.TrainStepWorkload(args, dev, 1234)
model, opt, cfg, inputs = wl.model, wl.opt, wl.cfg, wl.inputs
main = torch.cuda.current_stream(dev)
s_flow, s_pose = models._side_streams(dev)
E = lambda: torch.cuda.Event(enable_timing=True)
rec = []

orig_flow, orig_depth = models._flow_branches, models._depth_frames
marks = {}
def flow_wrapped(*a, **k):
    marks["flow_start"] = E(); marks["flow_start"].record(torch.cuda.current_stream())
    r = orig_flow(*a, **k)
    marks["flow_end"] = E(); marks["flow_end"].record(torch.cuda.current_stream())
    return r
def depth_wrapped(*a, **k):
    marks["depth_start"] = E(); marks["depth_start"].record(torch.cuda.current_stream())
    r = orig_depth(*a, **k)
    marks["depth_end"] = E(); marks["depth_end"].record(torch.cuda.current_stream())
    return r
models._flow_branches, models._depth_frames = flow_wrapped, depth_wrapped

def step():
    m = {}
    m["t0"] = E(); m["t0"].record(main)
    opt.zero_grad(set_to_none=True)
    lp, _ = model(inputs)
    m["pose_end"] = E(); m["pose_end"].record(s_pose)
    m["fwd_done"] = E(); m["fwd_done"].record(main)          # after the join + loss stack forward
    loss = total_loss(lp, cfg)
    loss.backward()
    m["bwd_main"] = E(); m["bwd_main"].record(main)
    m["bwd_flow"] = E(); m["bwd_flow"].record(s_flow)
    m["bwd_pose"] = E(); m["bwd_pose"].record(s_pose)
    opt.step()
    m["end"] = E(); m["end"].record(main)
    m.update(marks)
    return m

for _ in range(6): step()
torch.cuda.synchronize()
acc = {}
N = 20
ms = [step() for _ in range(N)]          # no synchronisation between steps: the host runs ahead, as in bench.py
torch.cuda.synchronize()
print("steady state: %.2f ms per step" % (ms[2]["t0"].elapsed_time(ms[-1]["t0"]) / (N - 3)))
for m in ms[2:-1]:
    for k, e in m.items():
        if k != "t0":
            acc[k] = acc.get(k, 0.0) + m["t0"].elapsed_time(e)
N = N - 3
for k in sorted(acc, key=lambda k: acc[k]):
    print("%-12s %7.2f ms after the step's start" % (k, acc[k] / N))
