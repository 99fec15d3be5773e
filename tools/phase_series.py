# GPU time of forward / backward / optimiser of steady steps against the steps that follow a synchronisation
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unsupervised_depth_opticalflow_egomotion_amd  # noqa
import torch
import bench
from unsupervised_depth_opticalflow_egomotion_amd.train_step import total_loss
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
wl = bench.TrainStepWorkload(args, dev, seed=1234, world=1)
model, opt, cfg, inputs = wl.model, wl.opt, wl.cfg, wl.inputs
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for i in range(40):
    e = [E() for _ in range(4)]
    h0 = time.perf_counter()
    e[0].record()
    opt.zero_grad(set_to_none=True)
    lp, mp = model(inputs)
    loss = total_loss(lp, cfg)
    e[1].record(); h1 = time.perf_counter()
    loss.backward()
    e[2].record(); h2 = time.perf_counter()
    opt.step()
    e[3].record(); h3 = time.perf_counter()
    rows.append((e, (h1 - h0, h2 - h1, h3 - h2)))
    if i in (9, 19, 29):
        torch.cuda.synchronize()
torch.cuda.synchronize()
for i, (e, h) in enumerate(rows):
    g = [e[k].elapsed_time(e[k + 1]) for k in range(3)]
    print("step %2d%s  gpu fwd %.2f bwd %.2f opt %.2f = %.2f   host fwd %.2f bwd %.2f opt %.2f" % (
        i + 1, " (after sync)" if i in (10, 20, 30) else "             ", g[0], g[1], g[2], sum(g), 1e3 * h[0], 1e3 * h[1], 1e3 * h[2]))
