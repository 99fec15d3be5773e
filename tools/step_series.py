# per-step GPU time (events) of the first 60 training steps of a fresh process: where do the slow early steps come from?
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unsupervised_depth_opticalflow_egomotion_amd  # noqa
import torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
wl = bench.TrainStepWorkload(args, dev, seed=1234, world=1)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(122)]
host = []
evs[0].record()
for i in range(121):
    t0 = time.perf_counter()
    wl.step()
    host.append(time.perf_counter() - t0)
    evs[i + 1].record()
    if i in (4,):      # the bench's barrier after the warm-up
        torch.cuda.synchronize()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(121)]
print("gpu ms per step:", " ".join("%.2f" % t for t in ts))
print("host ms per step:", " ".join("%.2f" % (1e3 * h) for h in host))
print("mem reserved MB", torch.cuda.memory_reserved() / 2**20, "num alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"))
print("segments", torch.cuda.memory_stats().get("segment.all.allocated"))
