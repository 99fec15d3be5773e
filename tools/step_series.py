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
    if i in (4, 29, 59, 89):      # 4: the bench's barrier after the warm-up; 29: a bare synchronisation; 59: + 50 ms of idle GPU;
        torch.cuda.synchronize()  # 89: the synchronisation followed at once by a 2 ms filler kernel sequence on the same stream
        if i == 59:
            time.sleep(0.05)
        if i == 89:
            x = torch.empty(64 << 20, device=dev)
            for _ in range(40):
                x.add_(1.0)
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(121)]
print("gpu ms per step:", " ".join("%.2f" % t for t in ts))
print("steps after the synchronisations (6, 31, 61, 91):", ts[5], ts[30], ts[60], ts[90], " median of the rest", sorted(ts[6:])[len(ts[6:]) // 2])
print("host ms per step:", " ".join("%.2f" % (1e3 * h) for h in host))
print("mem reserved MB", torch.cuda.memory_reserved() / 2**20, "num alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"))
print("segments", torch.cuda.memory_stats().get("segment.all.allocated"))
