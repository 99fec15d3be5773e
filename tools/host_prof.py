"""Where the HOST spends its time enqueueing a train step (cProfile over 10 steps; default B = 1, where the host is the bound).
    python tools/host_prof.py [B]      (on the GPU box)"""
import cProfile, pstats, sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = types.SimpleNamespace(mode="geom", scales=3, height=256, width=832, batch=B, force_ddp=False, graph=False)
dev = torch.device("cuda:0")
wl = bench.TrainStepWorkload(args, dev, seed=1234)
for _ in range(6): wl.step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): wl.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue ms/step %.2f, wall %.2f" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10): wl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
