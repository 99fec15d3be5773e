#!/usr/bin/env python3
"""One-off sweep of the CPU baseline's thread count (bench.py cpu_baseline leg: same networks on the host + the oracle loss
stack + Adam, B=4 256x832): s/step at 8 / 16 / 32 / 64 torch threads -> the default of bench.py.
  python tools/cpu_thread_sweep.py > gpurun_out/r03_cpu_thread_sweep.txt"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = types.SimpleNamespace(batch=4, height=256, width=832, scales=3, mode="geom")
wl = bench.TrainStepWorkload.__new__(bench.TrainStepWorkload)
from unsupervised_depth_opticalflow_egomotion_amd import synthetic
from unsupervised_depth_opticalflow_egomotion_amd.train_step import make_cfg
wl.args, wl.mode = args, "geom"
wl.cfg = make_cfg(num_scales=3, img_hw=(256, 832), mode="geom")
wl.np_inputs = synthetic.make_triplet_batch(4, 256, 832, 3, seed=1234)
print("cpu: %s, %d logical cores" % (bench.cpu_model(), os.cpu_count()))
for threads in (8, 16, 32, 64):
    torch.set_num_threads(threads)
    run = wl.cpu_step_fn(threads)
    run()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        run()
    dt = (time.perf_counter() - t0) / n
    print("threads %3d: %.2f s/step = %.2f frame-pairs/s" % (threads, dt, 8 / dt), flush=True)
