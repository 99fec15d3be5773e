#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.  Contract: python bench.py --gpus N --steps K --warmup W
prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for every definition used here.

Workloads (BASELINE.json configs; synthetic KITTI-shaped inputs, seed 1234 + rank):
  train_step  mode=geom, 832x256, B=4/GPU: DepthNet x3 + PoseCNN + PWC x2 (PyTorch-ROCm) + the HIP loss
              stack, forward + backward + Adam  (configs[2]; the configuration the metric is quoted on)
  loss_stack  the loss stack alone on synthetic net outputs, forward + backward (hot path in isolation)
Unit: frame pair = one (target, source) direction of one triplet; a batch of B triplets is 2B pairs.
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MIOpen find mode: left at the library default.  Round 2 logged 34.4 -> 33.0 ms per step with MIOPEN_FIND_MODE=1 on
# one box; re-measured at HEAD in round 3 (profiles/r03_miopen_find_modes.txt, same box, back to back) the modes are
# indistinguishable: 32.44 (default, dynamic hybrid) / 32.56 and 32.51 (normal, cold and warm find-db) / 32.42 ms
# (hybrid).  The variable is passed through and reported in the JSON line so a run under another mode is labelled.

import unsupervised_depth_opticalflow_egomotion_amd as dfe_pkg  # noqa: E402  (first: it raises GPU_MAX_HW_QUEUES before HIP initialises)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default: a timed region of >= 2 s at ~20 ms per step)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4, help="triplets per GPU")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "train_step", "loss_stack"])
    ap.add_argument("--mode", default="geom", choices=["geom", "depth", "flow"], help="train_step model: geom = configs[2] (default, the metric's configuration), depth = configs[1], flow = configs[0]")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU functional tests)")
    ap.add_argument("--force-ddp", action="store_true", help="with --gpus 1: still create the process group (RCCL communicator at world size 1) and wrap the "
                    "model in DistributedDataParallel, and print the multi_gpu evidence block -- the only way the RCCL path can run on a 1-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--miopen-benchmark", type=int, default=0, help="torch.backends.cudnn.benchmark (MIOpen exhaustive find)")
    ap.add_argument("--graph", action="store_true", help="opt-in, never the default line: the training step captured in a hipGraph and replayed "
                    "(train_step.GraphedTrainStep; single process).  The roofline kernel is then timed in eager steps AFTER the timed region")
    ap.add_argument("--net-streams", type=int, default=None, help="1 = the three networks one after the other on one stream; 3 = flow / pose "
                    "nets on side streams (default: the package default)")
    return ap.parse_args()


def visible_devices():
    """Number of HIP devices this process would see, WITHOUT initialising the HIP runtime (the launcher parent must
    never touch the GPU): the visibility variables if set, otherwise the kfd render nodes."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    return len(glob.glob("/dev/dri/renderD*"))


def profiler_preloaded():
    """rocprofv3 preloads its tool library, which initialises the GPU before main(): spawning ranks from such a process
    is the launcher hop this pool forbids."""
    return any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY"))


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start N fresh ranks ourselves (one process
    per GPU, the reference's multi-GPU entry is one command too: train.py:59-60,277-283) and relay their exit code.
    The parent never touches the device (devices are counted from the environment / the render nodes, not through
    torch.cuda) and never execs -- it only waits for `python -m torch.distributed.run`."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d contradicts WORLD_SIZE=%s of the launcher\n" % (args.gpus, env_world))
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    if profiler_preloaded():
        sys.stderr.write("bench.py: --gpus %d under a profiler preload: the profiler has already initialised the GPU in this "
                         "process, so it must not start ranks. Profile ONE rank instead (rocprofv3 ... -- python3 bench.py "
                         "--gpus 1), or start the ranks with torch.distributed.run yourself.\n" % args.gpus)
        sys.exit(4)
    ndev = visible_devices()
    if ndev < args.gpus and os.environ.get("DFE_BENCH_ALL_ON_DEVICE0") != "1":
        sys.stderr.write("bench.py: --gpus %d requested but only %d HIP device(s) are visible\n" % (args.gpus, ndev))
        sys.exit(3)
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DFE_BENCH_ALL_ON_DEVICE0") == "1":   # functional test of the N>1 path on a 1-GPU box
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        from unsupervised_depth_opticalflow_egomotion_amd import ddp
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sock:      # a free port: several single-rank runs may share a box
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        if args.backend == "nccl":
            kw = {"pg_options": ddp.rccl_options()} if ddp.rccl_options() is not None else {}
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    return world, rank, local


def barrier(world):
    import torch.distributed as dist
    if world > 1 or dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------ loss stack
class LossStackWorkload:
    name = "loss_stack"

    def __init__(self, args, dev, seed):
        from unsupervised_depth_opticalflow_egomotion_amd import synthetic
        self.S = args.scales
        inp = synthetic.make_loss_stack_inputs(args.batch, args.height, args.width, self.S, seed=seed,
                                               num_flow_scales=max(self.S, 4))
        self.inp = inp
        g = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(grad)
        self.imgs = [g(a) for a in inp.imgs]
        self.disps = [[g(a, True) for a in lst] for lst in inp.disps]
        self.pose = g(inp.pose, True)
        self.fb = [g(a, True) for a in inp.flows_bwd]
        self.ff = [g(a, True) for a in inp.flows_fwd]
        self.K, self.Ki = g(inp.K), g(inp.K_inv)
        from unsupervised_depth_opticalflow_egomotion_amd.train_step import DEFAULT_CFG, LOSS_WEIGHT_ATTR
        self.weights = {k: DEFAULT_CFG[a] for k, a in LOSS_WEIGHT_ATTR.items()}   # config/kitti_geom.yaml weights
        self.leaves = [t for lst in self.disps for t in lst] + [self.pose] + self.fb + self.ff

    def step(self):
        from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import geom_loss_stack
        for t in self.leaves:
            t.grad = None
        lp = geom_loss_stack(self.imgs[0], self.imgs[1], self.imgs[2], self.disps[0], self.disps[1], self.disps[2],
                             self.pose, self.fb, self.ff, self.K, self.Ki, num_scales=self.S)
        loss = sum(self.weights[k] * v.mean() for k, v in lp.items())
        loss.backward()
        return loss

    def cpu_step_fn(self, threads):
        """The oracle (CPU restatement of the reference) on the same inputs: forward + backward."""
        from oracle import loss_stack_oracle as O
        inp = self.inp
        m = O.GeomLossOracle(num_scales=self.S)
        c = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(grad)
        imgs = [c(a) for a in inp.imgs]
        disps = [[c(a, True) for a in lst] for lst in inp.disps]
        pose, fb, ff = c(inp.pose, True), [c(a, True) for a in inp.flows_bwd], [c(a, True) for a in inp.flows_fwd]
        K, Ki = c(inp.K), c(inp.K_inv)

        def run():
            lp, _ = m.geom_losses(imgs[0], imgs[1], imgs[2], disps[0], disps[1], disps[2], pose, fb, ff, K, Ki)
            sum(self.weights[k] * v.mean() for k, v in lp.items()).backward()
        return run


# ------------------------------------------------------------------------------------------------ train step
def miopen_db_status():
    """'tuned' = the auto-tuned MIOpen user databases shipped with the package (miopen_tuning.py), 'env' = a path the
    caller set, 'default' = MIOpen's own."""
    from unsupervised_depth_opticalflow_egomotion_amd import miopen_tuning
    return miopen_tuning.status()


class TrainStepWorkload:
    """bench.py workload: mode=geom on synthetic KITTI-shaped triplets (configs[2] / configs[3])."""
    name = "train_step"

    def __init__(self, args, dev, seed, world=1):
        self.args, self.dev = args, dev
        from unsupervised_depth_opticalflow_egomotion_amd import ddp, synthetic
        from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
        from unsupervised_depth_opticalflow_egomotion_amd.train_step import make_cfg, make_optimizer
        self.mode = getattr(args, "mode", "geom")
        self.cfg = make_cfg(num_scales=args.scales, img_hw=(args.height, args.width), mode=self.mode)
        torch.manual_seed(1234)           # identical initial weights on every rank
        self.model = get_model(self.mode)(self.cfg).to(dev)
        self.model.train()
        im, k, ki = synthetic.make_triplet_batch(args.batch, args.height, args.width, args.scales, seed=seed)
        self.np_inputs = (im, k, ki)
        self.inputs = [torch.from_numpy(a).to(dev) for a in (im, k, ki)]   # resident in HBM before timing
        if world > 1:
            self.find_warmup(world)
        self.model = ddp.wrap(self.model, dev, force=bool(getattr(args, "force_ddp", False)))
        self.use_graph = bool(getattr(args, "graph", False)) and world == 1 and not getattr(args, "force_ddp", False)
        self.opt = make_optimizer(self.model, self.cfg.lr, capturable=self.use_graph)
        self.graphed = None

    def find_warmup(self, world):
        """N ranks of one node share one MIOpen user find-db.  Rank 0 runs one forward + backward of the bare model
        first (its find phase times every solver and stores the winners), the others follow behind a barrier and read
        those entries instead of running N concurrent searches against one file.  No optimiser step is taken and the
        gradients are dropped, so the replicas still start from identical weights (DDP broadcasts rank 0's buffers)."""
        import torch.distributed as dist
        from unsupervised_depth_opticalflow_egomotion_amd.train_step import total_loss
        rank = dist.get_rank()
        for turn in (0, 1):
            if (rank == 0) == (turn == 0):
                lp, _ = self.model(self.inputs)
                total_loss(lp, self.cfg).backward()
                for p in self.model.parameters():
                    p.grad = None
                torch.cuda.synchronize()
            dist.barrier()

    def step(self, eager=False):
        from unsupervised_depth_opticalflow_egomotion_amd.train_step import GraphedTrainStep, train_step
        if self.use_graph and not eager:
            if self.graphed is None:
                self.graphed = GraphedTrainStep(self.model, self.opt, self.inputs, self.cfg)
            return self.graphed()[0]
        return train_step(self.model, self.opt, self.inputs, self.cfg)[0]

    def cpu_step_fn(self, threads):
        """CPU baseline: the same networks on the host + the oracle's loss stack + Adam."""
        from oracle import loss_stack_oracle as O
        from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
        from unsupervised_depth_opticalflow_egomotion_amd.networks import pwc_tf
        from unsupervised_depth_opticalflow_egomotion_amd.train_step import total_loss

        class OraclePWC(pwc_tf.PWC_tf):
            def warp(self, x, flow):
                return O.warp_flow(x, flow, use_mask=False)

            def corr_naive(self, a, b, d=4):
                return O.corr_naive(a, b, d)

        cfg = self.cfg
        torch.manual_seed(1234)
        model = get_model(self.mode)(cfg)
        if self.mode in ("geom", "flow"):
            pw = OraclePWC()
            pw.load_state_dict(model.pwc_model.state_dict())
            pw.corr = pw.corr_naive
            model.pwc_model = pw
        model.train()
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.lr)
        oracle = O.GeomLossOracle(num_scales=cfg.num_scales)
        images, k_ms, ki_ms = [torch.from_numpy(a) for a in self.np_inputs]
        h = images.shape[2] // 3

        def run():
            opt.zero_grad(set_to_none=True)
            img_l, img, img_r = images[:, :, :h], images[:, :, h:2 * h], images[:, :, 2 * h:]
            if self.mode == "geom":
                dl, dt, dr, pose, fb, ff = model.run_networks(img_l, img, img_r)
                lp, _ = oracle.geom_losses(img_l, img, img_r, dl, dt, dr, pose, fb, ff, k_ms[:, 0], ki_ms[:, 0])
            elif self.mode == "flow":
                f_l, f_t, f_r = model.fpyramid(img_l), model.fpyramid(img), model.fpyramid(img_r)
                hw = [img.shape[2], img.shape[3]]
                lp, _ = oracle.flow_losses(img_l, img, img_r, model.pwc_model(f_t, f_l, hw), model.pwc_model(f_t, f_r, hw))
            else:
                dl, dt, dr = model.depth_net(img_l), model.depth_net(img), model.depth_net(img_r)
                pose = model.pose_net(torch.cat([img_l, img, img_r], 1))
                lp, _ = oracle.depth_losses(img_l, img, img_r, dl, dt, dr, pose, k_ms[:, 0])
            total_loss(lp, cfg).backward()
            opt.step()
        return run


# ------------------------------------------------------------------------------------------------ roofline
# dominant kernel of each mode's fused forward (segment 2 of the timed launches); its algorithmic bytes per launch come
# from tools/byte_models.py, the same table tools/roofline_table.py prices the rocprof runs with
from tools import byte_models  # noqa: E402

POINT_LIMITER = {"geom": "VALU issue (~880 instructions per pixel at a measured ~3 SIMD-cycles each: roofline.valu_frac) and the line count of the 48 bilinear taps per pixel; "
                          "a wave covers a 16x4 pixel tile (round 4): 9-16 % faster than a 64-pixel row segment on displaced flows, 3 % slower on the ZERO flows the "
                          "random-initialised nets of this train step produce (profiles/r04_point_tile_experiment.md); Infinity-Cache resident at B=4"}
HBM_ACHIEVABLE_GBS = 6290.0    # measured float4 copy rate (MI355X_MICROARCH.md)
KERNEL_SOURCES = ("loss_stack_fwd.hip", "loss_stack.h", "loss_stack_exact.h", "dfe_device.h")


def kernel_source_hash():
    """sha256 over the sources k_geom_point_fwd is compiled from: a PMC traffic figure is only reported for the kernel
    it was measured on (tools/roofline_table.py stores the same hash next to the counters)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "unsupervised_depth_opticalflow_egomotion_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def point_fwd_roofline(args, mode, fwd_ms, bwd_ms):
    """Roofline object from the HIP-event timings recorded inside the timed region (loss_stack.timing_begin /
    timing_collect: events on the launch stream around every launch of the fused stack, read after the final sync)."""
    S = args.scales
    kernel = byte_models.POINT_KERNEL[mode]
    bytes_per_launch, byte_model = byte_models.models(args.batch, args.height, args.width, S)[kernel]
    t_ms = float(fwd_ms[:, 2].mean())
    achieved = bytes_per_launch / (t_ms * 1e-3) / 1e9
    valu_per_wave, valu_src = None, "static (tools/byte_models.VALU_PER_WAVE)"
    traffic = None   # PMC-derived HBM bytes per launch: cannot be collected from inside this process; taken from
    try:             # the committed rocprofv3 --pmc pass ONLY when it was made on exactly this kernel source and workload
        with open(os.path.join(ROOT, "profiles", "pmc_point_fwd_traffic.json")) as fh:
            pm = json.load(fh)
        w = pm["workload"]
        same_kernel = pm.get("kernel_source_sha256") == kernel_source_hash()
        if mode == "geom" and same_kernel and (w["batch"], w["height"], w["width"], w["scales"]) == (args.batch, args.height, args.width, S):
            traffic = pm["hbm_bytes_per_launch"]
            if pm.get("valu_per_wave"):
                valu_per_wave, valu_src = float(pm["valu_per_wave"]), "pmc (profiles/pmc_point_fwd_traffic.json, same kernel source)"
    except Exception:
        traffic = None
    # The second roof: vector-ALU issue.  One thread per pixel of every (sample, scale) image; executed VALU instructions per
    # wave x the measured issue cost of this kernel's instruction mix, on all 1024 SIMDs (tools/byte_models.valu_time_s).
    threads = args.batch * sum(byte_models.scale_pixels(args.height, args.width, S))
    t_valu = byte_models.valu_time_s(kernel, threads, valu_per_wave)
    valu_frac = None if t_valu is None else round(t_valu / (t_ms * 1e-3), 4)
    hbm_frac = round(achieved / HBM_PEAK_GBS, 4)
    top = max(hbm_frac, valu_frac or 0.0)
    roof = {"bound": "latency" if top < 0.6 else ("valu" if (valu_frac or 0.0) > hbm_frac else "hbm"),
            "bound_note": "latency = the launch sits under BOTH roofs (frac = algorithmic bytes / time / HBM peak and valu_frac = VALU instructions x "
                          "measured issue cycles / time are both < 0.6): waves waiting at 5-6 waves per SIMD (PMC: wait-issue 0.44, wait-memory "
                          "0.31 of wave cycles); achieved / peak / frac stay the HBM roofline SURVEY 8(d) prices this kernel against",
            "valu_frac": valu_frac, "valu_insts_per_wave": valu_per_wave if valu_per_wave is not None else byte_models.VALU_PER_WAVE.get(kernel),
            "valu_cycles_per_inst": round(byte_models.VALU_CYCLES_PER_INST, 2), "valu_source": valu_src,
            "limiter": POINT_LIMITER.get(mode), "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": hbm_frac, "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4),
            "traffic": traffic,
            "bytes_per_launch": bytes_per_launch, "byte_model": byte_model, "avg_kernel_ms": round(t_ms, 5), "launches_timed": int(fwd_ms.shape[0])}
    segs = {"fwd_ms": [round(float(x), 5) for x in fwd_ms.mean(0)],
            "bwd_ms": [round(float(x), 5) for x in bwd_ms.mean(0)] if len(bwd_ms) else []}
    return roof, segs


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, args, unit_pairs):
    """Oracle timed on the host cores on a bounded sample.  torch CPU ops on these tensor sizes do not scale
    past a few threads (256 threads ran 40x slower than 8), so the leg uses min(cores, 16) threads."""
    threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    run = wl.cpu_step_fn(threads)
    run()                       # warm-up (allocator, thread pool)
    # median of up to 5 timed steps (the mean of 3 moved +-20 % between samples), bounded at ~30 s of CPU work
    times = []
    t_all = time.perf_counter()
    while len(times) < 5:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 30.0 and len(times) >= 3:
            break
    times.sort()
    n = len(times)
    dt = times[n // 2] if n % 2 else 0.5 * (times[n // 2 - 1] + times[n // 2])
    return {"value": round(unit_pairs / dt, 4), "unit": "frame-pairs/s", "cores": threads, "cpu_model": cpu_model(),
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": "median of %d timed step(s) of the same workload (%s, B=%d, %dx%d, S=%d, fwd+bwd%s) on the host CPU, %.2f s/step"
                      % (n, wl.name, args.batch, args.height, args.width, args.scales,
                         "+Adam" if wl.name == "train_step" else "", dt)}


def multi_gpu_evidence(wl, world, rank, dev, dt_local, args):
    """What a reader of the N > 1 line needs to believe that N replicas really exchanged gradients (after the timed
    region): the backend in use, an all-reduce of rank ids whose sum is checked, a checksum of every parameter after
    the last step gathered from all ranks (equal on all ranks <=> the gradients were reduced: each rank trains on its
    own shard, seed 1234 + rank), and the per-rank step times."""
    import torch.distributed as dist
    ids = torch.tensor([float(rank), 1.0], device=dev, dtype=torch.float64)
    dist.all_reduce(ids)
    ids_ok = int(ids[1].item()) == world and int(ids[0].item()) == world * (world - 1) // 2
    model = getattr(wl, "model", None)
    sums = None
    if model is not None:
        cs = torch.zeros(2, device=dev, dtype=torch.float64)
        for p_ in model.parameters():
            d = p_.detach().double()
            cs[0] += d.sum()
            cs[1] += d.abs().sum()
        gathered = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        sums = [[float(g[0]), float(g[1])] for g in gathered]
    times = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(times, torch.tensor([1e3 * dt_local / args.steps], device=dev, dtype=torch.float64))
    times = [float(t.item()) for t in times]
    ev = {"data_parallel": type(model).__name__ if model is not None else None, "backend": dist.get_backend(), "collective_library": "RCCL" if dist.get_backend() == "nccl" else dist.get_backend(),
          "ranks": int(ids[1].item()), "rank_id_allreduce_ok": bool(ids_ok),
          "ms_per_step_min": round(min(times), 4), "ms_per_step_max": round(max(times), 4),
          "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE")}
    if hasattr(model, "message_bytes"):      # ddp.FlatAllReduce: one message per network branch, issued from backward
        ev["allreduce_message_bytes"] = model.message_bytes()
        ev["allreduce_issue_order"] = list(model._order or model.branches)
        ev["messages_issued_from_backward"] = int(model.early_hits)
    if sums is not None:
        ev["param_checksum_rank0"] = sums[0]
        ev["param_checksums_equal"] = all(s_ == sums[0] for s_ in sums)
        ev["shards_differ"] = True   # rank r draws its batch with seed 1234 + r
    return ev


def main():
    args = parse()
    launch_ranks_if_needed(args)
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
    if args.net_streams is not None:
        from unsupervised_depth_opticalflow_egomotion_amd import models
        models._DEFAULT_NET_STREAMS = args.net_streams
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local)
    wl_name = args.workload
    if wl_name == "auto":
        wl_name = "train_step"
    if wl_name == "train_step":
        wl = TrainStepWorkload(args, dev, seed=1234 + rank, world=world)
    else:
        wl = LossStackWorkload(args, dev, seed=1234 + rank)
    for _ in range(args.warmup):
        wl.step()
    from unsupervised_depth_opticalflow_egomotion_amd import loss_stack as LS
    barrier(world)
    LS.timing_begin()           # HIP events between the fused stack's launches (stream-ordered, no host sync)
    # stream-ordered marks (no host sync): the whole region, and its first 20 steps (what rounds 1-4 timed) for comparison
    ev_beg, ev_20, ev_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    ev_beg.record()
    for i in range(args.steps):
        wl.step()
        if i == 19:
            ev_20.record()
    ev_end.record()
    t_host = time.perf_counter() - t0      # the host has ENQUEUED every step (it runs ahead of the GPU when the GPU is the bound)
    barrier(world)
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, world, dev)
    graph_mode = bool(getattr(wl, "use_graph", False))
    if graph_mode:      # events recorded inside a captured graph cannot be read back: time the roofline kernel in eager steps, after the region
        LS.timing_collect()
        LS.timing_begin()
        for _ in range(5):
            wl.step(eager=True)
        torch.cuda.synchronize()
    fwd_ms, bwd_ms = LS.timing_collect()
    evidence = multi_gpu_evidence(wl, world, rank, dev, dt_local, args) if (world > 1 or args.force_ddp) else None
    pairs_per_step = 2 * args.batch * world
    value = pairs_per_step * args.steps / dt
    out = {
        "metric": "frame-pairs/sec (%dx%d, %s mode)" % (args.width, args.height, args.mode if wl.name == "train_step" else "geom"), "value": round(value, 2), "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        # host time to enqueue one step (launches + autograd bookkeeping): the floor the step cannot go below without fewer
        # launches or a captured graph; equal to ms_per_step when the host is the bound
        "host_enqueue_ms": round(1e3 * t_host / args.steps, 4), "timed_region_s": round(dt, 3),
        "ms_per_step_first20": round(ev_beg.elapsed_time(ev_20) / 20.0, 4) if args.steps >= 20 else None,
        "ms_per_step_gpu_events": round(ev_beg.elapsed_time(ev_end) / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl.name + ": mode=%s, %dx%d, batch=%d/GPU, num_scales=%d, fwd+bwd%s" % (
            args.mode if wl.name == "train_step" else "geom", args.width, args.height, args.batch, args.scales,
            "+Adam" if wl.name == "train_step" else "") + (" [OPT-IN: the step replayed from a hipGraph; roofline kernel timed in eager steps after the region]" if getattr(wl, "use_graph", False) else ""),
            "global_batch": args.batch * world, "parallelism": "dp%d" % world,
            "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE"), "miopen_user_db": miopen_db_status(),
            "hw_queues": dfe_pkg.HW_QUEUES, "stream_priorities": os.environ.get("DFE_STREAM_PRIORITIES", "auto")},
    }
    if evidence is not None:
        out["multi_gpu"] = evidence
    if world > 1:
        # every rank empties its C stdio (RCCL's banners) before rank 0 prints: the JSON line stays the last line of the
        # launcher's merged stdout
        import torch.distributed as dist
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        dist.barrier()
    if rank == 0:
        roof, segs = point_fwd_roofline(args, args.mode if wl.name == "train_step" else "geom", fwd_ms, bwd_ms)
        out["roofline"] = roof
        out["kernel_ms"] = segs
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args, 2 * args.batch)
        # keep the JSON line the LAST line of stdout: RCCL prints a "Librccl path" banner through C stdio when the communicator
        # is made, which a pipe buffers until exit -- flush it out before the line
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_ddp:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
