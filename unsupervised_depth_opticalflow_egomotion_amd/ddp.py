"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (train.py:59-60,277-283).  Every loss term is a
per-sample (B,) vector with per-sample normalisers, so equal shards + gradient averaging reproduce the
single-process gradient.  One collective per step: the bucketed all-reduce of 21.06 M fp32
gradients (84 MB: the model's 21.57 M parameters minus the never-used 513 k of ``depth_net.encoder.encoder.fc``),
overlapped with backward by DistributedDataParallel.  The loss-stack
kernels themselves are single-GPU; nothing on the data path is exchanged."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def rccl_options():
    """Process-group options for backend "nccl" (= RCCL): the communicator's stream is a HIGH-priority HIP stream.
    models.run_networks puts the flow and pose branches on high-priority streams (the flow branch is the step's long
    pole); DDP's reducer enqueues every bucket's all-reduce on the process group's own stream while
    backward is still running, and a default-priority stream gets its kernels scheduled only when no high-priority
    queue has a wave ready -- the buckets would drain after backward instead of under it."""
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except AttributeError:          # a torch build without the NCCL/RCCL process group
        return None


def init_process_group(backend=None, force=False):
    """One process per GPU: reads RANK / LOCAL_RANK / WORLD_SIZE.  ``force`` initialises the group at world size 1 too
    (the RCCL communicator, DDP's reducer and its bucket views then run on a single GPU: tests, bench.py --force-ddp)."""
    world, rank, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
            if rccl_options() is not None:
                kw["pg_options"] = rccl_options()
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def unused_parameter_names(model):
    """``depth_net.encoder.encoder.fc.{weight,bias}`` (513 k parameters) are never used in forward
    (depth_model.py:85-95) and never receive a gradient in the reference either.  They stay ``requires_grad=True`` and
    stay in the optimizer's single parameter group -- the reference builds Adam over all parameters (train.py:85-87),
    so checkpoints' ``optimizer_state_dict`` keep the reference's parameter indexing; Adam skips ``grad is None``."""
    return [n for n, _ in model.named_parameters()
            if ".encoder.encoder.fc." in n or n.startswith("encoder.encoder.fc.")]


def wrap(model, device=None, bucket_cap_mb=25, force=False):
    """DistributedDataParallel when WORLD_SIZE > 1 (or ``force``: DDP over a world-size-1 group, which still builds the
    reducer, the bucket views and the communicator), the bare module otherwise.  The never-used ``fc`` parameters are
    excluded from the reducer (it would wait for their gradients forever) instead of being frozen: 21.06 M of the
    model's 21.57 M parameters are all-reduced (84 MB fp32 per step)."""
    world, _, local = env_world()
    if world == 1 and not force:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    DDP._set_params_and_buffers_to_ignore_for_model(model, unused_parameter_names(model))
    # static_graph: the set of parameters that receive a gradient never changes (the ignored fc pair aside), so the
    #   reducer can skip its per-iteration bookkeeping.  broadcast_buffers=False: the only buffers are BatchNorm's running
    #   statistics; train-mode forwards do not read them, rank 0's are what checkpoints hold either way (the reference's
    #   DataParallel keeps device 0's and discards the replicas', train.py:59-60), so the per-forward broadcast buys
    #   nothing.  Measured at world size 1 on MI355X (bench.py --force-ddp, plain step 25.3 ms): DDP defaults 28.4 ms,
    #   static_graph 26.9, both 26.5 (profiles/r04_ddp_overhead.txt).
    # DFE_DDP_OPTS: experiment switches, e.g. "broadcast_buffers=1,static_graph=0,bucket_cap_mb=100"
    kw = dict(bucket_cap_mb=bucket_cap_mb, static_graph=True, broadcast_buffers=False)
    for item in filter(None, os.environ.get("DFE_DDP_OPTS", "").split(",")):
        k, v = item.split("=")
        kw[k] = float(v) if k == "bucket_cap_mb" else bool(int(v))
    if device is not None and device.type == "cuda":
        kw.setdefault("gradient_as_bucket_view", True)
        return DDP(model, device_ids=[device.index], output_device=device.index, **kw)
    return DDP(model, **kw)


def shard_indices(n, world, rank):
    """Disjoint strided index ranges (DistributedSampler-style, no padding)."""
    return list(range(rank, n, world))


def unwrap(model):
    return model.module if hasattr(model, "module") else model
