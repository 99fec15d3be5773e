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


def init_process_group(backend=None):
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def unused_parameter_names(model):
    """``depth_net.encoder.encoder.fc.{weight,bias}`` (513 k parameters) are never used in forward
    (depth_model.py:85-95) and never receive a gradient in the reference either.  They stay ``requires_grad=True`` and
    stay in the optimizer's single parameter group -- the reference builds Adam over all parameters (train.py:85-87),
    so checkpoints' ``optimizer_state_dict`` keep the reference's parameter indexing; Adam skips ``grad is None``."""
    return [n for n, _ in model.named_parameters()
            if ".encoder.encoder.fc." in n or n.startswith("encoder.encoder.fc.")]


def wrap(model, device=None, bucket_cap_mb=25):
    """DistributedDataParallel when WORLD_SIZE > 1, the bare module otherwise.  The never-used ``fc`` parameters are
    excluded from the reducer (it would wait for their gradients forever) instead of being frozen: 21.06 M of the
    model's 21.57 M parameters are all-reduced (84 MB fp32 per step)."""
    world, _, local = env_world()
    if world == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    DDP._set_params_and_buffers_to_ignore_for_model(model, unused_parameter_names(model))
    if device is not None and device.type == "cuda":
        return DDP(model, device_ids=[device.index], output_device=device.index, bucket_cap_mb=bucket_cap_mb,
                   gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=bucket_cap_mb)


def shard_indices(n, world, rank):
    """Disjoint strided index ranges (DistributedSampler-style, no padding)."""
    return list(range(rank, n, world))


def unwrap(model):
    return model.module if hasattr(model, "module") else model
