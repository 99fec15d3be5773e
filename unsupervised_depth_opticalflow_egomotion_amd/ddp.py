"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (train.py:59-60,277-283).  Every loss term is a
per-sample (B,) vector with per-sample normalisers, so equal shards + gradient averaging reproduce the
single-process gradient.  One collective per step: the bucketed all-reduce of the
21.57 M fp32 gradients (86 MB), overlapped with backward by DistributedDataParallel.  The loss-stack
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


def freeze_unused(model):
    """``depth_net.encoder.encoder.fc`` (513 k parameters) is never used in forward (depth_model.py:85-95) and
    never receives a gradient in the reference either; excluding it keeps DDP's reducer from waiting on it."""
    for name, p in model.named_parameters():
        if ".encoder.encoder.fc." in name or name.startswith("encoder.encoder.fc."):
            p.requires_grad_(False)
    return model


def wrap(model, device=None, bucket_cap_mb=25):
    """DistributedDataParallel when WORLD_SIZE > 1, the bare module otherwise."""
    world, _, local = env_world()
    freeze_unused(model)
    if world == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    if device is not None and device.type == "cuda":
        return DDP(model, device_ids=[device.index], output_device=device.index, bucket_cap_mb=bucket_cap_mb,
                   gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=bucket_cap_mb)


def shard_indices(n, world, rank):
    """Disjoint strided index ranges (DistributedSampler-style, no padding)."""
    return list(range(rank, n, world))


def unwrap(model):
    return model.module if hasattr(model, "module") else model
