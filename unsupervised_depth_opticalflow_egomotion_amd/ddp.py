"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (train.py:59-60,277-283).  Every loss term is a
per-sample (B,) vector with per-sample normalisers, so equal shards + gradient averaging reproduce the
single-process gradient.  One collective per step: the all-reduce of 21.06 M fp32 gradients (84 MB:
the model's 21.57 M parameters minus the never-used 513 k of ``depth_net.encoder.encoder.fc``) -- as ONE flat message
after backward (``FlatAllReduce``, the default since round 4) or bucketed and overlapped with backward by torch's
DistributedDataParallel (``strategy="torch"``).  The loss-stack kernels themselves are single-GPU; nothing on the data
path is exchanged."""
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


class FlatAllReduce(torch.nn.Module):
    """The model's replica in a data-parallel job, round 4's default (``wrap(strategy="flat")``).

    One collective per step, issued AFTER backward: the gradients are copied into ONE persistent flat fp32 buffer (a
    single multi-tensor copy kernel), all-reduced in place as one 84 MB message -- the size RCCL's ring over xGMI is most
    efficient at -- and handed to the optimiser as views of that buffer (no copy back: ``optim.FusedAdam`` refreshes its
    gradient pointers every step).  ``make_optimizer`` hooks ``reduce_gradients`` in front of ``optimizer.step()``, so the
    reference's loop (``loss.backward(); optimizer.step()``, train.py:215-216) needs no extra call.

    Why not torch's DistributedDataParallel (``strategy="torch"`` keeps it): measured on MI355X at world size 1 around the
    real joint model (profiles/r04_ddp_overhead.txt; plain step 25.3 ms) its reducer costs 2.5-3.1 ms per step in
    per-parameter autograd hooks and bookkeeping (250 parameters) before any byte is communicated -- a weak-scaling
    ceiling of 0.89-0.91 -- and its ``static_graph`` mode, which would halve that, silently stops reducing this model's
    gradients on the device (cross-rank parameter checksums diverge: bench.py's multi_gpu evidence caught it).  This
    class costs 0.1 ms per step at world size 1; what it gives up is the overlap of the all-reduce with backward
    (~1 ms of ring time at 8 GPUs, exposed: expected efficiency ~0.96).

    The never-used ``fc`` parameters are left out (no gradient ever exists for them).  Parameters and buffers start from
    rank 0's values (one coalesced broadcast at construction); BatchNorm's running statistics are NOT re-broadcast every
    forward: train-mode forwards do not read them and rank 0's are what checkpoints hold either way (the reference's
    DataParallel keeps device 0's and discards the replicas', train.py:59-60)."""

    def __init__(self, module, ignore=()):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self.backend = dist.get_backend()
        self.ignored = sorted(ignore)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and n not in set(ignore)]
        self._params = [p for _, p in named]
        self._numel = sum(p.numel() for p in self._params)
        self._flat = None
        self._views = None
        tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
        if tensors and self.world > 1:
            dist._broadcast_coalesced(dist.group.WORLD, tensors, 250 * 1024 * 1024, 0)
            if tensors[0].is_cuda:
                from . import ops
                ops.wino_weights.invalidate()       # written through .data: no version bump for the cached filters to miss on

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _buffers_for(self, ref):
        if self._flat is None or self._flat.device != ref.device:
            self._flat = torch.zeros(self._numel, device=ref.device, dtype=torch.float32)
            self._views, off = [], 0
            for p in self._params:
                self._views.append(self._flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._flat, self._views

    @torch.no_grad()
    def reduce_gradients(self):
        """Average the gradients over the ranks (call once after backward; ``make_optimizer`` does it before every step)."""
        grads = [p.grad for p in self._params]
        if all(g is not None for g in grads):        # the steady state: every reducible parameter has a gradient
            flat, views = self._buffers_for(grads[0])
            torch._foreach_copy_(views, grads)
            have = None
        else:
            have = [i for i, g in enumerate(grads) if g is not None]
            if not have:
                return
            flat, views = self._buffers_for(grads[have[0]])
            present = set(have)
            for i, v in enumerate(views):            # a parameter without a gradient this step contributes zeros
                if i not in present:
                    v.zero_()
            torch._foreach_copy_([views[i] for i in have], [grads[i] for i in have])
        if self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat)
            flat.div_(self.world)
        for i in (range(len(views)) if have is None else have):
            self._params[i].grad = views[i]


def wrap(model, device=None, bucket_cap_mb=25, force=False, strategy=None):
    """The data-parallel replica of ``model`` when WORLD_SIZE > 1 (or ``force``: the same machinery over a world-size-1
    group), the bare module otherwise.  ``strategy``: "flat" (default, FlatAllReduce above) or "torch" (torch's
    DistributedDataParallel with 25 MB buckets overlapped with backward, gradients as bucket views); DFE_DP_STRATEGY
    overrides.  Either way the never-used ``fc`` parameters are excluded from the reduction instead of being frozen:
    21.06 M of the model's 21.57 M parameters are all-reduced (84 MB fp32 per step)."""
    world, _, local = env_world()
    if world == 1 and not force:
        return model
    strategy = os.environ.get("DFE_DP_STRATEGY", strategy or "flat")
    if strategy == "flat":
        return FlatAllReduce(model, unused_parameter_names(model))
    from torch.nn.parallel import DistributedDataParallel as DDP
    DDP._set_params_and_buffers_to_ignore_for_model(model, unused_parameter_names(model))
    # broadcast_buffers=False: the only buffers are BatchNorm's running statistics (see FlatAllReduce).  static_graph stays
    #   False: with it this model's gradients are no longer reduced on the device (see FlatAllReduce).
    # DFE_DDP_OPTS: experiment switches, e.g. "broadcast_buffers=1,static_graph=1,bucket_cap_mb=100"
    kw = dict(bucket_cap_mb=bucket_cap_mb, static_graph=False, broadcast_buffers=False)
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
