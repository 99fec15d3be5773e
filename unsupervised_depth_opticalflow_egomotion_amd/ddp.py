"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (train.py:59-60,277-283).  Every loss term is a
per-sample (B,) vector with per-sample normalisers, so equal shards + gradient averaging reproduce the
single-process gradient.  The one exchange per step is the all-reduce of 21.06 M fp32 gradients (84 MB:
the model's 21.57 M parameters minus the never-used 513 k of ``depth_net.encoder.encoder.fc``) -- as one flat message per
network branch, issued as that branch's backward pass ends (``FlatAllReduce``, the default) or bucketed and overlapped
with backward by torch's DistributedDataParallel (``strategy="torch"``).  The loss-stack kernels themselves are single-GPU; nothing on the data
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


def _branch_of(name):
    """The network a parameter belongs to: the three branches of the joint model run their backward passes on their own
    streams (models.run_networks) and finish at different times."""
    head = name.split(".", 1)[0]
    return "flow" if head in ("fpyramid", "pwc_model") else head


class FlatAllReduce(torch.nn.Module):
    """The model's replica in a data-parallel job, the default (``wrap(strategy="flat")``).

    The gradients live in ONE persistent flat fp32 buffer, one contiguous segment per network branch (depth / pose / flow:
    57 / 5 / 22 MB of the joint model's 84 MB), and each segment is all-reduced as one message -- the sizes RCCL's ring
    over xGMI is efficient at -- and handed to the optimiser as views of the buffer (no copy back: ``optim.FusedAdam``
    refreshes its gradient pointers every step).

    Overlap without per-parameter hooks (round 5).  torch's DistributedDataParallel pays 2.5-3.1 ms per step for its 250
    autograd hooks (profiles/r04_ddp_overhead.txt).  Here ONE hook per branch sits on the parameter whose gradient arrives
    last in that branch (found on the first backward pass: the first layer's weight): when it fires, every gradient of the
    branch has been enqueued, so the branch's segment is filled by one multi-tensor copy and its all-reduce is issued with
    ``async_op=True`` right there -- the communicator's (high-priority) stream waits for the stream the hook runs on and the
    ring runs under the other branches' backward kernels.  ``reduce_gradients`` (hooked in front of ``optimizer.step()`` by
    ``make_optimizer``; the reference's loop, train.py:215-216, needs no extra call) only waits for the three messages;
    whatever was not issued by then (first step, a branch without gradients) is reduced there.  At 8 GPUs only the branch
    that finishes last stays exposed.  Safety nets: the collectives are always issued in ONE branch order on every rank --
    the static module order until the ranks have AGREED on a calibrated one (round 6: ``_agree``, one 8-byte-per-branch
    collective per step while calibrating, none afterwards: the triggers are armed only once EVERY rank has seen a first
    backward pass that produced every branch's gradients, and the order every rank then uses is rank 0's; a rank whose
    first steps lack a branch keeps the whole job in calibration, it does not make the ranks issue differently sized
    messages in different orders); a gradient that changes after its branch was reduced (a second backward pass, late
    accumulation, a backward pass after a skipped optimiser step) is detected -- the branch's trigger fires again, or
    identity + version differ -- and the branch is reduced again; every segment carries a presence word per parameter, so
    ranks whose sets of parameters with gradients differ neither hang nor diverge (a parameter with a gradient on ANY rank
    gets the averaged gradient on EVERY rank, as with DistributedDataParallel; one without a gradient anywhere keeps
    ``None``).  What the ranks must share is the PROGRAM (the same number of backward passes per step), as with
    DistributedDataParallel.

    The never-used ``fc`` parameters are left out (no gradient ever exists for them).  Parameters and buffers start from
    rank 0's values (one coalesced broadcast at construction); BatchNorm's running statistics are NOT re-broadcast every
    forward: train-mode forwards do not read them and rank 0's are what checkpoints hold either way (the reference's
    DataParallel keeps device 0's and discards the replicas', train.py:59-60)."""

    def __init__(self, module, ignore=(), overlap=None):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self.backend = dist.get_backend()
        self.ignored = sorted(ignore)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and n not in set(ignore)]
        # flat order: branch by branch (first appearance), parameters in module order inside a branch
        order = []
        for n, _ in named:
            if _branch_of(n) not in order:
                order.append(_branch_of(n))
        named.sort(key=lambda t: order.index(_branch_of(t[0])))
        self._names = [n for n, _ in named]
        self._params = [p for _, p in named]
        self.branches = order
        self._members = {k: [i for i, n in enumerate(self._names) if _branch_of(n) == k] for k in order}
        self._numel = sum(p.numel() for p in self._params)
        self._flat = None
        self._views = None
        self._seg = None                 # branch -> (offset, gradient floats, total floats incl. the presence words)
        self.overlap = (os.environ.get("DFE_DP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self._order = None               # the fixed order the branches' collectives are issued in (= arrival order, step 1)
        self._arrival = []               # calibration (first backward): parameter indices in arrival order
        self._hooks = []
        self._ready, self._issued, self._work, self._sig = set(), [], {}, {}
        self._dirty, self._events = set(), {}
        self.calibration_steps = 0       # reduce_gradients calls spent agreeing on the order (statistics / tests)
        self.early_hits = 0              # branches whose all-reduce was issued from backward (statistics / tests)
        if self.overlap:
            for i, p in enumerate(self._params):
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._arrival.append(i)))
        tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
        if tensors and self.world > 1:
            dist._broadcast_coalesced(dist.group.WORLD, tensors, 250 * 1024 * 1024, 0)
            if tensors[0].is_cuda:
                from . import ops
                ops.wino_weights.invalidate()       # written through .data: no version bump for the cached filters to miss on

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def message_bytes(self):
        """Per-branch all-reduce message sizes in bytes (gradients + one presence word per parameter)."""
        return {k: 4 * (sum(self._params[i].numel() for i in self._members[k]) + len(self._members[k])) for k in self.branches}

    def _buffers_for(self, ref):
        if self._flat is None or self._flat.device != ref.device:
            total = self._numel + len(self._params)
            self._flat = torch.zeros(total, device=ref.device, dtype=torch.float32)
            self._views, self._seg, self._present, off = [None] * len(self._params), {}, {}, 0
            for k in self.branches:
                beg = off
                for i in self._members[k]:
                    p = self._params[i]
                    self._views[i] = self._flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
                self._present[k] = self._flat[off:off + len(self._members[k])]
                off += len(self._members[k])
                self._seg[k] = (beg, off - beg)
        return self._flat, self._views

    # ------------------------------------------------------------------ one branch
    def _issue(self, k):
        """Fill branch k's segment from the gradients that exist and start its all-reduce (asynchronous)."""
        idx = self._members[k]
        grads = [self._params[i].grad for i in idx]
        ref = next((g for g in grads if g is not None), self._params[idx[0]])
        flat, views = self._buffers_for(ref)
        ev = self._events.pop(k, None)
        if ev is not None:      # issued from ANOTHER branch's hook (it was blocked by the fixed order): this stream has not
            torch.cuda.current_stream().wait_event(ev)      # waited for the one that produced branch k's gradients
        have = [j for j, g in enumerate(grads) if g is not None]
        if len(have) == len(idx):
            torch._foreach_copy_([views[i] for i in idx], grads)
            self._present[k].fill_(1.0)
        else:
            present = torch.zeros(len(idx))
            for j in have:
                present[j] = 1.0
            self._present[k].copy_(present)
            for j, i in enumerate(idx):
                if grads[j] is None:
                    views[i].zero_()              # a parameter without a gradient on this rank contributes zeros
            if have:
                torch._foreach_copy_([views[idx[j]] for j in have], [grads[j] for j in have])
        beg, n = self._seg[k]
        seg = flat[beg:beg + n]
        if self.backend == "nccl":
            self._work[k] = (dist.all_reduce(seg, op=dist.ReduceOp.AVG, async_op=True), seg, len(have) == len(idx))
        else:
            self._work[k] = (dist.all_reduce(seg, async_op=True), seg, len(have) == len(idx))
        self._sig[k] = self._signature(k)
        self._issued.append(k)

    def _signature(self, k):
        """Identity + version of the branch's first and last gradients: a second backward pass (or any in-place accumulation)
        after the branch's message left changes them."""
        idx = self._members[k]
        return tuple((id(g), g._version) if g is not None else None for g in (self._params[idx[0]].grad, self._params[idx[-1]].grad))

    def _pump(self):
        """Issue every ready branch whose predecessors (in the fixed order) have been issued."""
        for k in self._order:
            if k in self._issued:
                continue
            if k not in self._ready:
                break
            self._issue(k)
            self.early_hits += 1

    def _trigger(self, k):
        # a trigger that fires for a branch whose message has left (backward of a second loss in the same step, or a new
        # backward pass after a step that was skipped): the branch is reduced again in reduce_gradients, whatever the
        # allocator made of the gradients' identities
        if k in self._issued:
            self._dirty.add(k)
            return
        if any(self._params[i].grad is None for i in self._members[k]):
            return
        self._ready.add(k)
        g = self._params[self._members[k][0]].grad
        if g.is_cuda:
            ev = torch.cuda.Event()
            ev.record()         # on the stream this branch's backward runs on (the hook's current stream)
            self._events[k] = ev
        self._pump()

    def _agree(self):
        """Calibration, once per ``reduce_gradients`` until armed -- on EVERY rank, whatever its own backward produced: the
        ranks agree (a) that each of them has seen a backward pass producing every branch's full gradient set, and (b) on
        ONE issue order, rank 0's arrival order.  Until (a) holds everywhere nothing is armed and this step's collectives go
        out in the static module order.  One small all-reduce (MIN of the completeness flags) and, on the step that arms,
        one broadcast of ``len(branches)`` integers."""
        self.calibration_steps += 1
        pos = {i: n for n, i in enumerate(self._arrival)}
        last = {}
        for k in self.branches:
            seen = [i for i in self._members[k] if i in pos]
            if len(seen) == len(self._members[k]):
                last[k] = max(seen, key=lambda i: pos[i])
        dev = self._params[0].device if self.backend == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if len(last) == len(self.branches) else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            self._arrival.clear()        # not every rank is ready: calibrate again on the next backward pass
            return
        mine = sorted(range(len(self.branches)), key=lambda b: pos[last[self.branches[b]]])
        order = torch.tensor(mine, dtype=torch.int64, device=dev)
        dist.broadcast(order, src=0)
        self._arm([self.branches[b] for b in order.tolist()], last)

    def _arm(self, order, last):
        """One trigger per branch, on the parameter whose gradient arrived last on THIS rank; the issue order is the agreed one."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._order = list(order)
        self.triggers = {k: self._names[i] for k, i in last.items()}
        for k, i in last.items():
            self._hooks.append(self._params[i].register_post_accumulate_grad_hook(lambda _p, k=k: self._trigger(k)))
        self._arrival = None

    @torch.no_grad()
    def reduce_gradients(self):
        """Average the gradients over the ranks (call once after backward; ``make_optimizer`` does it before every step):
        waits for the branch messages issued during backward and reduces whatever is left, in the fixed order."""
        if self.overlap and self._order is None:
            self._agree()               # every rank, every step until armed (a collective: not conditional on local state)
            armed_now = self._order is not None
        else:
            armed_now = False
        # the step that arms still goes out in the static order (nothing was issued from its backward)
        order = self.branches if (self._order is None or armed_now) else self._order
        for k in order:
            stale = False
            if k in self._issued:       # issued from backward: still the gradients it copied?
                stale = k in self._dirty or self._signature(k) != self._sig[k]
                if stale:
                    self._work.pop(k)[0].wait()
                    self._issued.remove(k)
            if k not in self._issued:
                self._issue(k)
        for k in order:
            work, seg, complete = self._work.pop(k)
            work.wait()
            if self.backend != "nccl":
                seg.div_(self.world)
            idx = self._members[k]
            if complete:
                for i in idx:
                    self._params[i].grad = self._views[i]
            else:       # rare: some parameter had no gradient here -- does any rank have one?  (one device -> host copy)
                present = self._present[k].cpu()
                for j, i in enumerate(idx):
                    if float(present[j]) > 0.0:
                        self._params[i].grad = self._views[i]
        self._ready, self._issued, self._sig = set(), [], {}
        self._dirty, self._events = set(), {}


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
