"""One optimiser step of the joint model (train.py:171-216): forward, weighted sum of the loss_pack means,
backward, Adam.  Used by train.py and by bench.py's ``train_step`` workload (the workload class itself, with its
CPU-baseline leg that drives the oracle, lives in bench.py -- nothing in this package imports the oracle)."""
import types

DEFAULT_CFG = dict(
    dataset="kitti_depth", num_scales=3, num_input_frames=3, flow_consist_alpha=0.01, flow_consist_beta=0.5,
    h_flow_consist_alpha=0.01, h_flow_consist_beta=0.5, geometric_ratio=0.3, geometric_num=6000, pose_beta=1,
    w_flow_pixel=0.15, w_flow_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, w_depth_pixel=1.0, w_depth_ssim=0.85,
    w_depth_smooth=0.5, w_depth_consis=0.1, w_depth_flow_consis=1.0, w_epipolar=0.1, w_triangle=0.001, w_pnp=0.1,
    w_8point=0.1, img_hw=(256, 832), lr=1e-4, mode="geom", enable_depth_ssim=False, enable_depth_consis=False)

LOSS_WEIGHT_ATTR = {
    "loss_flow_pixel": "w_flow_pixel", "loss_flow_ssim": "w_flow_ssim", "loss_flow_smooth": "w_flow_smooth",
    "loss_flow_consis": "w_flow_consis", "loss_depth_pixel": "w_depth_pixel", "loss_depth_ssim": "w_depth_ssim",
    "loss_depth_smooth": "w_depth_smooth", "loss_depth_consis": "w_depth_consis",
    "loss_depth_flow_consis": "w_depth_flow_consis", "loss_epipolar": "w_epipolar", "loss_triangle": "w_triangle",
    "loss_pnp": "w_pnp", "loss_eight_point": "w_8point"}


def make_cfg(**over):
    d = dict(DEFAULT_CFG)
    d.update(over)
    return types.SimpleNamespace(**d)


def make_optimizer(model, lr, capturable=False):
    """The reference's ``torch.optim.Adam(model.parameters(), lr)`` (train.py:85-87), same hyper-parameters, state
    layout and update rule.  On a HIP device: ``optim.FusedAdam``, one launch for all parameters (ATen's fused
    multi-tensor Adam needs six for this model; ``DFE_FUSED_ADAM=0`` selects it).  ``capturable``: the step count on the
    device, for a step replayed from a hipGraph (GraphedTrainStep)."""
    import os
    import torch
    params = [p for p in model.parameters() if p.requires_grad]
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if on_gpu and os.environ.get("DFE_FUSED_ADAM", "1") != "0":
        from .optim import FusedAdam
        opt = FusedAdam(params, lr=lr, capturable=capturable)
    else:
        opt = torch.optim.Adam(params, lr=lr, fused=True, capturable=capturable) if on_gpu else torch.optim.Adam(params, lr=lr)
    if hasattr(model, "reduce_gradients"):      # ddp.FlatAllReduce: the gradient all-reduce runs in front of every step
        opt.register_step_pre_hook(lambda _opt, _args, _kwargs: model.reduce_gradients())
    return opt


_WEIGHT_ROWS = {}


def total_loss(loss_pack, cfg):
    """train.py:211-214: sum_k w_k * mean(loss_k).  The (2,)-shaped zero placeholders of the disabled terms
    (models._zeros2) contribute exactly +0.0 and are left out.  When the pack carries the fused stack's whole
    [rows, B] loss tensor (loss_stack.LossRows) the terms living in it are summed as one weighted reduction of that
    tensor -- sum_k (w_k / B) * sum_b L[k, b] -- instead of a mean, a multiply and an add per term (and their three
    backward kernels)."""
    import torch
    rows = getattr(loss_pack, "rows", None)
    total, fused = None, {}
    if rows is not None:
        losses, fused = rows
        key = (losses.device, losses.shape, tuple(sorted((i, float(getattr(cfg, LOSS_WEIGHT_ATTR[k]))) for k, i in fused.items())))
        w = _WEIGHT_ROWS.get(key)
        if w is None:
            host = torch.zeros(losses.shape[0])
            for k, i in fused.items():
                host[i] = float(getattr(cfg, LOSS_WEIGHT_ATTR[k])) / losses.shape[1]
            w = _WEIGHT_ROWS[key] = host.to(losses.device)
        total = torch.dot(losses.sum(1), w)
    for k, v in loss_pack.items():
        if k in fused or getattr(v, "_dfe_zero_placeholder", False):
            continue
        term = getattr(cfg, LOSS_WEIGHT_ATTR[k]) * v.mean()
        total = term if total is None else total + term
    return total


def train_step(model, optimizer, inputs, cfg, profiler=None):
    """``profiler``: a profiling.Profiler (train.py --profile): the reference's per-section wall times (synchronising marks)."""
    from . import profiling
    optimizer.zero_grad(set_to_none=True)
    with profiling.range("forward"):
        loss_pack, mask_pack = model(inputs)
        loss = total_loss(loss_pack, cfg)
    if profiler is not None:
        profiler.report_process("forward")
    with profiling.range("backward"):
        loss.backward()
    if profiler is not None:
        profiler.report_process("backward")
    with profiling.range("optimizer"):
        optimizer.step()
    if profiler is not None:
        profiler.report_process("optimizer")
    return loss, loss_pack, mask_pack


class GraphedTrainStep:
    """``train_step`` captured ONCE in a hipGraph and replayed (train.py --graph, bench.py --graph): the three network streams,
    the fused loss stack, backward and the optimiser step become one graph launch per iteration -- no per-kernel host work.
    Where it pays: whenever the host is the bound (the host needs ~16 ms to enqueue a step whatever the batch: B = 1 on one
    MI355X 17.9 ms eager -> 8.4 ms replayed; at B = 4 the GPU is the bound: 19.6 -> 19.4).
    Requirements: static shapes; an optimiser made with ``make_optimizer(..., capturable=True)`` (the step count on the device);
    a single process (a data-parallel reducer's collectives are not captured: use the eager step there).  ``warmup`` eager steps
    run first on a side stream (allocator, MIOpen's solver choice, the Winograd filter cache); with ``restore`` (default) the
    parameters, buffers and optimiser state they changed are put back IN PLACE afterwards, so constructing the object trains
    nothing: the first ``__call__`` is the first optimiser step on that batch and Adam's count stays equal to the caller's
    iteration count (checkpoints: train.py's ``iteration`` key).  ``__call__`` copies the new batch into the static input tensors
    and replays.  The returned loss / loss_pack / mask_pack are the graph's static outputs.
    The graph holds lr / betas / eps as kernel arguments and the optimiser's tables by address: ``__call__`` raises if a
    hyper-parameter changed or ``optimizer.load_state_dict`` ran since the capture (build a new GraphedTrainStep then)."""

    def __init__(self, model, optimizer, inputs, cfg, warmup=3, restore=True):
        import torch
        if hasattr(model, "reduce_gradients") or type(model).__name__ == "DistributedDataParallel":
            raise NotImplementedError("GraphedTrainStep: single-process training only (collectives are not captured)")
        self.model, self.optimizer, self.cfg = model, optimizer, cfg
        self.static_inputs = [t.clone() for t in inputs]
        saved = self._snapshot() if restore else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                train_step(model, optimizer, self.static_inputs, cfg)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if saved is not None:
            self._restore(saved)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = train_step(model, optimizer, self.static_inputs, cfg)
        # The captured filter refresh (ops.WinoWeightCache.refresh inside optimizer.step) reads the cache's device tables of THIS
        # moment; the cache replaces them when another model registers filters.  Hold them: a replay must never read freed memory
        # (found by the eager-twin test: the twin's registration freed the table, the next replay scattered filters through it).
        from . import ops
        self._cache_tables = (ops.wino_weights.table, ops.wino_weights.blockmap)
        self._hyper = self._hyper_now()
        self._epoch = getattr(optimizer, "capture_epoch", 0)

    def _hyper_now(self):
        return [(float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])) for g in self.optimizer.param_groups]

    def _snapshot(self):
        """Parameters, buffers and the optimiser's state before the warm-up steps (device copies; ~3x the model)."""
        opt = self.optimizer
        state = {}
        for group in opt.param_groups:
            for p in group["params"]:
                st = opt.state.get(p)
                if st:
                    state[p] = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in st.items()}
        dev_count = {gi: [t.clone() for t in dc] for gi, dc in getattr(opt, "_dev_count", {}).items()}
        return ([p.detach().clone() for p in self.model.parameters()], [b.detach().clone() for b in self.model.buffers()],
                state, dev_count)

    def _restore(self, saved):
        """Put the snapshot back IN PLACE (every address the capture is about to record stays what it is)."""
        import torch
        params, buffers, state, dev_count = saved
        opt = self.optimizer
        with torch.no_grad():
            for p, v in zip(self.model.parameters(), params):
                p.copy_(v)
            for b, v in zip(self.model.buffers(), buffers):
                b.copy_(v)
            for group in opt.param_groups:
                for p in group["params"]:
                    st = opt.state.get(p)
                    if not st:
                        continue
                    old = state.get(p)
                    for k, v in st.items():
                        if not hasattr(v, "copy_"):
                            continue
                        if old is not None and k in old:
                            v.copy_(old[k])
                        else:           # created by the warm-up: back to its initial value (zero moments, count 0)
                            v.zero_()
            for gi, dc in getattr(opt, "_dev_count", {}).items():
                if gi in dev_count:
                    for t, v in zip(dc, dev_count[gi]):
                        t.copy_(v)
                else:
                    dc[0].zero_()
        from . import ops
        ops.wino_weights.invalidate()       # parameters written in place without a step: cached transformed filters are stale
        ops.wino_weights.refresh()

    def __call__(self, inputs=None):
        if self._hyper_now() != self._hyper:
            raise RuntimeError("GraphedTrainStep: lr / betas / eps changed since the capture (they are kernel arguments of the "
                               "captured optimiser step): build a new GraphedTrainStep")
        if getattr(self.optimizer, "capture_epoch", 0) != self._epoch:
            raise RuntimeError("GraphedTrainStep: optimizer.load_state_dict() ran since the capture (the graph updates the "
                               "previous moment tensors): build a new GraphedTrainStep")
        if inputs is not None:
            for dst, src in zip(self.static_inputs, inputs):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs
