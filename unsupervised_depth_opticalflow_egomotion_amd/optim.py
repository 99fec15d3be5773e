"""The optimiser of the reference's training loop -- ``torch.optim.Adam(model.parameters(), lr)`` (train.py:85-87) -- as
one HIP launch per step (csrc/ops_adam.hip).

``FusedAdam`` keeps ``torch.optim.Adam``'s hyper-parameters, update rule and state layout (``step``, ``exp_avg``,
``exp_avg_sq`` per parameter; the same ``param_groups`` keys), so a checkpoint written by one loads into the other
(train.py:90-94 resumes from ``optimizer.state_dict()``).  What changes is how the update reaches the GPU: ATen's fused
multi-tensor kernel takes its pointer tables as kernel arguments -- six launches of ~48 us for this model's ~250 tensors,
on the serial tail of the step -- here the tables are device tensors (the gradient pointers are refreshed each step, the
rest is built once) and one launch updates everything.

fp32 parameters on one HIP device, dense gradients, no weight decay / amsgrad / maximize (the reference uses none);
anything else raises."""
import ctypes

import torch

from ._lib import check, get_lib, stream_ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """``capturable``: the step count lives on the device (dfe_adam_step_dev), so that a whole training step -- this
        ``step()`` included -- can be captured in a hipGraph and replayed (train_step.GraphedTrainStep, train.py --graph);
        ``state_dict()`` reads the count back.  All parameters of a group must then step together."""
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=bool(capturable), differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._plans = {}
        self._dev_count = {}          # capturable groups: group index -> (step count [1] float64, coefficients [2] float32) on the device
        # what a captured step (hipGraph) reads by ADDRESS: plans (table, block map, its own pinned pointer table) and the
        # device counts.  Held for the optimiser's lifetime, so that a replay can never read freed memory; ``capture_epoch``
        # moves whenever they stop being the live ones (load_state_dict) and GraphedTrainStep refuses to replay across that.
        self._captured = []
        self.capture_epoch = 0

    MAX_PLANS = 8

    def state_dict(self):
        """As torch.optim.Adam's.  Inside this optimiser the parameters of a group share ONE step-count tensor (one
        increment per step instead of ~250); a checkpoint must not carry that sharing -- torch.optim.Adam increments
        every parameter's count, i.e. a shared one once per parameter -- so every entry gets its own copy here."""
        for gi, (count, _) in self._dev_count.items():      # capturable: the count advanced on the device (graph replays)
            t = float(count.cpu()[0])
            for p in self.param_groups[gi]["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    st["step"] = torch.tensor(t, dtype=torch.float32)
        sd = super().state_dict()
        sd["state"] = {k: ({**v, "step": v["step"].clone()} if "step" in v else v) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}                       # the moment tensors were replaced
        self._dev_count = {}                   # re-seeded from the loaded counts at the next step
        if self._captured:
            self.capture_epoch += 1            # a graph captured before this point updates the OLD moments: re-capture

    def _fingerprint(self, params):
        """Cheap check that the cached parameter / moment pointers are still the live ones."""
        a, b = params[0], params[-1]
        return (a.data_ptr(), b.data_ptr(), self.state[a]["exp_avg"].data_ptr(), self.state[b]["exp_avg_sq"].data_ptr())

    # ------------------------------------------------------------------ tables
    def _plan(self, gi, params):
        """Static part of the launch for one param group and one set of parameters that have gradients."""
        key = (gi, tuple(id(p) for p in params))
        plan = self._plans.get(key)
        if plan is not None and plan["fingerprint"] == self._fingerprint(params):
            return plan
        lib = get_lib()
        chunk = lib.dfe_adam_chunk()
        dev = params[0].device
        rows, blocks = [], []
        for t, p in enumerate(params):
            st = self.state[p]
            rows.append([p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()])
            blocks += [[t, c] for c in range((p.numel() + chunk - 1) // chunk)]
        # the pointer table reaches the device through a small ring of pinned buffers (an un-pinned source would make the
        # copy synchronise the stream, i.e. stall the host once per step); a buffer is reused only after its copy ran
        plan = {"host": torch.tensor(rows, dtype=torch.int64),
                "pinned": [torch.empty(len(rows), 5, dtype=torch.int64).pin_memory() for _ in range(4)],
                # the captured step's own source buffer (see step()): allocated here, pinning memory is not capturable
                "graph_pinned": torch.empty(len(rows), 5, dtype=torch.int64).pin_memory(),
                "events": [None] * 4, "turn": 0,
                "table": torch.empty(len(rows), 5, dtype=torch.int64, device=dev),
                "blockmap": torch.tensor(blocks, dtype=torch.int32).to(dev), "nblocks": len(blocks),
                "fingerprint": self._fingerprint(params)}
        # a group normally has ONE live plan; parameters that got their first gradient at different steps need one per
        # distinct count.  Plans are keyed by (group, parameter ids) only -- never by the count, which changes every
        # step -- and a group keeps at most MAX_PLANS of them (oldest dropped: pinned buffers, table, block map).
        mine = [k for k in self._plans if k[0] == gi]
        for k in mine[:max(0, len(mine) - (self.MAX_PLANS - 1))]:
            del self._plans[k]
        self._plans[key] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = get_lib()
        for gi, group in enumerate(self.param_groups):
            if group.get("weight_decay", 0) or group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdam implements the reference's plain Adam (no weight decay / amsgrad / maximize)")
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            dev = params[0].device
            grads = []
            for p in params:
                g = p.grad
                if (g.is_sparse or g.dtype != torch.float32 or p.dtype != torch.float32 or not p.is_cuda or p.device != dev
                        or not p.is_contiguous()):
                    raise NotImplementedError("FusedAdam needs dense fp32 parameters and gradients on one HIP device")
                grads.append(g if g.is_contiguous() else g.contiguous())
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # parameters of a group that stepped together share their count: one launch per distinct count
            by_step = {}
            for p, g in zip(params, grads):
                by_step.setdefault(float(self.state[p]["step"]), []).append((p, g))
            for t0, pg in by_step.items():
                ps = [p for p, _ in pg]
                plan = self._plan(gi, ps)
                host = plan["host"]
                host[:, 1] = torch.tensor([g.data_ptr() for _, g in pg], dtype=torch.int64)
                capturing = torch.cuda.is_current_stream_capturing()
                if capturing:
                    # the graph's copy node re-reads its source on every replay: a pinned buffer of its OWN, outside the ring
                    # the eager steps cycle through (an eager step after the capture would otherwise leave its gradient
                    # pointers where the next replay picks them up)
                    plan["graph_pinned"].copy_(host)
                    plan["table"].copy_(plan["graph_pinned"], non_blocking=True)
                    self._captured.append(plan)
                else:
                    k = plan["turn"]
                    plan["turn"] = (k + 1) % len(plan["pinned"])
                    if plan["events"][k] is not None:
                        plan["events"][k].synchronize()
                    plan["pinned"][k].copy_(host)
                    plan["table"].copy_(plan["pinned"][k], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    plan["events"][k] = ev
                t = t0 + 1.0
                b1, b2 = group["betas"]
                if group.get("capturable"):
                    if len(by_step) != 1:
                        raise NotImplementedError("FusedAdam(capturable=True): all parameters of a group must step together")
                    dc = self._dev_count.get(gi)
                    if dc is None:      # seeded from the host count once; from then on the device count is the truth
                        dc = self._dev_count[gi] = (torch.full((1,), t0, dtype=torch.float64, device=dev),
                                                    torch.zeros(2, dtype=torch.float32, device=dev))
                    if capturing:
                        self._captured.append(dc)
                    check(lib.dfe_adam_step_dev(ctypes.c_void_p(plan["table"].data_ptr()), ctypes.c_void_p(plan["blockmap"].data_ptr()),
                                                plan["nblocks"], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                ctypes.c_void_p(dc[0].data_ptr()), ctypes.c_void_p(dc[1].data_ptr()), stream_ptr()),
                          "dfe_adam_step_dev")
                else:
                    check(lib.dfe_adam_step(ctypes.c_void_p(plan["table"].data_ptr()), ctypes.c_void_p(plan["blockmap"].data_ptr()),
                                            plan["nblocks"], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                            1.0 - b1 ** t, 1.0 - b2 ** t, stream_ptr()), "dfe_adam_step")
                first = self.state[ps[0]]["step"]
                shared = all(self.state[p]["step"] is first for p in ps)
                if shared:
                    # torch.optim.Adam counts per parameter: one that holds the shared tensor but sits this launch out
                    # (no gradient this step) keeps its count on a copy of its own
                    if len(ps) != len(group["params"]):
                        inside = {id(p) for p in ps}
                        for q in group["params"]:
                            sq = self.state.get(q)
                            if sq and sq.get("step") is first and id(q) not in inside:
                                sq["step"] = first.clone()
                    first += 1.0                    # one CPU tensor shared by the launch's parameters
                else:
                    new = torch.tensor(t, dtype=torch.float32)
                    for p in ps:
                        self.state[p]["step"] = new
        # the parameters changed behind autograd's back (raw pointers, no version bump): the Winograd kernel's cached
        # transformed filters are rebuilt here, in one launch on this stream, right behind the update
        from . import ops
        ops.wino_weights.refresh()
        return loss
