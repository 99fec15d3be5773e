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


def make_optimizer(model, lr):
    """The reference's ``torch.optim.Adam(model.parameters(), lr)`` (train.py:85-87), same hyper-parameters, state
    layout and update rule.  On a HIP device: ``optim.FusedAdam``, one launch for all parameters (ATen's fused
    multi-tensor Adam needs six for this model; ``DFE_FUSED_ADAM=0`` selects it)."""
    import os
    import torch
    params = [p for p in model.parameters() if p.requires_grad]
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if on_gpu and os.environ.get("DFE_FUSED_ADAM", "1") != "0":
        from .optim import FusedAdam
        opt = FusedAdam(params, lr=lr)
    else:
        opt = torch.optim.Adam(params, lr=lr, fused=True) if on_gpu else torch.optim.Adam(params, lr=lr)
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
