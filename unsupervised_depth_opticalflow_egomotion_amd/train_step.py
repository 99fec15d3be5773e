"""One optimiser step of the joint model (train.py:171-216): forward, weighted sum of the loss_pack means,
backward, Adam.  Used by train.py and by bench.py's ``train_step`` workload."""
import types

import numpy as np
import torch

from . import ddp, synthetic
from .models import get_model

DEFAULT_CFG = dict(
    dataset="kitti_depth", num_scales=3, num_input_frames=3, flow_consist_alpha=0.01, flow_consist_beta=0.5,
    h_flow_consist_alpha=0.01, h_flow_consist_beta=0.5, geometric_ratio=0.3, geometric_num=6000, pose_beta=1,
    w_flow_pixel=0.15, w_flow_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, w_depth_pixel=1.0, w_depth_ssim=0.85,
    w_depth_smooth=0.5, w_depth_consis=0.1, w_depth_flow_consis=1.0, w_epipolar=0.1, w_triangle=0.001, w_pnp=0.1,
    w_8point=0.1, img_hw=(256, 832), lr=1e-4, mode="geom")

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


def total_loss(loss_pack, cfg):
    """train.py:211-214: sum_k w_k * mean(loss_k)."""
    return sum(getattr(cfg, LOSS_WEIGHT_ATTR[k]) * v.mean() for k, v in loss_pack.items())


def train_step(model, optimizer, inputs, cfg):
    optimizer.zero_grad(set_to_none=True)
    loss_pack, mask_pack = model(inputs)
    loss = total_loss(loss_pack, cfg)
    loss.backward()
    optimizer.step()
    return loss, loss_pack, mask_pack


class TrainStepWorkload:
    """bench.py workload: mode=geom on synthetic KITTI-shaped triplets (configs[2] / configs[3])."""
    name = "train_step"

    def __init__(self, args, dev, seed, world=1):
        self.args, self.dev = args, dev
        self.cfg = make_cfg(num_scales=args.scales, img_hw=(args.height, args.width))
        torch.manual_seed(1234)           # identical initial weights on every rank
        self.model = get_model("geom")(self.cfg).to(dev)
        import os
        if os.environ.get("DFE_CHANNELS_LAST", "0") == "1":
            self.model.use_channels_last(True)
        self.model.train()
        self.model = ddp.wrap(self.model, dev)
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.opt = torch.optim.Adam(params, lr=self.cfg.lr)
        im, k, ki = synthetic.make_triplet_batch(args.batch, args.height, args.width, args.scales, seed=seed)
        self.np_inputs = (im, k, ki)
        self.inputs = [torch.from_numpy(a).to(dev) for a in (im, k, ki)]   # resident in HBM before timing
        self._ls = None

    def step(self):
        return train_step(self.model, self.opt, self.inputs, self.cfg)[0]

    def loss_stack_workload(self):
        """Loss-stack-only view on the same shapes, for the per-kernel roofline measurement."""
        if self._ls is None:
            import bench
            self._ls = bench.LossStackWorkload(self.args, self.dev, seed=1234)
        return self._ls

    def cpu_step_fn(self, threads):
        """CPU baseline: the same networks on the host + the oracle's loss stack + Adam."""
        from oracle import loss_stack_oracle as O
        from .networks import pwc_tf

        class OraclePWC(pwc_tf.PWC_tf):
            def warp(self, x, flow):
                return O.warp_flow(x, flow, use_mask=False)

            def corr_naive(self, a, b, d=4):
                return O.corr_naive(a, b, d)

        cfg = self.cfg
        torch.manual_seed(1234)
        model = get_model("geom")(cfg)
        pw = OraclePWC()
        pw.load_state_dict(model.pwc_model.state_dict())
        pw.corr = pw.corr_naive
        model.pwc_model = pw
        model.train()
        ddp.freeze_unused(model)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.lr)
        oracle = O.GeomLossOracle(num_scales=cfg.num_scales)
        images, k_ms, ki_ms = [torch.from_numpy(a) for a in self.np_inputs]
        h = images.shape[2] // 3

        def run():
            opt.zero_grad(set_to_none=True)
            img_l, img, img_r = images[:, :, :h], images[:, :, h:2 * h], images[:, :, 2 * h:]
            dl, dt, dr, pose, fb, ff = model.run_networks(img_l, img, img_r)
            lp, _ = oracle.geom_losses(img_l, img, img_r, dl, dt, dr, pose, fb, ff, k_ms[:, 0], ki_ms[:, 0])
            total_loss(lp, cfg).backward()
            opt.step()
        return run
