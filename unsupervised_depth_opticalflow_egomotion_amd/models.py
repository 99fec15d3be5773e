"""``Model_geometry`` / ``Model_depth`` / ``Model_flow`` and ``get_model`` with the reference's API surface
(core/networks/__init__.py:22-30, model_geometry.py, model_depth.py, model_flow.py).

forward(inputs) with inputs = [images [B,3,3H,W] (left/target/right stacked along H), K_ms [B,S,3,3],
K_inv_ms [B,S,3,3]] returns (loss_pack, mask_pack) (Model_flow returns them too -- the shipped
Model_flow.forward raises NameError and returns one value where train.py:172 unpacks two; fixed here).
Sub-module names and state-dict keys are the reference's.  The joint model's loss stack runs as the fused
HIP launches of loss_stack.py; the networks run on PyTorch-ROCm."""
import numpy as np
import torch
import torch.nn as nn

from .loss_stack import LossRows, geom_loss_stack, depth_loss_stack, flow_loss_stack, decode_mask
from .geometry_solvers import GeometrySolvers
from .loss_terms import LossTerms
from .networks import Depth_Model, PoseCNN, FeaturePyramid, PWC_tf

PLACEHOLDER_KEYS_GEOM = ("loss_depth_ssim", "loss_depth_consis", "loss_triangle", "loss_pnp", "loss_eight_point")
LOSS_ORDER_GEOM = ("loss_depth_pixel", "loss_depth_ssim", "loss_depth_smooth", "loss_depth_consis", "loss_flow_pixel",
                   "loss_flow_ssim", "loss_flow_smooth", "loss_flow_consis", "loss_depth_flow_consis", "loss_epipolar",
                   "loss_triangle", "loss_pnp", "loss_eight_point")


class LazyPack(dict):
    """dict whose values are produced on first access.  The reference builds ``mask_pack`` with nine
    ``.cpu().numpy()`` calls every iteration (model_geometry.py:871-880), a forced device sync per step;
    the keys and value types are kept, the device->host copies happen only when a value is read."""

    def __init__(self, thunks):
        super().__init__()
        self._thunks = dict(thunks)
        for k in self._thunks:
            dict.__setitem__(self, k, None)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None and k in self._thunks:
            v = self._thunks.pop(k)()
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


def _split_frames(images):
    assert images.shape[1] == 3
    h = int(images.shape[2] / 3)
    return images[:, :, :h, :], images[:, :, h:2 * h, :], images[:, :, 2 * h:3 * h, :], h, images.shape[3]


def _contiguous_frames(images):
    """The three frames of the [B,3,3h,W] strip (train.py:171-175) as contiguous tensors.  On the GPU one strided
    copy produces the [3,B,3,h,W] block whose slices are the frames -- and whose flat view is the depth net's batch
    of 3B (no further concatenation); otherwise three plain copies."""
    img_l, img, img_r, h, w = _split_frames(images)
    if images.is_cuda and images.shape[2] == 3 * h:
        B = images.shape[0]
        x = images.reshape(B, 3, 3, h, w).permute(2, 0, 1, 3, 4).contiguous()
        return x[0], x[1], x[2], x.view(3 * B, 3, h, w)
    return img_l.contiguous(), img.contiguous(), img_r.contiguous(), None


def _depth_frames(depth_net, img_l, img, img_r, batched=None):
    """Disparity pyramids of the three frames (model_geometry.py:786-788, model_depth.py:286-288: one depth_net call per
    frame).  On the GPU the frames go through the net as one batch with per-frame BatchNorm statistics
    (Depth_Model.forward_frames); on the host, the three calls as written."""
    if img.is_cuda:
        return depth_net.forward_frames([img_l, img, img_r], batched=batched)
    return depth_net(img_l), depth_net(img), depth_net(img_r)


def _flow_branches(fpyramid, pwc_model, img_l, img, img_r):
    """Both flow directions of a triplet (model_geometry.py:790-793, model_flow.py:214-222: three fpyramid calls and
    two pwc_model calls).  Neither net holds batch statistics, so one pass over the 3B frames and one over the 2B
    (target, source) pairs give the same per-sample results with a third / half of the launches and larger
    convolutions at the coarse levels.  Returns (flows target->left, flows target->right)."""
    B, h, w = img.shape[0], img.shape[2], img.shape[3]
    feats = fpyramid(torch.cat([img, img_l, img_r], 0))
    # split() instead of slices: its backward is one concatenation of the parts' gradients, where every slice would
    # zero-fill a full-size gradient, copy its part in and leave the sum to a chain of adds
    parts = [f.split([B, 2 * B]) for f in feats]              # target | (left, right)
    # the target's features for both directions; the finest level is never read by the decoder (pwc_tf.py:108-179
    # uses levels 2..6), so its 27 MB are not duplicated
    f1 = [t if k == 0 else torch.cat([t, t], 0) for k, (t, _) in enumerate(parts)]
    f2 = [lr for _, lr in parts]
    flows = [f.split(B) for f in pwc_model(f1, f2, [h, w])]
    return [f[0] for f in flows], [f[1] for f in flows]


_PLACEHOLDERS = {}
_SIDE_STREAMS = {}
_DEFAULT_NET_STREAMS = int(__import__("os").environ.get("DFE_NET_STREAMS", "3"))   # 1 = everything on the current stream


def _side_streams(dev):
    """(flow stream, pose stream) of a device, with their HIP priorities (DFE_STREAM_PRIORITIES = "flow,pose"; the depth net
    stays on the caller's stream at priority 0).  Both HIGH (-1,-1).  Round 3, when the flow branch was the step's long pole:
    25.73 ms against 26.09 for 0,0.  Round 4, after this build's Winograd kernel left the DEPTH branch the longer one (14.4
    against 12.7 ms of kernels), 0,-1 / 0,0 measure 0.18 ms better in a plain process (21.77 against 21.95) -- but a
    normal-priority side stream is only concurrent while it gets a hardware queue of its own: with a process group alive
    (RCCL's or gloo's streams present) the flow branch at priority 0 ran SERIALISED behind the depth branch, 27.0 ms =
    the one-stream time, against 22.2 with both side streams high.  High-priority streams have their own queues.
    Round 6 re-measured both (three alternating rounds on one box): plain process 18.79 ms at -1,-1 against 18.59 at 0,0 / 0,-1
    (the replayed hipGraph: 19.04 against 18.50) -- at equal priority the dispatcher deals workgroups to the depth and flow
    branches in turn instead of letting the flow branch's kernels cut in front of the longer depth branch --, under a world-size-1
    RCCL group 19.15 ms at -1,-1 against 23.3 at 0,0.  So the default now FOLLOWS THE PROCESS: 0,0 while no process group
    exists, -1,-1 once one does (the streams are re-made if a group appears later); DFE_STREAM_PRIORITIES overrides both.
    The serialisation under a process group is the HIP runtime running out of hardware queues (GPU_MAX_HW_QUEUES, default 4):
    with 8 the same run measures 19.04 ms at 0,0 against 19.24 at -1,-1 (profiles/r06_hw_queues.txt), so when the package
    raised the limit before HIP initialised (or the user exported >= 8; see the package's HW_QUEUES) 0,0 is used there too."""
    import os
    from . import HW_QUEUES
    pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    key = (dev.type, dev.index, pg)
    if key not in _SIDE_STREAMS:
        env = os.environ.get("DFE_STREAM_PRIORITIES")
        pf, pp = (int(v) for v in env.split(",")) if env else ((-1, -1) if pg and HW_QUEUES < 8 else (0, 0))
        _SIDE_STREAMS[key] = (torch.cuda.Stream(dev, priority=pf), torch.cuda.Stream(dev, priority=pp))
    return _SIDE_STREAMS[key]



def _zeros2(dev):
    """The reference's ``torch.zeros([2]).to(device).requires_grad_()`` placeholder of a disabled loss term
    (model_geometry.py:891,899,943-951).  Tagged so that train_step.total_loss can leave its exact-zero contribution
    (three tiny kernels forward, three backward per placeholder) out of the sum; one tensor per device is created once
    and handed out again (nothing ever writes to it or accumulates a gradient in it through total_loss)."""
    key = (dev.type, dev.index)
    t = _PLACEHOLDERS.get(key)
    if t is None:
        t = torch.zeros([2], device=dev).requires_grad_()   # device-side fill (no H2D copy: capturable in a hipGraph)
        t._dfe_zero_placeholder = True
        _PLACEHOLDERS[key] = t
    return t


class Model_geometry(LossTerms, GeometrySolvers, nn.Module):
    """Joint depth + pose + flow model ("monodepth2 + dynamic mask", model_geometry.py:15-953)."""

    def __init__(self, cfg):
        nn.Module.__init__(self)
        self.dataset = cfg.dataset
        self.num_scales = cfg.num_scales
        self.flow_consist_alpha = cfg.flow_consist_alpha
        self.flow_consist_beta = cfg.flow_consist_beta
        self.depth_net = Depth_Model(cfg.num_scales)
        self.pose_net = PoseCNN(cfg.num_input_frames)
        self.fpyramid = FeaturePyramid()
        self.pwc_model = PWC_tf()
        self.inlier_thres = 0.1
        self.rigid_thres = 0.5
        self.ratio = getattr(cfg, "geometric_ratio", 0.3)
        self.num = getattr(cfg, "geometric_num", 6000)
        self.beta = getattr(cfg, "pose_beta", 1)
        # the two depth terms the reference keeps commented (model_geometry.py:889-891,897-899); off = placeholders
        self.enable_depth_ssim = bool(getattr(cfg, "enable_depth_ssim", False))
        self.enable_depth_consis = bool(getattr(cfg, "enable_depth_consis", False))

    # ---- inference API (model_geometry.py:282-302)
    def infer_depth(self, img):
        return self.disp2depth(self.depth_net(img)[0])

    def inference_flow(self, img1, img2):
        hw = [img1.shape[2], img1.shape[3]]
        return self.pwc_model(self.fpyramid(img1), self.fpyramid(img2), hw)[0]

    def infer_pose(self, imgs):
        return self.pose_net(imgs)

    def run_networks(self, img_l, img, img_r, batched=None):
        """model_geometry.py:781-795.  depth_net is called once per frame (BatchNorm statistics per call).

        The three networks are independent until the loss stack.  On a HIP device (``net_streams`` > 1, the default) the
        flow branch and the pose net are enqueued on side streams while the depth net runs on the current one: the
        coarse pyramid levels launch grids far smaller than 256 CUs (512->512 at 8x26 x 12 images: 80 workgroups) and
        leave most of the chip idle when the nets run one after the other.  autograd replays every backward node on
        the stream of its forward op, so the backward passes overlap the same way.  Results are identical (same
        kernels, same order within a net).  Measured (MI355X, B=4): 32.4 -> 27.4 ms per step.  Measured and rejected in
        round 3: also taking the depth net off the current stream and giving every convolution's weight gradient a node
        and a stream of its own (two autograd nodes per convolution, so that no backward chain waits for a weight
        gradient): 28.8-29.7 ms -- ~90 more nodes and stream switches per step on the host, and the split
        data- / weight-gradient calls cost more than the joint ones."""
        n_streams = int(getattr(self, "net_streams", _DEFAULT_NET_STREAMS)) if img.is_cuda else 1
        if n_streams <= 1:
            disp_l, disp_t, disp_r = _depth_frames(self.depth_net, img_l, img, img_r, batched)
            pose = self.pose_net(torch.cat([img_l, img, img_r], 1))
            flows_bwd, flows_fwd = _flow_branches(self.fpyramid, self.pwc_model, img_l, img, img_r)
            return disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd
        main = torch.cuda.current_stream(img.device)
        s_flow, s_pose = _side_streams(img.device)
        s_flow.wait_stream(main)
        s_pose.wait_stream(main)
        for t in (img_l, img, img_r):
            t.record_stream(s_flow)
            t.record_stream(s_pose)
        with torch.cuda.stream(s_flow):
            flows_bwd, flows_fwd = _flow_branches(self.fpyramid, self.pwc_model, img_l, img, img_r)
        with torch.cuda.stream(s_pose):
            pose = self.pose_net(torch.cat([img_l, img, img_r], 1))
        disp_l, disp_t, disp_r = _depth_frames(self.depth_net, img_l, img, img_r, batched)
        main.wait_stream(s_flow)
        main.wait_stream(s_pose)
        for t in list(flows_bwd) + list(flows_fwd) + [pose]:
            t.record_stream(main)       # produced on a side stream, consumed by the loss stack on this one
        return disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd

    def forward(self, inputs):
        images, K_ms, K_inv_ms = inputs
        K, K_inv = K_ms[:, 0, :, :], K_inv_ms[:, 0, :, :]
        img_l, img, img_r, batched = _contiguous_frames(images)
        disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd = self.run_networks(img_l, img, img_r, batched)
        return self.loss_stack(img_l, img, img_r, disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd, K, K_inv)

    def disabled_depth_terms(self, img_l, img, img_r, disp_l, disp_t, disp_r, pose, K, mask_handle):
        """``loss_depth_ssim`` / ``loss_depth_consis`` exactly as the commented lines of the reference read
        (model_geometry.py:889-891,897-899), enabled by ``cfg.enable_depth_ssim`` /
        ``cfg.enable_depth_consis``.  They run on the per-operator HIP kernels (inverse_warp2 with projected / computed
        depth and its gradients wrt the target disparity, the SOURCE disparity and the pose; SSIM forward / backward;
        resize) under autograd; the texture-gated masks are decoded from the fused stack's mask pack, so the mask
        decisions are the fused stack's own.  ``loss_stack`` computes the same two terms inside the fused launches
        (``dfe_geom_args.depth_terms``); this per-operator composition is kept as its cross-check
        (tests/test_hip_models.py::test_disabled_depth_terms_vs_reference)."""
        S = self.num_scales
        img_list = self.generate_img_pyramid(img, S)
        rec_l, _, pd_l, cd_l = self.reconstruction(img_l, K, disp_t, disp_l, pose[:, 0].contiguous())
        rec_r, _, pd_r, cd_r = self.reconstruction(img_r, K, disp_t, disp_r, pose[:, 1].contiguous())
        with torch.no_grad():
            def gated(d):
                return [decode_mask(mask_handle, "valid_" + d, s) * decode_mask(mask_handle, "occ_" + d, s) *
                        decode_mask(mask_handle, "dyna_" + d, s) * decode_mask(mask_handle, "texture_" + d, s)
                        for s in range(S)]
            bwd_tex, fwd_tex = gated("bwd"), gated("fwd")
        out = {}
        if self.enable_depth_ssim:
            out["loss_depth_ssim"] = self.compute_ssim_loss(img_list, rec_l, bwd_tex) + \
                self.compute_ssim_loss(img_list, rec_r, fwd_tex)
        if self.enable_depth_consis:
            out["loss_depth_consis"] = self.compute_consis_loss(pd_l, cd_l, bwd_tex) + \
                self.compute_consis_loss(pd_r, cd_r, fwd_tex)
        return out

    def loss_stack(self, img_l, img, img_r, disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd, K, K_inv):
        """Everything from model_geometry.py:797 to :951 -> (loss_pack, mask_pack)."""
        S = self.num_scales
        active, masks = geom_loss_stack(img_l, img, img_r, disp_l, disp_t, disp_r, pose, flows_bwd, flows_fwd,
                                        K.contiguous(), K_inv.contiguous(), num_scales=S,
                                        flow_consist_alpha=self.flow_consist_alpha,
                                        flow_consist_beta=self.flow_consist_beta, return_masks="lazy",
                                        enable_depth_ssim=self.enable_depth_ssim,
                                        enable_depth_consis=self.enable_depth_consis)
        dev = img.device
        loss_pack = LossRows({k: (active[k] if k in active else _zeros2(dev)) for k in LOSS_ORDER_GEOM})
        loss_pack.rows = active.rows

        def u8(*names):
            """sample 0, scale 0 of the product of the named masks, decoded from the 1-byte mask pack only when the
            entry is read (the reference materialises all of them, with a device sync, every iteration)."""
            def run():
                t = decode_mask(masks, names[0])
                for n in names[1:]:
                    t = t * decode_mask(masks, n)
                return 255 * t[0].cpu().numpy().astype(np.uint8)
            return run

        def epi_masks(which):
            def run():
                with torch.no_grad():
                    dist = self.compute_epipolar_map(pose[:, 1].detach(), flows_fwd[0].detach(), K, K_inv)
                    rigid, inlier, _ = self.get_rigid_mask(dist)
                return 255 * (rigid if which == 0 else inlier)[0].cpu().numpy().astype(np.uint8)
            return run

        def valid_to_r():
            with torch.no_grad():
                from .structures import inverse_warp2
                v = inverse_warp2(img_r, disp_t[0].detach(), disp_r[0].detach(), pose[:, 1].detach().contiguous(), K)[1]
            return 255 * v[0].cpu().numpy().astype(np.uint8)

        mask_pack = LazyPack({
            "occ_fwd_mask": u8("occ_fwd"),
            "rigid_fwd_mask": epi_masks(0),
            "inlier_fwd_mask": epi_masks(1),
            "dyna_fwd_mask": u8("dyna_fwd"),
            "valid_fwd_mask": valid_to_r,
            "fwd_mask": u8("valid_fwd", "occ_fwd", "dyna_fwd"),
            "texture_mask_fwd": u8("texture_fwd"),
            "pred_depth_img": lambda: disp_t[0][0],
            "pred_flow_img": lambda: flows_fwd[0][0].detach().cpu().numpy().transpose([1, 2, 0]),
            "origin_middle_image": lambda: img[0].cpu().detach().numpy(),
        })
        return loss_pack, mask_pack


class Model_depth(LossTerms, nn.Module):
    """Depth + pose only (model_depth.py:272-337): pixel + smoothness terms, fused HIP launches (mode 1)."""

    def __init__(self, cfg):
        nn.Module.__init__(self)
        self.dataset = cfg.dataset
        self.num_scales = cfg.num_scales
        self.depth_net = Depth_Model(cfg.num_scales)
        self.pose_net = PoseCNN(cfg.num_input_frames)
        # the two depth terms the reference keeps commented (model_depth.py:326-327,332-333); off = placeholders
        self.enable_depth_ssim = bool(getattr(cfg, "enable_depth_ssim", False))
        self.enable_depth_consis = bool(getattr(cfg, "enable_depth_consis", False))

    def infer_depth(self, img):
        return self.disp2depth(self.depth_net(img)[0])

    def infer_pose(self, imgs):
        return self.pose_net(imgs)

    def fusion_mask(self, valid_mask, texture_mask):
        return [valid_mask[s] * texture_mask[s] for s in range(self.num_scales)]

    def forward(self, inputs):
        images, K_ms, K_inv_ms = inputs
        K = K_ms[:, 0, :, :]
        img_l, img, img_r, batched = _contiguous_frames(images)
        if img.is_cuda and int(getattr(self, "net_streams", _DEFAULT_NET_STREAMS)) > 1:
            # the pose net beside the depth net (see Model_geometry.run_networks)
            main = torch.cuda.current_stream(img.device)
            _, s_pose = _side_streams(img.device)
            s_pose.wait_stream(main)
            for t in (img_l, img, img_r):
                t.record_stream(s_pose)
            with torch.cuda.stream(s_pose):
                pose = self.pose_net(torch.cat([img_l, img, img_r], 1))
            depth_l, depth_t, depth_r = _depth_frames(self.depth_net, img_l, img, img_r, batched)
            main.wait_stream(s_pose)
            pose.record_stream(main)
        else:
            depth_l, depth_t, depth_r = _depth_frames(self.depth_net, img_l, img, img_r, batched)
            pose = self.pose_net(torch.cat([img_l, img, img_r], 1))
        return self.loss_stack(img_l, img, img_r, depth_l, depth_t, depth_r, pose, K)

    def loss_stack(self, img_l, img, img_r, depth_l, depth_t, depth_r, pose, K):
        """model_depth.py:296-335 in the fused HIP launches (mode 1 of dfe_geom_loss_fwd/bwd)."""
        active = depth_loss_stack(img_l, img, img_r, depth_l, depth_t, depth_r, pose, K.contiguous(),
                                  num_scales=self.num_scales,
                                  enable_depth_ssim=getattr(self, "enable_depth_ssim", False),
                                  enable_depth_consis=getattr(self, "enable_depth_consis", False))
        dev = img.device
        loss_pack = LossRows({"loss_depth_pixel": active["loss_depth_pixel"],
                              "loss_depth_ssim": active["loss_depth_ssim"] if "loss_depth_ssim" in active else _zeros2(dev),
                              "loss_depth_smooth": active["loss_depth_smooth"],
                              "loss_depth_consis": active["loss_depth_consis"] if "loss_depth_consis" in active else _zeros2(dev)})
        loss_pack.rows = active.rows
        return loss_pack, {}

    def loss_stack_per_op(self, img_l, img, img_r, depth_l, depth_t, depth_r, pose, K):
        """The same terms through the per-operator kernels and the per-method API (kept for cross-checking)."""
        S = self.num_scales
        pyr_t, pyr_l, pyr_r = (self.generate_img_pyramid(x, S) for x in (img, img_l, img_r))
        rec_l, valid_l, _, _ = self.reconstruction(img_l, K, depth_t, depth_l, pose[:, 0, :].contiguous())
        rec_r, valid_r, _, _ = self.reconstruction(img_r, K, depth_t, depth_r, pose[:, 1, :].contiguous())
        tex_b = self.compute_texture_mask(pyr_t, rec_l, pyr_l)
        tex_f = self.compute_texture_mask(pyr_t, rec_r, pyr_r)
        m_b, m_f = self.fusion_mask(valid_l, tex_b), self.fusion_mask(valid_r, tex_f)
        dev = img.device
        loss_pack = {
            "loss_depth_pixel": self.compute_photometric_loss(pyr_t, rec_l, m_b) + self.compute_photometric_loss(pyr_t, rec_r, m_f),
            "loss_depth_ssim": _zeros2(dev),
            "loss_depth_smooth": self.compute_smooth_loss(img, depth_t) + self.compute_smooth_loss(img_l, depth_l)
            + self.compute_smooth_loss(img_r, depth_r),
            "loss_depth_consis": _zeros2(dev),
        }
        return loss_pack, {}


class Model_flow(LossTerms, nn.Module):
    """Flow only (model_flow.py:14-261): soft occlusion weights, box-mean pyramid, fused HIP launches (mode 2)."""

    def __init__(self, cfg):
        nn.Module.__init__(self)
        self.fpyramid = FeaturePyramid()
        self.pwc_model = PWC_tf()
        if getattr(cfg, "mode", "flow") in ("depth", "flowposenet"):
            for p in list(self.fpyramid.parameters()) + list(self.pwc_model.parameters()):
                p.requires_grad = False
        self.dataset = cfg.dataset
        self.num_scales = cfg.num_scales
        # the shipped class reads cfg.h_flow_consist_* which the YAML does not define (model_flow.py:29-30)
        self.flow_consist_alpha = getattr(cfg, "h_flow_consist_alpha", getattr(cfg, "flow_consist_alpha", 0.01))
        self.flow_consist_beta = getattr(cfg, "h_flow_consist_beta", getattr(cfg, "flow_consist_beta", 0.5))

    def inference_flow(self, img1, img2):
        hw = [img1.shape[2], img1.shape[3]]
        return self.pwc_model(self.fpyramid(img1), self.fpyramid(img2), hw)[0]

    def get_occlusion_mask_from_flow(self, tensor_size, flow):
        """model_flow.py:33-39 -- dead in the reference (it calls an undefined ``transformerFwd``); here the bilinear
        forward splat of a ones image by ``flow``, clamped to [0,1], broadcast to ``tensor_size`` [B,C,H,W]."""
        from . import ops
        b, c, h, w = tensor_size
        if tuple(flow.shape) != (b, 2, h, w):
            raise ValueError("flow must be [B,2,H,W] matching tensor_size")
        return ops.forward_splat_ones(flow, clamp=True).expand(b, c, h, w)

    def forward(self, inputs):
        images = inputs[0]
        img_l, img, img_r, _ = _contiguous_frames(images)
        flows_bwd, flows_fwd = _flow_branches(self.fpyramid, self.pwc_model, img_l, img, img_r)
        return self.loss_stack(img_l, img, img_r, flows_bwd, flows_fwd)

    def loss_stack(self, img_l, img, img_r, flows_bwd, flows_fwd):
        """model_flow.py:224-255 in the fused HIP launches (mode 2 of dfe_geom_loss_fwd/bwd)."""
        return flow_loss_stack(img_l, img, img_r, flows_bwd, flows_fwd, num_scales=self.num_scales), {}

    def loss_stack_per_op(self, img_l, img, img_r, flows_bwd, flows_fwd):
        """The same terms through the per-operator kernels and the per-method API (kept for cross-checking)."""
        n = len(flows_fwd)
        pl, pt, pr = (self.generate_img_pyramid_avgpool(x, n) for x in (img_l, img, img_r))
        warp_l, warp_r = self.warp_flow_pyramid(pl, flows_bwd), self.warp_flow_pyramid(pr, flows_fwd)
        d_b, d_f, w_b, w_f = self.compute_diff_weight(warp_l, pt, warp_r)
        loss_pack = {
            "loss_flow_pixel": self.compute_loss_with_mask(d_f, w_f) + self.compute_loss_with_mask(d_b, w_b),
            "loss_flow_ssim": self.compute_loss_ssim(pt, warp_r, w_f) + self.compute_loss_ssim(pt, warp_l, w_b),
            "loss_flow_smooth": self.compute_loss_flow_smooth(flows_fwd, pt) + self.compute_loss_flow_smooth(flows_bwd, pt),
            "loss_flow_consis": self.compute_loss_flow_consis(flows_fwd, flows_bwd, w_f),
        }
        return loss_pack, {}


def get_model(mode):
    if mode == "flow":
        return Model_flow
    elif mode == "depth":
        return Model_depth
    elif mode == "geom":
        return Model_geometry
    raise ValueError("Mode {} not found.".format(mode))
