"""Host side of the fused loss stack: dfe_geom_loss_fwd / dfe_geom_loss_bwd (include/dfe_hip.h).

``geom_loss_stack`` computes the eight active ``loss_pack`` vectors of the reference's
``Model_geometry.forward`` (model_geometry.py:797-951) from the nets' outputs in a handful of
HIP launches, and is differentiable wrt the disparities, the flows and the pose.  PyTorch is
only the allocator, the stream and the autograd graph."""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import DfeError, check, f32c, get_lib, stream_ptr

MAX_SCALES = 8
LOSS_ROWS = ("loss_depth_pixel", "loss_depth_smooth", "loss_flow_pixel", "loss_flow_ssim", "loss_flow_smooth",
             "loss_flow_consis", "loss_depth_flow_consis", "loss_epipolar", "loss_depth_ssim", "loss_depth_consis")
DEPTH_TERM_SSIM, DEPTH_TERM_CONSIS = 1, 2     # dfe_geom_args.depth_terms bits (rows 8 / 9 of the loss buffer)
MASK_BITS = dict(valid_bwd=0x01, valid_fwd=0x02, occ_bwd=0x04, occ_fwd=0x08, dyna_bwd=0x10, dyna_fwd=0x20,
                 texture_bwd=0x40, texture_fwd=0x80)

_FP = ctypes.c_void_p


class LossRows(dict):
    """dict of per-term (B,) loss vectors that also carries the fused stack's whole [rows, B] loss tensor
    (``rows`` = (tensor, {term name: row index})), so that the weighted total can be formed from that one tensor
    (train_step.total_loss) instead of a mean, a multiply and an add per term."""
    rows = None


class GeomArgs(ctypes.Structure):
    """Mirror of ``dfe_geom_args`` (include/dfe_hip.h)."""
    _fields_ = [
        ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("num_scales", ctypes.c_int),
        ("align_corners", ctypes.c_int), ("mode", ctypes.c_int),
        ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
        ("img", _FP * 3),
        ("disp", (_FP * MAX_SCALES) * 3),
        ("flow", (_FP * MAX_SCALES) * 2),
        ("pose", _FP), ("K", _FP), ("K_inv", _FP),
        ("workspace", _FP), ("workspace_floats", ctypes.c_long),
        ("losses", _FP), ("grad_losses", _FP),
        ("grad_disp", (_FP * MAX_SCALES) * 3),
        ("grad_flow", (_FP * MAX_SCALES) * 2),
        ("grad_pose", _FP),
        ("depth_terms", ctypes.c_int),
    ]


def _dp(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise DfeError("loss stack tensors must be contiguous fp32 HIP tensors (no CPU fallback)")
    return t.data_ptr()


def _scale_hw(h, w, s):
    return int(h / (2 ** s)), int(w / (2 ** s))


def _fill_args(imgs, disps, flows, pose, K, K_inv, S, alpha, beta, ac, mode=0, depth_terms=0):
    a = GeomArgs()
    B, _, H, W = imgs[0].shape
    a.B, a.H, a.W, a.num_scales, a.align_corners, a.mode = B, H, W, S, int(ac), int(mode)
    a.depth_terms = int(depth_terms) if mode != 2 else 0
    a.alpha, a.beta = float(alpha), float(beta)
    for f in range(3):
        if tuple(imgs[f].shape) != (B, 3, H, W):
            raise ValueError("frames must be [B,3,H,W]")
        a.img[f] = _dp(imgs[f])
        for s in range(S if mode != 2 else 0):
            hs, ws = _scale_hw(H, W, s)
            if tuple(disps[f][s].shape) != (B, 1, hs, ws):
                raise ValueError("disp[%d][%d] must be %s, got %s" % (f, s, (B, 1, hs, ws), tuple(disps[f][s].shape)))
            a.disp[f][s] = _dp(disps[f][s])
    for d in range(2 if mode != 1 else 0):
        for s in range(S):
            hs, ws = _scale_hw(H, W, s)
            if tuple(flows[d][s].shape) != (B, 2, hs, ws):
                raise ValueError("flow[%d][%d] must be %s, got %s" % (d, s, (B, 2, hs, ws), tuple(flows[d][s].shape)))
            a.flow[d][s] = _dp(flows[d][s])
    if mode != 2:
        if tuple(pose.shape) != (B, 2, 6) or tuple(K.shape) != (B, 3, 3) or (mode == 0 and tuple(K_inv.shape) != (B, 3, 3)):
            raise ValueError("pose must be [B,2,6] and K, K_inv [B,3,3]")
        a.pose, a.K = _dp(pose), _dp(K)
        a.K_inv = _dp(K_inv) if mode == 0 else None
    return a


def _unpack(t, S, mode):
    imgs = t[0:3]
    if mode == 2:
        return imgs, None, [t[3 + d * S: 3 + (d + 1) * S] for d in range(2)], None, None, None
    disps = [t[3 + f * S: 3 + (f + 1) * S] for f in range(3)]
    if mode == 0:
        flows = [t[3 + 3 * S + d * S: 3 + 3 * S + (d + 1) * S] for d in range(2)]
        pose, K, K_inv = t[3 + 5 * S:]
    else:
        flows, (pose, K), K_inv = None, t[3 + 3 * S:], None
    return imgs, disps, flows, pose, K, K_inv


# ---- optional in-step kernel timing (bench.py): HIP events around the launches, read after the timed region
_TIMING = {"on": False, "fwd": [], "bwd": []}


def timing_begin():
    """From now on every fused forward / backward records HIP events between its launches (no host sync)."""
    _TIMING["on"], _TIMING["fwd"], _TIMING["bwd"] = True, [], []


def timing_collect():
    """Stop recording and return (fwd_ms [n,7], bwd_ms [n,6]) for the calls since timing_begin()."""
    import numpy as np
    lib = get_lib()
    _TIMING["on"] = False
    out = []
    for key, nseg in (("fwd", 7), ("bwd", 6)):
        rows = []
        for h in _TIMING[key]:
            buf = (ctypes.c_float * 7)()
            check(lib.dfe_geom_timed_collect(h, ctypes.cast(buf, ctypes.c_void_p)), "dfe_geom_timed_collect")
            rows.append(list(buf)[:nseg])
        _TIMING[key] = []
        out.append(np.array(rows, dtype=np.float64).reshape(-1, nseg))
    return out[0], out[1]


def _launch(kind, a):
    lib = get_lib()
    if _TIMING["on"]:
        h = ctypes.c_void_p()
        fn = lib.dfe_geom_loss_fwd_timed if kind == "fwd" else lib.dfe_geom_loss_bwd_timed
        check(fn(ctypes.byref(a), stream_ptr(), ctypes.byref(h)), "dfe_geom_loss_%s_timed" % kind)
        _TIMING[kind].append(h)
    else:
        fn = lib.dfe_geom_loss_fwd if kind == "fwd" else lib.dfe_geom_loss_bwd
        check(fn(ctypes.byref(a), stream_ptr()), "dfe_geom_loss_%s" % kind)


class GeomLossFn(torch.autograd.Function):
    """forward(*tensors) -> losses [10,B].  Tensor order: 3 frames, 3*S disps (frame-major), then for
    mode 0 (Model_geometry) 2*S flows (bwd scales then fwd scales), pose, K, K_inv; for mode 1 (Model_depth)
    pose, K.  ``dt``: DEPTH_TERM_* bits (mode 0; rows 8 / 9 are zero without them)."""

    @staticmethod
    def forward(ctx, mode, S, alpha, beta, ac, dt, *t):
        lib = get_lib()
        t = [f32c(x) for x in t]
        imgs, disps, flows, pose, K, K_inv = _unpack(t, S, mode)
        a = _fill_args(imgs, disps, flows, pose, K, K_inv, S, alpha, beta, ac, mode, dt)
        n = lib.dfe_geom_workspace_floats(ctypes.byref(a))
        if n < 0:
            check(int(n), "dfe_geom_workspace_floats")
        dev = imgs[0].device
        ws = torch.empty(n, device=dev, dtype=torch.float32)
        losses = torch.empty(len(LOSS_ROWS), a.B, device=dev, dtype=torch.float32)
        a.workspace, a.workspace_floats, a.losses = ws.data_ptr(), n, losses.data_ptr()
        _launch("fwd", a)
        ctx.save_for_backward(*t)
        ctx.cfg = (mode, S, alpha, beta, ac, dt)
        ctx.ws = ws
        ctx.mark_non_differentiable(ws)
        ctx.set_materialize_grads(False)     # no zero-filled "gradient" of the 92 MB workspace output
        return losses, ws

    @staticmethod
    def backward(ctx, glosses, _gws=None):
        lib = get_lib()
        mode, S, alpha, beta, ac, dt = ctx.cfg
        t = list(ctx.saved_tensors)
        imgs, disps, flows, pose, K, K_inv = _unpack(t, S, mode)
        a = _fill_args(imgs, disps, flows, pose, K, K_inv, S, alpha, beta, ac, mode, dt)
        if glosses is None:
            return (None,) * (6 + len(t))
        glosses = f32c(glosses)
        a.workspace, a.workspace_floats = ctx.ws.data_ptr(), ctx.ws.numel()
        a.grad_losses = glosses.data_ptr()
        gd = [[torch.empty_like(x) for x in lst] for lst in disps] if mode != 2 else []
        gf = [[torch.empty_like(x) for x in lst] for lst in flows] if mode != 1 else []
        gp = torch.empty_like(pose) if mode != 2 else None
        for f in range(len(gd)):
            for s in range(S):
                a.grad_disp[f][s] = gd[f][s].data_ptr()
        for d in range(len(gf)):
            for s in range(S):
                a.grad_flow[d][s] = gf[d][s].data_ptr()
        if gp is not None:
            a.grad_pose = gp.data_ptr()
        _launch("bwd", a)
        grads = [None] * 9          # mode, S, alpha, beta, ac, dt, 3 frames
        for f in range(len(gd)):
            grads += gd[f]
        for d in range(len(gf)):
            grads += gf[d]
        grads += {0: [gp, None, None], 1: [gp, None], 2: []}[mode]
        return tuple(grads)


def geom_loss_stack(img_l, img, img_r, disp_l_list, disp_list, disp_r_list, pose_vectors, flows_bwd, flows_fwd,
                    K, K_inv, num_scales=3, flow_consist_alpha=0.01, flow_consist_beta=0.5, align_corners=None,
                    return_masks=False, enable_depth_ssim=False, enable_depth_consis=False):
    """Active ``loss_pack`` entries of Model_geometry.forward as a dict of (B,) tensors.

    ``enable_depth_ssim`` / ``enable_depth_consis`` add the two terms the reference keeps commented
    (model_geometry.py:889-891,897-899) to the same launches: ``loss_depth_ssim`` =
    compute_ssim_loss over the rigid reconstructions, ``loss_depth_consis`` = compute_consis_loss (:182-193) over the
    projected / computed depths, both on the texture-gated masks; they appear in the returned dict only when enabled.

    ``flows_*`` may hold more scales than ``num_scales`` (the reference's zip() drops the 1/8
    flow, model_geometry.py:74-78); extra scales receive no gradient.  With ``return_masks`` the
    second return value maps mask names to per-scale float {0,1} tensors [B,1,Hs,Ws] decoded from
    the kernel's 1-byte mask pack."""
    S = int(num_scales)
    ac = ops.get_align_corners() if align_corners is None else bool(align_corners)
    tensors = [img_l, img, img_r] + list(disp_l_list[:S]) + list(disp_list[:S]) + list(disp_r_list[:S]) \
        + list(flows_bwd[:S]) + list(flows_fwd[:S]) + [pose_vectors, K, K_inv]
    dt = (DEPTH_TERM_SSIM if enable_depth_ssim else 0) | (DEPTH_TERM_CONSIS if enable_depth_consis else 0)
    losses, ws = GeomLossFn.apply(0, S, float(flow_consist_alpha), float(flow_consist_beta), int(ac), dt, *tensors)
    rows = losses.unbind(0)          # one stack in the backward pass instead of a zero-filled [10,B] per selected row
    pack = LossRows({name: rows[i] for i, name in enumerate(LOSS_ROWS[:8])})
    if enable_depth_ssim:
        pack["loss_depth_ssim"] = rows[8]
    if enable_depth_consis:
        pack["loss_depth_consis"] = rows[9]
    pack.rows = (losses, {name: LOSS_ROWS.index(name) for name in pack})
    if not return_masks:
        return pack
    B, _, H, W = img.shape
    if return_masks == "lazy":       # (workspace, dims): decode single masks on demand with decode_mask()
        return pack, (ws, B, H, W, S)
    return pack, decode_masks(ws, B, H, W, S)


def depth_loss_stack(img_l, img, img_r, depth_l_list, depth_list, depth_r_list, pose_vectors, K, num_scales=3,
                     align_corners=None, return_masks=False, enable_depth_ssim=False, enable_depth_consis=False):
    """Active ``loss_pack`` entries of Model_depth.forward (model_depth.py:272-337): ``loss_depth_pixel`` with
    the inverse_warp2-validity x texture mask and ``loss_depth_smooth``, in the fused launches (mode 1).
    ``enable_depth_ssim`` / ``enable_depth_consis`` add the two terms the reference keeps commented there
    (model_depth.py:326-327,332-333: SSIM on the same mask, compute_consis_loss :154-163 without a mask)."""
    S = int(num_scales)
    ac = ops.get_align_corners() if align_corners is None else bool(align_corners)
    tensors = [img_l, img, img_r] + list(depth_l_list[:S]) + list(depth_list[:S]) + list(depth_r_list[:S]) \
        + [pose_vectors, K]
    dt = (DEPTH_TERM_SSIM if enable_depth_ssim else 0) | (DEPTH_TERM_CONSIS if enable_depth_consis else 0)
    losses, ws = GeomLossFn.apply(1, S, 0.0, 0.0, int(ac), dt, *tensors)
    rows = losses.unbind(0)
    pack = LossRows({"loss_depth_pixel": rows[0], "loss_depth_smooth": rows[1]})
    if enable_depth_ssim:
        pack["loss_depth_ssim"] = rows[8]
    if enable_depth_consis:
        pack["loss_depth_consis"] = rows[9]
    pack.rows = (losses, {name: LOSS_ROWS.index(name) for name in pack})
    if not return_masks:
        return pack
    B, _, H, W = img.shape
    m = decode_masks(ws, B, H, W, S)
    return pack, {"valid_to_l": m["valid_bwd"], "valid_to_r": m["valid_fwd"], "texture_bwd": m["texture_bwd"],
                  "texture_fwd": m["texture_fwd"]}


def flow_loss_stack(img_l, img, img_r, flows_bwd, flows_fwd, num_scales=3, align_corners=None):
    """Active ``loss_pack`` entries of Model_flow.forward (model_flow.py:209-255) in the fused launches (mode 2):
    box-mean pyramids, soft Gaussian occlusion weights, weighted L1 / SSIM, flow smoothness and consistency."""
    S = int(num_scales)
    ac = ops.get_align_corners() if align_corners is None else bool(align_corners)
    tensors = [img_l, img, img_r] + list(flows_bwd[:S]) + list(flows_fwd[:S])
    losses, _ws = GeomLossFn.apply(2, S, 0.0, 0.0, int(ac), 0, *tensors)
    rows = losses.unbind(0)
    pack = LossRows({"loss_flow_pixel": rows[2], "loss_flow_ssim": rows[3], "loss_flow_smooth": rows[4], "loss_flow_consis": rows[5]})
    pack.rows = (losses, {name: LOSS_ROWS.index(name) for name in pack})
    return pack


def decode_mask(handle, name, scale=0):
    """One float {0,1} mask [B,1,Hs,Ws] out of the 1-byte mask pack kept in the forward workspace
    (``handle`` = the second return value of ``geom_loss_stack(..., return_masks="lazy")``)."""
    ws, B, H, W, S = handle
    lib = get_lib()
    a = GeomArgs()
    a.B, a.H, a.W, a.num_scales, a.mode = B, H, W, S, 0
    off = lib.dfe_geom_maskpack_offset_bytes(ctypes.byref(a), scale)
    hs, ws_ = _scale_hw(H, W, scale)
    m = ws.view(torch.uint8)[off: off + B * hs * ws_].view(B, 1, hs, ws_)
    return ((m & MASK_BITS[name]) != 0).float()


def decode_masks(ws, B, H, W, S):
    """Decode the 1-byte mask pack kept in the forward workspace into float {0,1} masks."""
    lib = get_lib()
    a = GeomArgs()
    a.B, a.H, a.W, a.num_scales, a.mode = B, H, W, S, 0
    raw = ws.view(torch.uint8)
    out = {k: [] for k in MASK_BITS}
    for s in range(S):
        off = lib.dfe_geom_maskpack_offset_bytes(ctypes.byref(a), s)
        hs, ws_ = _scale_hw(H, W, s)
        m = raw[off: off + B * hs * ws_].view(B, 1, hs, ws_)
        for k, bit in MASK_BITS.items():
            out[k].append(((m & bit) != 0).float())
    return out
