"""torch.autograd wrappers over the C ABI (include/dfe_hip.h).

Each Function is the device-side implementation of one reference free function; the
reference-signature shims live in ``structures/`` and ``pytorch_ssim``.  PyTorch supplies
device memory, the current HIP stream and the autograd graph -- the arithmetic is in
libdfe_hip.so.  There is no CPU fallback: CPU tensors raise ``DfeError``."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib, convs
from ._lib import check, f32c, get_lib, ptr, stream_ptr

_ALIGN_CORNERS = False


def set_align_corners(flag: bool):
    """grid_sample convention used by every warp (the reference leaves it to the installed torch;
    False = torch>=1.3 default = what the golden vectors of this container use)."""
    global _ALIGN_CORNERS
    _ALIGN_CORNERS = bool(flag)


def get_align_corners() -> bool:
    return _ALIGN_CORNERS


def _ac(flag):
    return int(_ALIGN_CORNERS if flag is None else bool(flag))


def scatter_ws(n, device):
    """Workspace of the order-independent scatter-add into ``n`` floats (csrc/dfe_scatter.h; zero-filled by the library)."""
    return torch.empty(get_lib().dfe_scatter_ws_bytes(int(n)), device=device, dtype=torch.uint8)


# --------------------------------------------------------------------------- cameras
def prepare_cameras(pose, K, downscales):
    """pose [B,ndir,6] or [B,6], K [B,3,3] -> opaque camera buffer [B*ndir*len(downscales), 66]."""
    lib = get_lib()
    pose = f32c(pose.detach())
    if pose.dim() == 2:
        pose = pose.unsqueeze(1)
    B, ndir = pose.shape[0], pose.shape[1]
    K = f32c(K.detach())
    n = len(downscales)
    cams = torch.empty(B * ndir * n, lib.dfe_camera_floats(), device=pose.device, dtype=torch.float32)
    ds = (ctypes.c_float * n)(*[float(d) for d in downscales])
    check(lib.dfe_prepare_cameras(ptr(pose), ptr(K), ptr(cams), B, ndir, n, ctypes.cast(ds, ctypes.c_void_p),
                                  stream_ptr()), "dfe_prepare_cameras")
    return cams


# --------------------------------------------------------------------------- warp_flow
class WarpFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, use_mask, align_corners):
        lib = get_lib()
        x, flow = f32c(x), f32c(flow)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.dfe_warp_flow_fwd(ptr(x), ptr(flow), ptr(out), B, C, H, W, int(use_mask), align_corners,
                                    stream_ptr()), "dfe_warp_flow_fwd")
        ctx.save_for_backward(x, flow)
        ctx.cfg = (int(use_mask), align_corners)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        x, flow = ctx.saved_tensors
        use_mask, ac = ctx.cfg
        B, C, H, W = x.shape
        gout = f32c(gout)
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None         # every element written by the library
        if gflow is None and gx is None:
            return None, None, None, None
        ws = scatter_ws(x.numel(), x.device) if gx is not None else None
        check(lib.dfe_warp_flow_bwd(ptr(x), ptr(flow), ptr(gout), ptr(gflow), ptr(gx), ptr(ws), B, C, H, W, use_mask, ac,
                                    stream_ptr()), "dfe_warp_flow_bwd")
        return gx, gflow, None, None


def warp_flow(x, flow, use_mask=False, align_corners=None):
    B, C, H, W = x.size()
    if tuple(flow.shape) != (B, 2, H, W):
        raise ValueError("the shape of grid {0} is not equal to the shape of flow {1}.".format(
            torch.Size((B, 2, H, W)), flow.shape))
    return WarpFlowFn.apply(x, flow, bool(use_mask), _ac(align_corners))


# --------------------------------------------------------------------------- pose matrices
class PoseMatsFn(torch.autograd.Function):
    """vec [n,6] -> (T34 [n,3,4], E [n,3,3])."""

    @staticmethod
    def forward(ctx, vec):
        lib = get_lib()
        vec = f32c(vec)
        n = vec.shape[0]
        T = torch.empty(n, 3, 4, device=vec.device, dtype=torch.float32)
        E = torch.empty(n, 3, 3, device=vec.device, dtype=torch.float32)
        check(lib.dfe_pose_vec2mat_fwd(ptr(vec), ptr(T), ptr(E), n, stream_ptr()), "dfe_pose_vec2mat_fwd")
        ctx.save_for_backward(vec)
        return T, E

    @staticmethod
    def backward(ctx, gT, gE):
        lib = get_lib()
        (vec,) = ctx.saved_tensors
        gvec = torch.empty_like(vec)
        gT = f32c(gT) if gT is not None else None
        gE = f32c(gE) if gE is not None else None
        check(lib.dfe_pose_vec2mat_bwd(ptr(vec), ptr(gT), ptr(gE), ptr(gvec), vec.shape[0], stream_ptr()),
              "dfe_pose_vec2mat_bwd")
        return gvec


# --------------------------------------------------------------------------- inverse_warp2
class InverseWarp2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, ref_depth, pose, intrinsics, align_corners):
        lib = get_lib()
        img, depth, ref_depth = f32c(img), f32c(depth), f32c(ref_depth)
        pose, intrinsics = f32c(pose), f32c(intrinsics)
        B, _, H, W = img.shape
        cams = prepare_cameras(pose, intrinsics, [1.0])
        dev = img.device
        o_img = torch.empty(B, 3, H, W, device=dev)
        o_valid = torch.empty(B, 1, H, W, device=dev)
        o_pd = torch.empty(B, 1, H, W, device=dev)
        o_cd = torch.empty(B, 1, H, W, device=dev)
        check(lib.dfe_inverse_warp2_fwd(ptr(img), ptr(depth), ptr(ref_depth), ptr(cams), ptr(o_img), ptr(o_valid),
                                        ptr(o_pd), ptr(o_cd), B, H, W, align_corners, stream_ptr()),
              "dfe_inverse_warp2_fwd")
        ctx.save_for_backward(img, depth, ref_depth, cams)
        ctx.ac = align_corners
        ctx.mark_non_differentiable(o_valid)
        return o_img, o_valid, o_pd, o_cd

    @staticmethod
    def backward(ctx, g_img, g_valid, g_pd, g_cd):
        lib = get_lib()
        img, depth, ref_depth, cams = ctx.saved_tensors
        B, _, H, W = img.shape
        dev = img.device
        if ctx.needs_input_grad[0]:
            raise _lib.DfeError("gradient wrt the source image of inverse_warp2 is not implemented "
                                "(images are inputs on the reference's path)")
        g_img = f32c(g_img) if g_img is not None else None
        g_pd = f32c(g_pd) if g_pd is not None else None
        g_cd = f32c(g_cd) if g_cd is not None else None
        g_depth = torch.empty(B, 1, H, W, device=dev)
        g_ref = torch.empty(B, 1, H, W, device=dev) if (ctx.needs_input_grad[2] and g_pd is not None) else None
        g_ref_ws = scatter_ws(B * H * W, dev) if g_ref is not None else None
        g_pose = torch.empty(B, 6, device=dev)
        ws = torch.empty(lib.dfe_pose_partials_floats(B, H, W), device=dev)
        check(lib.dfe_inverse_warp2_bwd(ptr(img), ptr(depth), ptr(ref_depth), ptr(cams), ptr(g_img), ptr(g_pd),
                                        ptr(g_cd), ptr(g_depth), ptr(g_ref), ptr(g_ref_ws), ptr(g_pose), ptr(ws), B, H, W,
                                        ctx.ac, stream_ptr()), "dfe_inverse_warp2_bwd")
        return None, g_depth, g_ref, g_pose, None, None


# --------------------------------------------------------------------------- rigid flow
class RigidFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, pose, intrinsics):
        lib = get_lib()
        depth, pose, intrinsics = f32c(depth), f32c(pose), f32c(intrinsics)
        B, _, H, W = depth.shape
        cams = prepare_cameras(pose, intrinsics, [1.0])
        out = torch.empty(B, 2, H, W, device=depth.device)
        check(lib.dfe_rigid_flow_fwd(ptr(depth), ptr(cams), ptr(out), B, H, W, stream_ptr()), "dfe_rigid_flow_fwd")
        ctx.save_for_backward(depth, cams)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        depth, cams = ctx.saved_tensors
        B, _, H, W = depth.shape
        g_depth = torch.empty_like(depth)
        g_pose = torch.empty(B, 6, device=depth.device)
        ws = torch.empty(lib.dfe_pose_partials_floats(B, H, W), device=depth.device)
        check(lib.dfe_rigid_flow_bwd(ptr(depth), ptr(cams), ptr(f32c(gout)), ptr(g_depth), ptr(g_pose), ptr(ws),
                                     B, H, W, stream_ptr()), "dfe_rigid_flow_bwd")
        return g_depth, g_pose, None


# --------------------------------------------------------------------------- SSIM
class SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        lib = get_lib()
        x, y = f32c(x), f32c(y)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.dfe_ssim_fwd(ptr(x), ptr(y), ptr(out), B, C, H, W, stream_ptr()), "dfe_ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None
        check(lib.dfe_ssim_bwd(ptr(x), ptr(y), ptr(f32c(gout)), ptr(gx), ptr(gy), B, C, H, W, stream_ptr()),
              "dfe_ssim_bwd")
        return gx, gy


# --------------------------------------------------------------------------- correlation
class CorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, d):
        lib = get_lib()
        f1, f2 = f32c(f1), f32c(f2)
        B, C, H, W = f1.shape
        out = torch.empty(B, (2 * d + 1) ** 2, H, W, device=f1.device)
        check(lib.dfe_corr_fwd(ptr(f1), ptr(f2), ptr(out), B, C, H, W, d, stream_ptr()), "dfe_corr_fwd")
        ctx.save_for_backward(f1, f2)
        ctx.d = d
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        f1, f2 = ctx.saved_tensors
        B, C, H, W = f1.shape
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        if g1 is None and g2 is None:
            return None, None, None
        check(lib.dfe_corr_bwd(ptr(f1), ptr(f2), ptr(f32c(gout)), ptr(g1), ptr(g2), B, C, H, W, ctx.d, stream_ptr()),
              "dfe_corr_bwd")
        return g1, g2, None


def corr81(f1, f2, d=4):
    assert f1.shape == f2.shape
    if d != 4:
        raise _lib.DfeError("corr_naive: only d=4 (81 taps) is implemented in HIP")
    return CorrFn.apply(f1, f2, int(d))


class MaxPool3x3s2Fn(torch.autograd.Function):
    """F.max_pool2d(x, 3, 2, 1) (the ResNet stem, depth_model.py:60-95) with a 1-byte window position instead of
    int64 indices; values and gradients bit-identical to ATen's."""

    @staticmethod
    def forward(ctx, x):
        lib = get_lib()
        x = f32c(x)
        B, C, H, W = x.shape
        Ho, Wo = lib.dfe_maxpool3x3s2_out(H), lib.dfe_maxpool3x3s2_out(W)
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        idx = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.uint8)
        check(lib.dfe_maxpool3x3s2_fwd(ptr(x), ptr(y), ptr(idx), B * C, H, W, stream_ptr()), "dfe_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        idx, = ctx.saved_tensors
        B, C = idx.shape[:2]
        H, W = ctx.hw
        gx = torch.empty(B, C, H, W, device=gy.device, dtype=torch.float32)
        check(lib.dfe_maxpool3x3s2_bwd(ptr(f32c(gy)), ptr(idx), ptr(gx), B * C, H, W, stream_ptr()), "dfe_maxpool3x3s2_bwd")
        return gx


def maxpool3x3s2(x):
    if x.dim() != 4:
        raise ValueError("maxpool3x3s2: x must be [B,C,H,W]")
    if x.shape[0] * x.shape[1] > 65535:
        raise _lib.DfeError("maxpool3x3s2: more than 65535 planes")
    return MaxPool3x3s2Fn.apply(x)


class PwcLevelInputFn(torch.autograd.Function):
    """x = cat(corr(c1, warp(c2, flow)), c1, flow) of one PWC decoder level (pwc_tf.py:119-121) as one operator."""

    @staticmethod
    def forward(ctx, c1, c2, flow, align_corners):
        lib = get_lib()
        c1, c2, flow = f32c(c1), f32c(c2), f32c(flow)
        B, C, H, W = c1.shape
        warped = torch.empty_like(c2)
        x = torch.empty(B, lib.dfe_pwc_level_channels(C), H, W, device=c1.device, dtype=torch.float32)
        # when dL/dc2 will be asked for, the feature warp also counts its taps per target pixel and the inverse map the backward's
        # gather walks is finished here (dfe_pwc_level_fwd_map): the backward pass is then three launches at every level
        ctx.map = None
        if PWC_LEVEL_MAP and ctx.needs_input_grad[1] and C >= 8:
            ctx.map = torch.empty(lib.dfe_pwc_level_map_bytes(B, H, W), device=c1.device, dtype=torch.uint8)
            check(lib.dfe_pwc_level_fwd_map(ptr(c1), ptr(c2), ptr(flow), ptr(warped), ptr(x), ptr(ctx.map), B, C, H, W, align_corners,
                                            stream_ptr()), "dfe_pwc_level_fwd_map")
        else:
            check(lib.dfe_pwc_level_fwd(ptr(c1), ptr(c2), ptr(flow), ptr(warped), ptr(x), B, C, H, W, align_corners,
                                        stream_ptr()), "dfe_pwc_level_fwd")
        ctx.save_for_backward(c1, c2, flow, warped)
        ctx.ac = align_corners
        return x

    @staticmethod
    def backward(ctx, gx):
        lib = get_lib()
        c1, c2, flow, warped = ctx.saved_tensors
        B, C, H, W = c1.shape
        gx = f32c(gx)
        g_c1 = torch.empty_like(c1)
        g_c2 = torch.empty_like(c2) if ctx.needs_input_grad[1] else None    # every element written by the library
        g_flow = torch.empty_like(flow) if (ctx.needs_input_grad[2] or g_c2 is None) else None
        g_warped = torch.empty_like(c2)
        if ctx.map is not None and g_c2 is not None:
            check(lib.dfe_pwc_level_bwd_map(ptr(c1), ptr(c2), ptr(flow), ptr(warped), ptr(gx), ptr(g_warped), ptr(g_c1),
                                            ptr(g_c2), ptr(ctx.map), ptr(g_flow), B, C, H, W, ctx.ac, stream_ptr()), "dfe_pwc_level_bwd_map")
        else:
            ws = scatter_ws(c2.numel(), c2.device) if g_c2 is not None else None
            check(lib.dfe_pwc_level_bwd(ptr(c1), ptr(c2), ptr(flow), ptr(warped), ptr(gx), ptr(g_warped), ptr(g_c1),
                                        ptr(g_c2), ptr(ws), ptr(g_flow), B, C, H, W, ctx.ac, stream_ptr()), "dfe_pwc_level_bwd")
        return g_c1, g_c2, (g_flow if ctx.needs_input_grad[2] else None), None


def pwc_level_input(c1, c2, flow, align_corners=None):
    """``torch.cat((corr(c1, warp(c2, flow)), c1, flow), 1)`` [B, 81+C+2, H, W]."""
    if c1.shape != c2.shape or tuple(flow.shape) != (c1.shape[0], 2, c1.shape[2], c1.shape[3]):
        raise ValueError("pwc_level_input: c1 %s, c2 %s, flow %s" % (tuple(c1.shape), tuple(c2.shape), tuple(flow.shape)))
    return PwcLevelInputFn.apply(c1, c2, flow, _ac(align_corners))


# --------------------------------------------------------------------------- resize (no grad)
def resize(img, out_hw, mode):
    """mode 'bilinear' (align_corners=False) or 'area'; images carry no gradient on this path."""
    lib = get_lib()
    img = f32c(img.detach())
    B, C, H, W = img.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty(B, C, oh, ow, device=img.device)
    check(lib.dfe_resize(ptr(img), ptr(out), B * C, H, W, oh, ow, {"bilinear": 0, "area": 1}[mode], stream_ptr()),
          "dfe_resize")
    return out


# --------------------------------------------------------------------------- device-side input pipeline
class ResizeBilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) with a scalar on one side of it, as PWC_tf writes
    it (pwc_tf.py:118-119, 175-178): ``pre`` -> resize(x * mult), else resize(x) * mult.  Forward in ATen's (CPU)
    association, backward as a gather (reproducible)."""

    @staticmethod
    def forward(ctx, x, out_hw, mult, pre):
        lib = get_lib()
        x = f32c(x)
        B, C, H, W = x.shape
        oh, ow = int(out_hw[0]), int(out_hw[1])
        out = torch.empty(B, C, oh, ow, device=x.device, dtype=torch.float32)
        check(lib.dfe_resize_bilinear_fwd(ptr(x), ptr(out), B * C, H, W, oh, ow, float(mult), int(pre), stream_ptr()),
              "dfe_resize_bilinear_fwd")
        ctx.cfg = (B, C, H, W, oh, ow, float(mult), int(pre))
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        B, C, H, W, oh, ow, mult, pre = ctx.cfg
        gin = torch.empty(B, C, H, W, device=gout.device, dtype=torch.float32)
        check(lib.dfe_resize_bilinear_bwd(ptr(f32c(gout)), ptr(gin), B * C, H, W, oh, ow, mult, pre, stream_ptr()),
              "dfe_resize_bilinear_bwd")
        return gin, None, None, None


def resize_bilinear(x, out_hw, mult=1.0, pre=False):
    """Differentiable ``F.interpolate(x * mult if pre else x, out_hw, mode='bilinear') * (1 if pre else mult)``;
    up-sampling ratios of at most 4 per axis (DfeError beyond that in the backward pass)."""
    return ResizeBilinearFn.apply(x, (int(out_hw[0]), int(out_hw[1])), float(mult), bool(pre))


def prepare_triplets(raw_u8, img_hw, flip=None):
    """KITTI_Prepared.__getitem__'s image half (kitti_prepared.py:63-90,132-152) for a whole batch on the device.

    raw_u8: uint8 [B, 3*H0, W0, 3] stacked triplets as stored (pin + ``.to(device, non_blocking=True)`` it); flip: optional
    uint8 / bool [B]; returns fp32 [B, 3, 3*H, W] in [0,1] -- the ``images`` entry of the models' ``inputs``."""
    lib = get_lib()
    if raw_u8.dtype != torch.uint8 or raw_u8.dim() != 4 or raw_u8.shape[3] != 3 or raw_u8.shape[1] % 3:
        raise ValueError("raw_u8 must be uint8 [B, 3*H0, W0, 3]")
    raw_u8 = raw_u8.contiguous()
    B, H0, W0 = raw_u8.shape[0], raw_u8.shape[1] // 3, raw_u8.shape[2]
    H, W = int(img_hw[0]), int(img_hw[1])
    fl = None if flip is None else flip.to(device=raw_u8.device, dtype=torch.uint8).contiguous()
    out = torch.empty(B, 3, 3 * H, W, device=raw_u8.device)
    check(lib.dfe_prepare_triplets(ptr(raw_u8), ptr(fl), ptr(out), B, H0, W0, H, W, stream_ptr()), "dfe_prepare_triplets")
    return out


def rescale_intrinsics(K, img_hw_orig, img_hw_new, num_scales):
    """rescale_intrinsics + get_multiscale_intrinsics (kitti_prepared.py:110-130): K [3,3] float64 numpy ->
    (K_ms, K_inv_ms) float32 tensors [S,3,3] (host side: a few dozen flops per sample)."""
    import numpy as np
    K = np.array(K, dtype=np.float64, copy=True)
    K[0, :] = K[0, :] * img_hw_new[1] / img_hw_orig[1]
    K[1, :] = K[1, :] * img_hw_new[0] / img_hw_orig[0]
    ks, kis = [], []
    for s in range(num_scales):
        k = K.copy()
        k[0, :] /= 2 ** s
        k[1, :] /= 2 ** s
        ks.append(k); kis.append(np.linalg.inv(k))
    return torch.from_numpy(np.stack(ks)).float(), torch.from_numpy(np.stack(kis)).float()


# --------------------------------------------------------------------------- forward splat (no grad)
def forward_splat_ones(flow, clamp=True):
    """Bilinear forward warp of a ones image by ``flow`` [B,2,H,W] -> [B,1,H,W] (model_flow.py:33-39's intent)."""
    lib = get_lib()
    fl = f32c(flow.detach())
    B, C, H, W = fl.shape
    if C != 2:
        raise ValueError("flow must be [B,2,H,W]")
    out = torch.empty(B, 1, H, W, device=fl.device)
    ws = scatter_ws(B * H * W, fl.device)
    check(lib.dfe_forward_splat_ones(ptr(fl), ptr(out), ptr(ws), B, H, W, int(bool(clamp)), stream_ptr()),
          "dfe_forward_splat_ones")
    return out


# --------------------------------------------------------------------------- mask decisions (no grad)
def occ_masks(from_l, tgt, from_r):
    """compute_occ_weight's decisions (model_geometry.py:105-132) -> occ_bwd, occ_fwd, valid_bwd, valid_fwd [B,1,H,W]."""
    lib = get_lib()
    il, it, ir = (f32c(t.detach()) for t in (from_l, tgt, from_r))
    B, C, H, W = it.shape
    if C != 3 or il.shape != it.shape or ir.shape != it.shape:
        raise ValueError("occ_masks expects three [B,3,H,W] tensors")
    outs = [torch.empty(B, 1, H, W, device=it.device) for _ in range(4)]
    check(lib.dfe_occ_masks(ptr(il), ptr(it), ptr(ir), *[ptr(o) for o in outs], B, H, W, stream_ptr()), "dfe_occ_masks")
    return outs


def texture_mask(img, warped, source):
    """compute_texture_mask's decision (model_geometry.py:134-140) -> [B,1,H,W] in {0,1}."""
    lib = get_lib()
    a, b, c = (f32c(t.detach()) for t in (img, warped, source))
    B, C, H, W = a.shape
    if C != 3 or b.shape != a.shape or c.shape != a.shape:
        raise ValueError("texture_mask expects three [B,3,H,W] tensors")
    out = torch.empty(B, 1, H, W, device=a.device)
    check(lib.dfe_texture_mask(ptr(a), ptr(b), ptr(c), ptr(out), B, H, W, stream_ptr()), "dfe_texture_mask")
    return out


def dynamic_mask(flow, rigid, alpha, beta):
    """compute_dynamic_mask's decision and score (model_geometry.py:699-711) -> mask, score [B,1,H,W]."""
    lib = get_lib()
    f, r = f32c(flow.detach()), f32c(rigid.detach())
    B, C, H, W = f.shape
    if C != 2 or r.shape != f.shape:
        raise ValueError("dynamic_mask expects two [B,2,H,W] tensors")
    mask, score = torch.empty(B, 1, H, W, device=f.device), torch.empty(B, 1, H, W, device=f.device)
    check(lib.dfe_dynamic_mask(ptr(f), ptr(r), ptr(mask), ptr(score), float(alpha), float(beta), B, H, W, stream_ptr()),
          "dfe_dynamic_mask")
    return mask, score


# --------------------------------------------------------------------------- depth-decoder glue
class EluPadFn(torch.autograd.Function):
    """reflect_pad1(elu(x + bias)) (elu, bias optional): the input of a Conv3x3 whose producer is a ConvBlock
    (depth_model.py); x is that block's convolution output computed without its bias."""

    @staticmethod
    def forward(ctx, x, bias, apply_elu):
        lib = get_lib()
        x = f32c(x)
        bias = f32c(bias) if bias is not None else None
        B, C, H, W = x.shape
        out = torch.empty(B, C, H + 2, W + 2, device=x.device)
        check(lib.dfe_elu_pad_fwd(ptr(x), ptr(bias), ptr(out), B, C, H, W, int(apply_elu), stream_ptr()), "dfe_elu_pad_fwd")
        ctx.apply_elu = int(apply_elu)
        ctx.shape = (B, C, H, W)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x if apply_elu else x.new_empty(0), bias if bias is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        x, bias = ctx.saved_tensors
        B, C, H, W = ctx.shape
        gout = f32c(gout)
        gx = torch.empty(B, C, H, W, device=gout.device)
        gb = part = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            gb = torch.empty(C, device=gout.device)
            part = torch.empty(lib.dfe_glue_partials_floats(B, C, H, W), device=gout.device)
        check(lib.dfe_elu_pad_bwd(ptr(x) if ctx.apply_elu else None, ptr(bias) if ctx.has_bias else None, ptr(gout), ptr(gx),
                                  ptr(gb), ptr(part), B, C, H, W, ctx.apply_elu, stream_ptr()), "dfe_elu_pad_bwd")
        return gx, gb, None


class EluUp2CatPadFn(torch.autograd.Function):
    """reflect_pad1(cat(bilinear_x2(elu(x + bias)), skip)): one decoder stage's glue (depth_model.py: ELU, upsample,
    cat, pad); x is the stage's first convolution output computed without its bias."""

    @staticmethod
    def forward(ctx, x, bias, skip):
        lib = get_lib()
        x = f32c(x)
        bias = f32c(bias) if bias is not None else None
        B, C1, h, w = x.shape
        C2 = 0
        if skip is not None:
            skip = f32c(skip)
            C2 = skip.shape[1]
            if tuple(skip.shape) != (B, C2, 2 * h, 2 * w):
                raise ValueError("skip must be [B,C2,2h,2w] = %s, got %s" % ((B, C2, 2 * h, 2 * w), tuple(skip.shape)))
        out = torch.empty(B, C1 + C2, 2 * h + 2, 2 * w + 2, device=x.device)
        check(lib.dfe_elu_up2_cat_pad_fwd(ptr(x), ptr(bias), ptr(skip), ptr(out), B, C1, C2, h, w, stream_ptr()),
              "dfe_elu_up2_cat_pad_fwd")
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, bias if bias is not None else x.new_empty(0))
        ctx.dims = (B, C1, C2, h, w)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        x, bias = ctx.saved_tensors
        B, C1, C2, h, w = ctx.dims
        gout = f32c(gout)
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        gx = torch.empty_like(x) if (ctx.needs_input_grad[0] or want_b) else None
        gskip = torch.empty(B, C2, 2 * h, 2 * w, device=x.device) if (C2 > 0 and ctx.needs_input_grad[2]) else None
        if gx is None and gskip is None:
            return None, None, None
        gb = part = None
        if want_b:
            gb = torch.empty(C1, device=x.device)
            part = torch.empty(lib.dfe_glue_partials_floats(B, C1, h, w), device=x.device)
        check(lib.dfe_elu_up2_cat_pad_bwd(ptr(x), ptr(bias) if ctx.has_bias else None, ptr(gout), ptr(gx), ptr(gskip), ptr(gb),
                                          ptr(part), B, C1, C2, h, w, stream_ptr()), "dfe_elu_up2_cat_pad_bwd")
        return gx, gb, gskip


def elu_pad(x, bias=None, apply_elu=True):
    return EluPadFn.apply(x, bias, bool(apply_elu))


def elu_up2_cat_pad(x, bias=None, skip=None):
    return EluUp2CatPadFn.apply(x, bias, skip)


# --------------------------------------------------------------------------- convolution epilogue (bias + activation)
class BiasActFn(torch.autograd.Function):
    """z <- act(z + bias[c]) in place on a fresh convolution output; act(v) = v > 0 ? v : slope * v."""

    @staticmethod
    def forward(ctx, z, bias, slope):
        lib = get_lib()
        if not z.is_contiguous() or z.dtype != torch.float32:
            raise _lib.DfeError("bias_act expects a contiguous fp32 convolution output")
        B, C, H, W = z.shape
        b = f32c(bias) if bias is not None else None
        check(lib.dfe_bias_act_fwd(ptr(z), ptr(b), B, C, H, W, float(slope), stream_ptr()), "dfe_bias_act_fwd")
        ctx.mark_dirty(z)
        ctx.save_for_backward(z)
        ctx.slope = float(slope)
        ctx.has_bias = bias is not None
        return z

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        (y,) = ctx.saved_tensors
        B, C, H, W = y.shape
        if gy.dtype != torch.float32:
            gy = gy.float()
        # a channel slice of a wider contiguous tensor (the gradient of a torch.cat) is read in place
        if not (gy.stride(3) == 1 and gy.stride(2) == W and gy.stride(1) == H * W and gy.stride(0) >= C * H * W):
            gy = gy.contiguous()
        gz = torch.empty_like(y)
        gb = part = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            gb = torch.empty(C, device=y.device, dtype=torch.float32)
            part = torch.empty(lib.dfe_bias_act_partials_floats(B, C, H, W), device=y.device, dtype=torch.float32)
        check(lib.dfe_bias_act_bwd(ptr(y), ptr(gy, strided=True), gy.stride(0), ptr(gz), ptr(gb), ptr(part), B, C, H, W, ctx.slope,
                                   stream_ptr()), "dfe_bias_act_bwd")
        return gz, gb, None


def bias_act(z, bias, slope):
    return BiasActFn.apply(z, bias, float(slope))


# --------------------------------------------------------------------------- small-plane 3x3 convolutions (fp32 MFMA)
# Planes of at most this many pixels per sample take the dfe_planeconv_* kernels inside DenseDecodeFn (PWC levels 6 and 5 of
# a 256x832 frame: 52 and 208 pixels); 0 = every layer stays on MIOpen.
PLANECONV_MAX_HW = int(os.environ.get("DFE_PLANECONV_MAX_HW", "208"))
WINO_EPILOGUE = os.environ.get("DFE_WINO_EPILOGUE", "1") != "0"      # switches read once at import (246 environment reads per step before)
FLOW_HEAD = os.environ.get("DFE_FLOW_HEAD", "1") != "0"
PWC_LEVEL_MAP = os.environ.get("DFE_PWC_LEVEL_MAP", "1") != "0"      # the PWC level's inverse map built in its forward pass (PwcLevelInputFn)


def planeconv_eligible(x, w):
    if PLANECONV_MAX_HW <= 0 or not x.is_cuda or x.dim() != 4:
        return False
    B, Ci, H, W = x.shape
    return (x.dtype == torch.float32 and w.dtype == torch.float32 and tuple(w.shape[1:]) == (Ci, 3, 3)
            and H * W <= PLANECONV_MAX_HW and get_lib().dfe_planeconv_supported(B, Ci, int(w.shape[0]), H, W) == 1)


def _planeconv_ws(B, Ci, Co, H, W, dev):
    return torch.empty(get_lib().dfe_planeconv_ws_floats(B, Ci, Co, H, W), device=dev, dtype=torch.float32)


def planeconv_fwd_into(x, w, bias, slope, d1, d1_off=0, d2=None, d2_off=0):
    """act(conv3x3(x, w) + bias) into channels d1_off.. of d1 and (optionally) d2_off.. of d2 (contiguous NCHW buffers)."""
    B, Ci, H, W = x.shape
    Co, HW = int(w.shape[0]), H * W
    p1 = ctypes.c_void_p(d1.data_ptr() + 4 * d1_off * HW)
    p2 = ctypes.c_void_p(d2.data_ptr() + 4 * d2_off * HW) if d2 is not None else None
    ws = _planeconv_ws(B, Ci, Co, H, W, x.device)
    check(get_lib().dfe_planeconv_fwd(ptr(x), ptr(w), ptr(bias), float(slope), p1, d1.stride(0), p2,
                                      d2.stride(0) if d2 is not None else 0, ptr(ws), B, Ci, Co, H, W, stream_ptr()),
          "dfe_planeconv_fwd")


def planeconv_forward(x, w, bias=None, slope=1.0):
    x, w = f32c(x), f32c(w)
    y = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], device=x.device, dtype=torch.float32)
    planeconv_fwd_into(x, w, None if bias is None else f32c(bias), slope, y)
    return y


def planeconv_backward(gy, x, w, want_x=True, want_w=True):
    """(gx, gw) of conv3x3(x, w), pad 1."""
    gy, x, w = f32c(gy), f32c(x), f32c(w)
    B, Ci, H, W = x.shape
    Co = int(w.shape[0])
    lib = get_lib()
    ws = _planeconv_ws(B, Ci, Co, H, W, x.device)
    gx = gw = None
    if want_x:
        gx = torch.empty_like(x)
        check(lib.dfe_planeconv_dgrad(ptr(gy), ptr(w), ptr(gx), ptr(ws), B, Ci, Co, H, W, stream_ptr()), "dfe_planeconv_dgrad")
    if want_w:
        gw = torch.empty_like(w)
        check(lib.dfe_planeconv_wgrad(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, Co, H, W, stream_ptr()), "dfe_planeconv_wgrad")
    return gx, gw


# --------------------------------------------------------------------------- Winograd F(2x2, 3x3) on the fp32 matrix cores
class WinoWeightCache:
    """Transformed filters (U = G g G^T, both orientations) of the parameters the Winograd kernel has been called with, kept
    across steps.  A training step uses every filter twice (forward, data gradient) and each call used to launch its own
    transform (96 launches of ~5.6 us per step of the joint model); the filters only change in the optimiser step, so
    ``optim.FusedAdam.step`` calls ``refresh()``: ONE launch (dfe_wino_transform_weights_multi) on the optimiser's stream,
    right behind the update -- whatever orders the next step's network streams behind the update orders them behind this.

    Validity: an entry is used only while the parameter object it was made for is alive, still owns the same storage and its
    autograd version counter equals the one recorded at the refresh -- ``load_state_dict``, ``copy_`` or any other in-place
    torch operation makes the entry miss, and the call transforms for itself (into its own scratch, on its own stream) as
    before.  FusedAdam writes parameters through raw pointers (no version bump), which is why it is the one that refreshes.
    Only ``refresh`` writes the cached buffers.  ``DFE_WINO_CACHE=0`` turns the cache off."""

    def __init__(self):
        self.enabled = os.environ.get("DFE_WINO_CACHE", "1") != "0"
        # DFE_WINO_CACHE_VERIFY=1: every hit re-transforms the filter and compares (synchronises; a debugging mode that catches
        # writes the version counters cannot see)
        self.verify = os.environ.get("DFE_WINO_CACHE_VERIFY", "0") == "1"
        self.entries = {}        # (data_ptr, Co, Ci of the weight tensor) -> entry
        self.table = None        # device tables of the last refresh (rebuilt when the membership changes)
        self.blockmap = None
        self.members = None
        self.hits = self.misses = 0
        if os.environ.get("DFE_WINO_CACHE_STATS"):
            import atexit
            import sys
            atexit.register(lambda: sys.stderr.write("[dfe] wino weight cache: %d filters, %d hits, %d misses\n"
                                                     % (len(self.entries), self.hits, self.misses)))

    def lookup(self, w, transposed):
        """The cached U for this call or None (transform per call).  Registers the parameter on first sight."""
        if not self.enabled or not w.is_cuda or w.dim() != 4:
            return None
        key = (w.data_ptr(), int(w.shape[0]), int(w.shape[1]))
        e = self.entries.get(key)
        if e is None:
            # only leaf parameters are worth keeping (a temporary's address is reused by unrelated tensors)
            # (and only what refresh() can transform: a non-fp32 / non-contiguous parameter would be registered again on every call)
            if isinstance(w, torch.nn.Parameter) and w.dtype == torch.float32 and w.is_contiguous():
                import weakref
                self.entries[key] = {"ref": weakref.ref(w), "version": -1, "U": None, "device": w.device}
                self.members = None      # the device table of the last refresh no longer describes the membership
            self.misses += 1
            return None
        p = e["ref"]()
        if p is None or p.data_ptr() != key[0]:
            # the Parameter died (its address may already belong to another tensor -- or to a NEW Parameter of the same shape,
            # which would re-register under the same key: the table must be rebuilt, its U pointers are about to be freed)
            del self.entries[key]
            self.members = None
            self.misses += 1
            return None
        if e["U"] is None or e["version"] != w._version or e["version"] != p._version:
            self.misses += 1
            return None
        self.hits += 1
        U = e["U"][1 if transposed else 0]
        if self.verify and not torch.equal(U, self.transform_now(w, transposed)):
            raise _lib.DfeError("WinoWeightCache: the cached filters of a %s parameter are stale -- it was written without a version "
                                "bump (p.data.mul_/copy_, a raw-pointer kernel) after the last refresh; call "
                                "ops.wino_weights.invalidate() after such writes" % (tuple(w.shape),))
        return U

    @staticmethod
    def transform_now(w, transposed):
        """U of ``w`` by a one-entry launch of the same kernel refresh() uses (DFE_WINO_CACHE_VERIFY=1; tests)."""
        lib = get_lib()
        Co, Ci = int(w.shape[0]), int(w.shape[1])
        K, C = (Ci, Co) if transposed else (Co, Ci)
        U = torch.empty(lib.dfe_wino_weight_floats(C, K), device=w.device, dtype=torch.float32)
        n = int(lib.dfe_wino_transform_blocks(C, K))
        table = torch.tensor([[w.data_ptr(), U.data_ptr(), K, C, int(bool(transposed)), 0]], dtype=torch.int64).to(w.device)
        bmap = torch.zeros(n, dtype=torch.int32, device=w.device)
        check(lib.dfe_wino_transform_weights_multi(ptr(table), ptr(bmap), n, stream_ptr()), "dfe_wino_transform_weights_multi")
        return U

    def invalidate(self):
        """Call after any write to a registered parameter that does not bump its autograd version counter (``p.data.copy_`` /
        ``mul_``, EMA or weight surgery through ``.data``, another raw-pointer kernel): every entry misses until the next
        refresh().  ddp.FlatAllReduce (after its ``.data`` broadcast) and the checkpoint loaders call it."""
        for e in self.entries.values():
            e["version"] = -1

    def refresh(self):
        """Transform every registered filter (forward and data-gradient orientation) in one launch on the current stream."""
        if not self.enabled or not self.entries:
            return
        lib = get_lib()
        live = []
        for key, e in list(self.entries.items()):
            p = e["ref"]()
            if p is None or p.data_ptr() != key[0] or not p.is_contiguous() or p.dtype != torch.float32:
                del self.entries[key]
                self.members = None
                continue
            live.append((key, e, p))
        if not live:
            return
        dev = live[0][2].device
        live = [t for t in live if t[2].device == dev]
        # the table is rebuilt whenever the membership changed OR any live entry has no buffers yet (a re-registered key)
        members = tuple((k, id(e)) for k, e, _ in live)
        if members != self.members or any(e["U"] is None for _, e, _ in live):
            rows, bmap, nb = [], [], 0
            for key, e, p in live:
                Co, Ci = key[1], key[2]
                if e["U"] is None:
                    e["U"] = (torch.empty(lib.dfe_wino_weight_floats(Ci, Co), device=dev, dtype=torch.float32),
                              torch.empty(lib.dfe_wino_weight_floats(Co, Ci), device=dev, dtype=torch.float32))
                # forward: conv Ci -> Co, U[Kpad(Co)][Ci]; data gradient: conv Co -> Ci on the transposed, flipped filter
                for tr, (K, C) in enumerate(((Co, Ci), (Ci, Co))):
                    n = int(lib.dfe_wino_transform_blocks(C, K))
                    rows.append([key[0], e["U"][tr].data_ptr(), K, C, tr, nb])
                    bmap.extend([len(rows) - 1] * n)
                    nb += n
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.blockmap = torch.tensor(bmap, dtype=torch.int32).to(dev)
            self.members = members
        for _, e, _ in live:         # exception-safe: nothing is valid until the launch has been enqueued
            e["version"] = -1
        check(lib.dfe_wino_transform_weights_multi(ptr(self.table), ptr(self.blockmap), int(self.blockmap.numel()), stream_ptr()),
              "dfe_wino_transform_weights_multi")
        for _, e, p in live:
            e["version"] = p._version


wino_weights = WinoWeightCache()


def wino_conv3x3(x, w, padding=1, transposed=False, dilation=1, bias=None, slope=1.0, out=None, out_off=0, out2=None, out2_off=0):
    """3x3 stride-1 convolution of x [B,Ci,H,W] on dfe_wino_conv3x3.  ``transposed``: w is the forward filter [Ci,Co,3,3] of a
    convolution whose output gradient is x; the result is its data gradient.  ``dilation`` > 1: a dilated convolution with
    padding = dilation (``padding`` is ignored).  Parameters' transformed filters come from ``wino_weights`` when it holds
    them (dfe_wino_conv3x3_u).

    ``bias`` / ``slope`` / ``out`` / ``out2`` (round 5): the epilogue act(conv + bias), act(v) = v > 0 ? v : slope v, inside the
    kernel's output transform (dfe_wino_conv3x3_u_act), written to channels out_off.. of ``out`` (a contiguous NCHW buffer with
    at least that many channels; default: a fresh tensor) and, if given, to channels out2_off.. of ``out2`` as well."""
    fused = bias is not None or float(slope) != 1.0 or out is not None or out2 is not None
    U = wino_weights.lookup(w, transposed)
    x, w = f32c(x), f32c(w)
    B, Ci, H, W = x.shape
    Co = int(w.shape[1] if transposed else w.shape[0])
    P, d = int(padding), int(dilation)
    lib = get_lib()
    Ho, Wo = (H, W) if d > 1 else (H + 2 * P - 2, W + 2 * P - 2)
    if fused:
        if U is None:
            U = WinoWeightCache.transform_now(w, transposed)
        y = torch.empty(B, Co, Ho, Wo, device=x.device, dtype=torch.float32) if out is None else out
        HWo = Ho * Wo
        for t in (y, out2):
            if t is not None and not (t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape[2:]) == (Ho, Wo) and t.shape[0] == B):
                raise _lib.DfeError("wino_conv3x3: out / out2 must be contiguous fp32 [B, >= Co, Ho, Wo] buffers")
        if y.shape[1] < int(out_off) + Co or (out2 is not None and out2.shape[1] < int(out2_off) + Co):
            raise _lib.DfeError("wino_conv3x3: the output channels do not fit the destination buffer")
        p1 = ctypes.c_void_p(y.data_ptr() + 4 * int(out_off) * HWo)
        p2 = ctypes.c_void_p(out2.data_ptr() + 4 * int(out2_off) * HWo) if out2 is not None else None
        npart = 0 if d > 1 else lib.dfe_wino_scratch_floats(B, Ci, Co, H, W, P) - lib.dfe_wino_weight_floats(Ci, Co)
        part = torch.empty(npart, device=x.device, dtype=torch.float32) if npart > 0 else None
        check(lib.dfe_wino_conv3x3_u_act(ptr(x), ptr(U), ptr(f32c(bias)) if bias is not None else None, float(slope), p1, y.stride(0), p2,
                                         out2.stride(0) if out2 is not None else 0, ptr(part), npart, B, Ci, Co, H, W, P, d, stream_ptr()),
              "dfe_wino_conv3x3_u_act")
        return y
    y = torch.empty(B, Co, Ho, Wo, device=x.device, dtype=torch.float32)
    if U is not None:
        npart = 0 if d > 1 else lib.dfe_wino_scratch_floats(B, Ci, Co, H, W, P) - lib.dfe_wino_weight_floats(Ci, Co)
        part = torch.empty(npart, device=x.device, dtype=torch.float32) if npart > 0 else None
        check(lib.dfe_wino_conv3x3_u(ptr(x), ptr(U), ptr(y), y.stride(0), ptr(part), npart, B, Ci, Co, H, W, P, d, stream_ptr()),
              "dfe_wino_conv3x3_u")
        return y
    nws = lib.dfe_wino_weight_floats(Ci, Co) if d > 1 else lib.dfe_wino_scratch_floats(B, Ci, Co, H, W, P)
    wbuf = torch.empty(nws, device=x.device, dtype=torch.float32)
    if d > 1:
        check(lib.dfe_wino_conv3x3_dilated(ptr(x), ptr(w), ptr(y), y.stride(0), ptr(wbuf), B, Ci, Co, H, W, d, int(bool(transposed)),
                                           stream_ptr()), "dfe_wino_conv3x3_dilated")
        return y
    check(lib.dfe_wino_conv3x3(ptr(x), ptr(w), ptr(y), y.stride(0), ptr(wbuf), nws, B, Ci, Co, H, W, P, int(bool(transposed)),
                               stream_ptr()), "dfe_wino_conv3x3")
    return y


class ConvBiasActFn(torch.autograd.Function):
    """``act(conv2d(x, w, stride 1, padding = dilation) + bias)`` for a 3x3 layer the Winograd kernel takes -- net_utils.conv()
    = Conv2d + LeakyReLU(0.1) (net_utils.py:7-11: FeaturePyramid's stride-1 layers, PWC's context network) -- as ONE launch
    forward (the epilogue inside the output transform) and one autograd node: backward = dfe_bias_act_bwd (gz = gy act'(y),
    bias gradient) + the convolution's data / weight gradients (convs.raw_backward)."""

    @staticmethod
    def forward(ctx, x, w, bias, slope, padding, dilation):
        y = wino_conv3x3(x, w, padding, dilation=dilation, bias=bias, slope=slope)
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (float(slope), int(padding), int(dilation), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        x, w, y = ctx.saved_tensors
        slope, P, d, has_bias = ctx.cfg
        B, C, H, W = y.shape
        if gy.dtype != torch.float32:
            gy = gy.float()
        if not (gy.stride(3) == 1 and gy.stride(2) == W and gy.stride(1) == H * W and gy.stride(0) >= C * H * W):
            gy = gy.contiguous()
        gz = torch.empty_like(y)
        gb = part = None
        if has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(C, device=y.device, dtype=torch.float32)
            part = torch.empty(lib.dfe_bias_act_partials_floats(B, C, H, W), device=y.device, dtype=torch.float32)
        check(lib.dfe_bias_act_bwd(ptr(y), ptr(gy, strided=True), gy.stride(0), ptr(gz), ptr(gb), ptr(part), B, C, H, W, slope,
                                   stream_ptr()), "dfe_bias_act_bwd")
        pad = (d, d) if d > 1 else (P, P)
        gx, gw, _ = convs.raw_backward(gz, x, w, (1, 1), pad, (d, d), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, gb, None, None, None


def conv_bias_act_eligible(x, conv):
    """A Conv2d whose forward the Winograd kernel runs (convs._wino_eligible: 3x3, stride 1, padding 1 or dilated with
    padding = dilation, enough tiles and channels, fp32)."""
    return (WINO_EPILOGUE and conv.bias is not None and conv.groups == 1
            and convs._wino_eligible(x, conv.weight.shape, conv.in_channels, conv.stride, conv.padding, conv.dilation, conv.groups)
            and conv.padding == ((1, 1) if conv.dilation == (1, 1) else conv.dilation))


def conv_bias_act(x, conv, slope):
    return ConvBiasActFn.apply(x, conv.weight, conv.bias, float(slope), int(conv.padding[0]), int(conv.dilation[0]))


def phase_images(t, d):
    """[B,C,H,W] -> [B d^2, C, H/d, W/d]: the d x d phase images x[:, :, p::d, q::d] stacked along the batch (one strided copy).
    A dilation-d 3x3 convolution with padding d is a dense 3x3 convolution with padding 1 on them."""
    B, C, H, W = t.shape
    return t.reshape(B, C, H // d, d, W // d, d).permute(0, 3, 5, 1, 2, 4).reshape(B * d * d, C, H // d, W // d)


def wino_wgrad3x3(x, gy, padding=1, dilation=1):
    """Weight gradient [Co,Ci,3,3] of a 3x3 stride-1 convolution of x [B,Ci,H,W] with ``padding`` in {0, 1} (or dilated with
    padding = dilation: on the phase images) for the output gradient gy [B,Co,Ho,Wo], on dfe_wino_wgrad3x3 (Winograd domain,
    fp32 MFMA).  x and gy may be batch-strided views (channel slices of concatenated buffers)."""
    d = int(dilation)
    if d > 1:
        if x.shape[2] % d or x.shape[3] % d or gy.shape[2:] != x.shape[2:]:
            raise _lib.DfeError("wino_wgrad3x3: a dilated layer needs H and W to be multiples of the dilation and padding = dilation")
        x, gy, padding = phase_images(x, d), phase_images(gy, d), 1
    def dense_chw(t):
        return t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(2) == t.shape[3] and t.stride(1) == t.shape[2] * t.shape[3]
    x = x if dense_chw(x) else f32c(x)
    gy = gy if dense_chw(gy) else f32c(gy)
    B, Ci, H, W = x.shape
    Co, P = int(gy.shape[1]), int(padding)
    lib = get_lib()
    gw = torch.empty(Co, Ci, 3, 3, device=x.device, dtype=torch.float32)
    n = lib.dfe_wino_wgrad_floats(B, Ci, Co, H, W, P)
    if n <= 0:
        raise _lib.DfeError("dfe_wino_wgrad3x3: unsupported shape %s padding %d" % (tuple(x.shape), P))
    ws = torch.empty(n, device=x.device, dtype=torch.float32)
    check(lib.dfe_wino_wgrad3x3(ptr(x, strided=True), x.stride(0), ptr(gy, strided=True), gy.stride(0), ptr(gw), ptr(ws), B, Ci, Co, H, W, P,
                                stream_ptr()), "dfe_wino_wgrad3x3")
    return gw


def _dense_chw(t):
    return t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(2) == t.shape[3] and t.stride(1) == t.shape[2] * t.shape[3]


def sconv_wgrad_supported(x_shape, co, k, stride, padding):
    """Whether dfe_sconv_wgrad takes the layer (csrc/ops_sconv.hip: 3x3 with Ci >= 16, the 3- / 9-channel stems, 5x5 x 16)."""
    B, Ci, H, W = (int(v) for v in x_shape)
    return get_lib().dfe_sconv_wgrad_floats(B, Ci, int(co), H, W, int(k), int(stride), int(padding)) > 0


def sconv_wgrad(x, gy, k, stride, padding):
    """Weight gradient [Co,Ci,k,k] of a strided k x k convolution of x [B,Ci,H,W] for the output gradient gy [B,Co,Ho,Wo], on
    dfe_sconv_wgrad (fp32 MFMA straight from NCHW; depth_model.py:60-95, feature_pyramid.py:7-36, pose_cnn.py:14-36).  x and gy may
    be batch-strided views."""
    x = x if _dense_chw(x) else f32c(x)
    gy = gy if _dense_chw(gy) else f32c(gy)
    B, Ci, H, W = x.shape
    Co, k, s, P = int(gy.shape[1]), int(k), int(stride), int(padding)
    if tuple(gy.shape[2:]) != ((H + 2 * P - k) // s + 1, (W + 2 * P - k) // s + 1) or gy.shape[0] != B:
        raise _lib.DfeError("sconv_wgrad: gy %s does not belong to x %s with k %d stride %d padding %d" % (tuple(gy.shape), tuple(x.shape), k, s, P))
    lib = get_lib()
    n = lib.dfe_sconv_wgrad_floats(B, Ci, Co, H, W, k, s, P)
    if n <= 0:
        raise _lib.DfeError("dfe_sconv_wgrad: unsupported layer %s -> %d, k %d stride %d padding %d" % (tuple(x.shape), Co, k, s, P))
    gw = torch.empty(Co, Ci, k, k, device=x.device, dtype=torch.float32)
    ws = torch.empty(n, device=x.device, dtype=torch.float32)
    check(lib.dfe_sconv_wgrad(ptr(x, strided=True), x.stride(0), ptr(gy, strided=True), gy.stride(0), ptr(gw), ptr(ws), B, Ci, Co, H, W, k, s, P,
                              stream_ptr()), "dfe_sconv_wgrad")
    return gw


class PlaneConvActFn(torch.autograd.Function):
    """act(conv3x3(x, w, pad 1) + bias) on a small plane as one operator (PoseCNN's refinement convolutions,
    pose_cnn.py:43-46, 66-69: Conv2d(12, 12, 3, 1, 1) + ReLU on 2x7 planes): dfe_planeconv_fwd with the epilogue inside;
    backward = dfe_bias_act_bwd (activation mask from the output, bias gradient) + dfe_planeconv_dgrad / _wgrad."""

    @staticmethod
    def forward(ctx, x, w, bias, slope):
        x, w = f32c(x), f32c(w)
        y = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], device=x.device, dtype=torch.float32)
        planeconv_fwd_into(x, w, None if bias is None else f32c(bias), slope, y)
        ctx.save_for_backward(x, w, y)
        ctx.slope, ctx.has_bias = float(slope), bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        x, w, y = ctx.saved_tensors
        B, C, H, W = y.shape
        gy = f32c(gy)
        gz = torch.empty_like(y)
        gb = part = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(C, device=y.device, dtype=torch.float32)
            part = torch.empty(lib.dfe_bias_act_partials_floats(B, C, H, W), device=y.device, dtype=torch.float32)
        check(lib.dfe_bias_act_bwd(ptr(y), ptr(gy), gy.stride(0), ptr(gz), ptr(gb), ptr(part), B, C, H, W, ctx.slope,
                                   stream_ptr()), "dfe_bias_act_bwd")
        gx, gw = planeconv_backward(gz, x, w, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, gb, None


def planeconv_act(x, w, bias, slope):
    return PlaneConvActFn.apply(x, w, bias, float(slope))


class Conv1x1SmallFn(torch.autograd.Function):
    """act(conv1x1(x, w) + bias) on a tiny plane (PoseCNN's pose_conv / refinement 1x1 convolutions on 2x7 planes,
    pose_cnn.py:32,43,48) as one operator: dfe_conv1x1_small_fwd; backward = dfe_bias_act_bwd + dfe_conv1x1_small_bwd."""

    @staticmethod
    def forward(ctx, x, w, bias, slope):
        x, w = f32c(x), f32c(w)
        B, Ci, H, W = x.shape
        Co = int(w.shape[0])
        y = torch.empty(B, Co, H, W, device=x.device, dtype=torch.float32)
        check(get_lib().dfe_conv1x1_small_fwd(ptr(x), ptr(w), ptr(None if bias is None else f32c(bias)), float(slope), ptr(y),
                                              B, Ci, Co, H, W, stream_ptr()), "dfe_conv1x1_small_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.slope, ctx.has_bias = float(slope), bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        x, w, y = ctx.saved_tensors
        B, Co, H, W = y.shape
        Ci = int(x.shape[1])
        gy = f32c(gy)
        gz = torch.empty_like(y)
        gb = part = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(Co, device=y.device, dtype=torch.float32)
            part = torch.empty(lib.dfe_bias_act_partials_floats(B, Co, H, W), device=y.device, dtype=torch.float32)
        check(lib.dfe_bias_act_bwd(ptr(y), ptr(gy), gy.stride(0), ptr(gz), ptr(gb), ptr(part), B, Co, H, W, ctx.slope,
                                   stream_ptr()), "dfe_bias_act_bwd")
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        check(lib.dfe_conv1x1_small_bwd(ptr(gz), ptr(x), ptr(w), ptr(gx), ptr(gw), B, Ci, Co, H, W, stream_ptr()),
              "dfe_conv1x1_small_bwd")
        return gx, gw, gb, None


def conv1x1_small_eligible(x, conv):
    return (PLANECONV_MAX_HW > 0 and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and get_lib().dfe_conv1x1_small_supported(x.shape[0], x.shape[1], conv.out_channels, x.shape[2], x.shape[3]) == 1)


def conv1x1_small(x, conv, slope):
    """``act(conv(x))`` for an nn.Conv2d with a 1x1 kernel on a tiny plane; slope 1 = no activation."""
    return Conv1x1SmallFn.apply(x, conv.weight, conv.bias, float(slope))


class DenseDecodeFn(torch.autograd.Function):
    """One PWC decoder level's DenseNet-style block (pwc_tf.py:113-118 and the same six lines per level)::

        x0 = conv_0(x); x1 = conv_1(x0); x2 = conv_2(cat(x0, x1)); x3 = conv_3(cat(x1, x2)); x4 = conv_4(cat(x2, x3))
        flow = predict_flow(cat(x3, x4))                                      -> (flow, x4)

    with every ``conv_k`` = Conv2d(3x3, pad 1, bias) + LeakyReLU(slope).  The convolutions stay MIOpen calls (bias-free);
    the bias + activation epilogue writes each layer output straight into the channel slices of the two concatenated
    buffers that consume it (``dfe_bias_act_fwd2``), so no ``torch.cat`` copies exist; the backward pass calls
    ``aten::convolution_backward`` layer by layer and the epilogue's backward sums the two consumers' gradient slices in
    place (``dfe_bias_act_bwd2``): no slice copies, no gradient-accumulation adds.  Same arithmetic as the composition."""

    @staticmethod
    def forward(ctx, slope, x, *wb):
        import torch.nn.functional as F
        lib = get_lib()
        x = f32c(x)
        w, b = list(wb[0:12:2]), list(wb[1:12:2])          # conv_0..conv_4, predict_flow
        B, _, H, W = x.shape
        HW = H * W
        co = [int(t.shape[0]) for t in w[:5]]              # 128, 128, 96, 64, 32
        dev = x.device
        cat = [torch.empty(B, co[k] + co[k + 1], H, W, device=dev, dtype=torch.float32) for k in range(4)]   # [x_k | x_k+1]
        st = stream_ptr()

        def epilogue(z, k, d1, d1_off, d2, d2_off):
            c = co[k]
            p1 = ctypes.c_void_p(d1.data_ptr() + 4 * d1_off * HW)
            p2 = ctypes.c_void_p(d2.data_ptr() + 4 * d2_off * HW) if d2 is not None else None
            check(lib.dfe_bias_act_fwd2(ptr(z), ptr(b[k]), p1, d1.stride(0), p2, d2.stride(0) if d2 is not None else 0,
                                        B, c, H, W, slope, st), "dfe_bias_act_fwd2")

        plane = planeconv_eligible(x, w[0])      # levels 6 / 5: the convolution and its epilogue are this build's kernels
        if plane:
            z0 = torch.empty(B, co[0], H, W, device=dev, dtype=torch.float32)
            x4 = torch.empty(B, co[4], H, W, device=dev, dtype=torch.float32)
            planeconv_fwd_into(x, w[0], b[0], slope, z0, 0, cat[0], 0)          # x0
            planeconv_fwd_into(z0, w[1], b[1], slope, cat[0], co[0], cat[1], 0)  # x1
            planeconv_fwd_into(cat[0], w[2], b[2], slope, cat[1], co[1], cat[2], 0)
            planeconv_fwd_into(cat[1], w[3], b[3], slope, cat[2], co[2], cat[3], 0)
            planeconv_fwd_into(cat[2], w[4], b[4], slope, x4, 0, cat[3], co[3])
        else:
            fuse = WINO_EPILOGUE

            def layer(inp, k, d1, d1_off, d2, d2_off):
                """act(conv_k(inp) + b_k) into channels d1_off.. of d1 (None: a fresh tensor, returned) and d2_off.. of d2"""
                if fuse and convs._wino_eligible(inp, w[k].shape, w[k].shape[1], (1, 1), (1, 1), (1, 1)):
                    # round 5: the epilogue inside the Winograd kernel's output transform (one launch, no pass over z)
                    return wino_conv3x3(inp, w[k], 1, bias=b[k], slope=slope, out=d1, out_off=d1_off, out2=d2, out2_off=d2_off)
                z = convs.raw_forward(inp, w[k], 1, 1)
                epilogue(z, k, z if d1 is None else d1, d1_off, d2, d2_off)
                return z
            z0 = layer(x, 0, None, 0, cat[0], 0)                                # x0: conv_1's input + cat0[:, :128]
            layer(z0, 1, cat[0], co[0], cat[1], 0)                              # x1
            layer(cat[0], 2, cat[1], co[1], cat[2], 0)                          # x2
            layer(cat[1], 3, cat[2], co[2], cat[3], 0)                          # x3
            x4 = layer(cat[2], 4, None, 0, cat[3], co[3])                       # x4: returned + cat3[:, 64:]
        if flow_head_eligible(cat[3], w[5], b[5]):
            flow = flow_head_fwd_raw(cat[3], f32c(w[5]), f32c(b[5]))
        else:
            flow = F.conv2d(cat[3], w[5], b[5], 1, 1)
        ctx.save_for_backward(x, z0, *cat, *w)
        ctx.slope, ctx.co, ctx.plane = slope, co, plane
        ctx.set_materialize_grads(False)
        return flow, x4

    @staticmethod
    def backward(ctx, g_flow, g_x4):
        lib = get_lib()
        saved = ctx.saved_tensors
        x, x0, cat, w = saved[0], saved[1], list(saved[2:6]), list(saved[6:12])
        co, slope = ctx.co, ctx.slope
        B, _, H, W = x.shape
        HW = H * W
        dev = x.device
        st = stream_ptr()
        need = ctx.needs_input_grad          # (slope, x, w0, b0, ..., wp, bp)
        nw = lambda k: bool(need[2 + 2 * k])
        nb = lambda k: bool(need[3 + 2 * k])
        def cb(g, inp, wt, want_in, want_w, bias_sizes=None, want_b=False):
            if ctx.plane and not want_b:
                return planeconv_backward(g, inp, wt, want_in, want_w) + (None,)
            return convs.raw_backward(g, inp, wt, 1, 1, 1, want_in, want_w, want_b)

        def epilogue_bwd(k, ysrc, y_off, g1, g1_off, g2, g2_off):
            c = co[k]
            gz = torch.empty(B, c, H, W, device=dev, dtype=torch.float32)
            gb = part = None
            if nb(k):       # only the per-block partial sums now; the level's bias gradients are finished by ONE launch below
                gb = torch.empty(c, device=dev, dtype=torch.float32)
                part = torch.empty(lib.dfe_bias_act_partials_floats(B, c, H, W), device=dev, dtype=torch.float32)
                pending.append((part, gb, c))
            sl = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off * HW)
            check(lib.dfe_bias_act_bwd2(sl(ysrc, y_off), ysrc.stride(0), sl(g1, g1_off), g1.stride(0),
                                        sl(g2, g2_off) if g2 is not None else None, g2.stride(0) if g2 is not None else 0,
                                        ptr(gz), None, ptr(part), B, c, H, W, slope, st), "dfe_bias_act_bwd2")
            return gz, gb

        pending = []

        gw, gbias = [None] * 6, [None] * 6
        if g_flow is not None:
            g_flow = f32c(g_flow)
            if flow_head_eligible(cat[3], w[5], True):
                g3, gw[5], gbias[5] = flow_head_bwd_raw(cat[3], f32c(w[5]), g_flow, nw(5), nb(5))
            else:
                g3, gw[5], gbias[5] = cb(g_flow, cat[3], w[5], True, nw(5), [2], nb(5))       # d/d cat(x3, x4)
        else:
            g3 = torch.zeros_like(cat[3])
        if g_x4 is not None:     # usually a channel slice of the gradient of torch.cat([flow, x4]): read in place
            if g_x4.dtype != torch.float32 or not (g_x4.stride(3) == 1 and g_x4.stride(2) == W and g_x4.stride(1) == HW
                                                   and g_x4.stride(0) >= co[4] * HW):
                g_x4 = f32c(g_x4)
        gz, gbias[4] = epilogue_bwd(4, cat[3], co[3], g3, co[3], g_x4, 0)                 # x4: cat3 slice + the returned copy
        g2, gw[4], _ = cb(gz, cat[2], w[4], True, nw(4))                                  # d/d cat(x2, x3)
        gz, gbias[3] = epilogue_bwd(3, cat[3], 0, g3, 0, g2, co[2])                       # x3
        g1, gw[3], _ = cb(gz, cat[1], w[3], True, nw(3))                                  # d/d cat(x1, x2)
        gz, gbias[2] = epilogue_bwd(2, cat[2], 0, g2, 0, g1, co[1])                       # x2
        g0, gw[2], _ = cb(gz, cat[0], w[2], True, nw(2))                                  # d/d cat(x0, x1)
        gz, gbias[1] = epilogue_bwd(1, cat[1], 0, g1, 0, g0, co[0])                       # x1
        gx0, gw[1], _ = cb(gz, x0, w[1], True, nw(1))                                     # d/d x0 through conv_1
        gz, gbias[0] = epilogue_bwd(0, cat[0], 0, g0, 0, gx0, 0)                          # x0
        if pending:
            n = len(pending)
            parts = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in pending])
            outs = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in pending])
            chans = (ctypes.c_int * n)(*[t[2] for t in pending])
            check(lib.dfe_bias_grad_final_multi(parts, outs, chans, n, B, H, W, st), "dfe_bias_grad_final_multi")
        gx, gw[0], _ = cb(gz, x, w[0], bool(need[1]), nw(0))
        out = [None, gx]
        for k in range(6):
            out += [gw[k], gbias[k]]
        return tuple(out)


def dense_decode(x, convs, predict_flow, slope=0.1):
    """(flow, x4) of one PWC decoder level; ``convs``: the five Conv2d modules, ``predict_flow``: the 3x3 flow head."""
    wb = []
    for c in list(convs) + [predict_flow]:
        wb += [c.weight, c.bias]
    return DenseDecodeFn.apply(float(slope), x, *wb)


# --------------------------------------------------------------------------- grouped BatchNorm (+ residual + ReLU)
class GroupedBatchNormFn(torch.autograd.Function):
    """Training-mode BatchNorm2d over G groups of consecutive samples (statistics per (group, channel); the running
    statistics get the G momentum updates in group order) fused with an optional residual add and ReLU."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, groups, eps, momentum, relu):
        lib = get_lib()
        x = f32c(x)
        N, C, H, W = x.shape
        G = int(groups)
        if N % G:
            raise ValueError("batch %d is not divisible into %d groups" % (N, G))
        Bg = N // G
        residual = f32c(residual) if residual is not None else None
        if residual is not None and residual.shape != x.shape:
            raise ValueError("residual must have the shape of x")
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(G * C, device=dev)
        invstd = torch.empty(G * C, device=dev)
        npart = lib.dfe_bn_partials_floats(G, Bg, C, H, W)
        part = torch.empty(max(npart, 1), device=dev)
        check(lib.dfe_bn_fwd(ptr(x), ptr(residual), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), ptr(y),
                             ptr(mean), ptr(invstd), ptr(part), G, Bg, C, H, W, float(eps), float(momentum), int(relu),
                             stream_ptr()), "dfe_bn_fwd")
        ctx.save_for_backward(x, y if relu else x.new_empty(0), weight if weight is not None else x.new_empty(0), mean, invstd)
        ctx.cfg = (G, Bg, C, H, W, int(relu), residual is not None, weight is not None)
        ctx.mark_non_differentiable(mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        x, y, weight, mean, invstd = ctx.saved_tensors
        G, Bg, C, H, W, relu, has_res, has_w = ctx.cfg
        gy = f32c(gy)
        dev = x.device
        gx = torch.empty_like(x)
        # without a ReLU the residual's gradient is gy itself
        gres = (torch.empty_like(x) if relu else None) if (has_res and ctx.needs_input_grad[1]) else None
        gw = torch.empty(C, device=dev) if (has_w and ctx.needs_input_grad[2]) else None
        gb = torch.empty(C, device=dev) if ctx.needs_input_grad[3] else None
        part = torch.empty(max(lib.dfe_bn_partials_floats(G, Bg, C, H, W), 1), device=dev)
        scratch = torch.empty(2 * G * C, device=dev)
        check(lib.dfe_bn_bwd(ptr(x), ptr(y) if relu else None, ptr(gy), ptr(weight) if has_w else None, ptr(mean), ptr(invstd),
                             ptr(gx), ptr(gres), ptr(gw), ptr(gb), ptr(part), ptr(scratch), G, Bg, C, H, W, relu,
                             stream_ptr()), "dfe_bn_bwd")
        if has_res and ctx.needs_input_grad[1] and not relu:
            gres = gy
        return gx, gres, gw, gb, None, None, None, None, None, None


def grouped_batch_norm(x, weight, bias, running_mean, running_var, groups=1, eps=1e-5, momentum=0.1, residual=None,
                       relu=False):
    return GroupedBatchNormFn.apply(x, residual, weight, bias, running_mean, running_var, int(groups), float(eps),
                                    float(momentum), bool(relu))


# --------------------------------------------------------------------------- disparity head (Conv3x3(C -> 1) + sigmoid)
class DispHeadFn(torch.autograd.Function):
    """sigmoid(conv3x3(p) + bias) on a reflection-padded activation p [B,C,H+2,W+2] -> [B,1,H,W]."""

    @staticmethod
    def forward(ctx, p, weight, bias):
        lib = get_lib()
        p, weight = f32c(p), f32c(weight)
        bias = f32c(bias) if bias is not None else None
        B, C, Hp, Wp = p.shape
        if tuple(weight.shape) != (1, C, 3, 3):
            raise ValueError("weight must be [1,%d,3,3], got %s" % (C, tuple(weight.shape)))
        H, W = Hp - 2, Wp - 2
        out = torch.empty(B, 1, H, W, device=p.device)
        check(lib.dfe_disp_head_fwd(ptr(p), ptr(weight), ptr(bias), ptr(out), B, C, H, W, stream_ptr()), "dfe_disp_head_fwd")
        ctx.save_for_backward(p, weight, out)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = get_lib()
        p, weight, out = ctx.saved_tensors
        B, C, Hp, Wp = p.shape
        H, W = Hp - 2, Wp - 2
        gout = f32c(gout)
        gp = torch.empty_like(p)
        gw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        gb = torch.empty(1, device=p.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        part = torch.empty(lib.dfe_disp_head_partials_floats(B, C, H, W), device=p.device)
        check(lib.dfe_disp_head_bwd(ptr(p), ptr(weight), ptr(out), ptr(gout), ptr(gp), ptr(gw), ptr(gb), ptr(part), B, C, H, W,
                                    stream_ptr()), "dfe_disp_head_bwd")
        return gp, gw, gb


def disp_head(p, weight, bias):
    return DispHeadFn.apply(p, weight, bias)


# --------------------------------------------------------------------------- thin 3x3 convolution (MFMA weight gradient)
def flow_head_fwd_raw(x, weight, bias):
    """Conv2d(C, 2, 3, 1, 1) forward on a contiguous fp32 HIP tensor (no autograd of its own)."""
    lib = get_lib()
    B, C, H, W = x.shape
    out = torch.empty(B, 2, H, W, device=x.device, dtype=torch.float32)
    check(lib.dfe_flow_head_fwd(ptr(x), ptr(weight), ptr(bias), ptr(out), B, C, H, W, stream_ptr()), "dfe_flow_head_fwd")
    return out


def flow_head_bwd_raw(x, weight, gout, want_w=True, want_b=True):
    """(gx, gweight, gbias) of the flow head."""
    lib = get_lib()
    B, C, H, W = x.shape
    gx = torch.empty_like(x)
    gw = torch.empty_like(weight) if want_w else None
    gb = torch.empty(2, device=x.device, dtype=torch.float32) if want_b else None
    part = torch.empty(lib.dfe_flow_head_partials_floats(B, C, H, W), device=x.device, dtype=torch.float32)
    check(lib.dfe_flow_head_bwd(ptr(x), ptr(weight), ptr(gout), ptr(gx), ptr(gw), ptr(gb), ptr(part), B, C, H, W,
                                stream_ptr()), "dfe_flow_head_bwd")
    return gx, gw, gb


def flow_head_eligible(x, weight, bias=None):
    """PWC's predict_flow layers (pwc_tf.py:39-40): two output channels, 3x3, channel count a multiple of 8."""
    return (FLOW_HEAD and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and weight.shape[0] == 2
            and weight.shape[1] == x.shape[1] and x.shape[1] % 8 == 0 and bias is not None)


class FlowHeadFn(torch.autograd.Function):
    """``F.conv2d(x, weight, bias, 1, 1)`` with two output channels on the rolling-window head kernels
    (csrc/ops_disphead.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight, bias = f32c(x), f32c(weight), f32c(bias)
        ctx.save_for_backward(x, weight)
        return flow_head_fwd_raw(x, weight, bias)

    @staticmethod
    def backward(ctx, gout):
        x, weight = ctx.saved_tensors
        gx, gw, gb = flow_head_bwd_raw(x, weight, f32c(gout), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return (gx if ctx.needs_input_grad[0] else None), gw, gb


class ThinConv3x3Fn(torch.autograd.Function):
    """Valid 3x3 convolution of a pre-padded activation, bias-free, for the decoder's thin full-resolution layers (at most 32
    output channels at >= 64x208 pixels).  Forward and data gradient: convs.raw_forward / raw_backward (the half-tile Winograd
    kernel since round 4; round 3's dfe_thin_conv3x3 was removed in round 5).  Weight gradient -- a (Co x 9 Ci) contraction over
    B*H*W pixels: the Winograd-domain kernel from 32 input channels up, dfe_wgrad3x3_fwd (fp32 MFMA straight from NCHW) for 16 -> 16."""

    @staticmethod
    def forward(ctx, p, weight):
        ctx.save_for_backward(p, weight)
        return convs.raw_forward(p, weight)

    @staticmethod
    def backward(ctx, gy):
        lib = get_lib()
        p, weight = ctx.saved_tensors
        gy = f32c(gy)
        B, Ci, Hp, Wp = p.shape
        Co, H, W = weight.shape[0], Hp - 2, Wp - 2
        gp = gw = None
        if ctx.needs_input_grad[0]:
            gp = convs.raw_backward(gy, p, weight, 1, 0, 1, True, False)[0]
        if ctx.needs_input_grad[1]:
            if Ci >= 32 and convs.WINO_WGRAD:
                # round 5: the Winograd-domain kernel is faster wherever there are >= 32 input channels (96 -> 32 at 130x418 x 12:
                # 229 against 374 us; 64 -> 32 at 66x210: 60 against 70; 32 -> 16 at 130x418: 102 against ~124); the 16 -> 16
                # layer at 258x834 stays here (133 against 304)
                gw = wino_wgrad3x3(p, gy, 0)
            else:
                gw = torch.empty_like(weight)
                part = torch.empty(lib.dfe_wgrad3x3_partials_floats(B, Ci, Co, H, W), device=p.device)
                check(lib.dfe_wgrad3x3_fwd(ptr(p), ptr(gy), ptr(gw), ptr(part), B, Ci, Co, H, W, stream_ptr()), "dfe_wgrad3x3_fwd")
        return gp, gw


def thin_conv3x3_eligible(p, weight):
    """Layers the MFMA weight gradient is measured to win on: <= 32 output channels at >= 64x208 pixels (the decoder's
    last stages), channel counts multiples of 16, width a multiple of 16."""
    Co, Ci = weight.shape[0], weight.shape[1]
    H, W = p.shape[2] - 2, p.shape[3] - 2
    return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and tuple(weight.shape[2:]) == (3, 3)
            and Co <= 32 and Co % 16 == 0 and Ci % 16 == 0 and W % 16 == 0 and H * W >= 64 * 208)


def conv3x3_valid(p, weight):
    """3x3 convolution without padding or bias of a pre-padded activation (the decoder's ConvBlock convolutions)."""
    if thin_conv3x3_eligible(p, weight):
        return ThinConv3x3Fn.apply(p, weight)
    return convs.conv2d(p, weight)
