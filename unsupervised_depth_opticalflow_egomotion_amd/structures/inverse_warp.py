"""Reference-signature shims for core/networks/structures/inverse_warp.py (HIP-backed).

pose_vec2mat :172-187, euler2mat :110-145, inverse_warp2 :263-303, calculate_rigid_flow
:311-342, compute_essential_matrix :354-364, compute_projection_matrix :366-374.
No module-global pixel grid (the reference's racy cache, :6-18): kernels derive pixel
coordinates from thread indices.

Legacy signatures nothing on the path calls ("keep, with a Python fallback"): ``inverse_warp``
:190-224, ``pixel2cam`` :30-45, ``cam2pixel`` :80-107, ``cam2pixel_change_shape`` :47-78, ``cam2pixel2`` :227-260,
``quat2mat`` :148-169, ``rotation_mode='quat'``, ``meshgrid`` :305-309, ``skewsymmetric`` :344-352.  They are plain tensor
expressions on whatever device their arguments live on; ``inverse_warp`` in Euler mode goes through the HIP operators
(rigid flow + bilinear warp); golden G12 pins all of them against the reference."""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops


def check_sizes(inp, name, expected):
    cond = [inp.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            cond.append(inp.size(i) == int(size))
    assert all(cond), "wrong size for {}, expected {}, got  {}".format(name, "x".join(expected), list(inp.size()))


def quat2mat(quat):
    """First three quaternion coefficients [B,3] (w is fixed to 1 before normalising) -> rotation [B,3,3] (:148-169)."""
    q = torch.cat([torch.ones_like(quat[:, :1]), quat], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    rows = [w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2]
    return torch.stack(rows, dim=1).reshape(quat.size(0), 3, 3)


def pose_vec2mat(vec, rotation_mode="euler"):
    """(tx,ty,tz,rx,ry,rz) [B,6] -> [R|t] [B,3,4], R = Rx Ry Rz ('euler', the HIP operator) or quat2mat ('quat')."""
    if rotation_mode == "quat":
        return torch.cat([quat2mat(vec[:, 3:]), vec[:, :3].unsqueeze(-1)], dim=2)
    if rotation_mode != "euler":
        raise ValueError("rotation_mode must be 'euler' or 'quat', got %r" % (rotation_mode,))
    return ops.PoseMatsFn.apply(vec)[0]


def euler2mat(angle):
    vec = torch.cat([torch.zeros_like(angle), angle], dim=1)
    return ops.PoseMatsFn.apply(vec)[0][:, :, :3]


def skewsymmetric(translation):
    """[t]x [B,3,3] of t [B,3] (:344-352)."""
    x, y, z = translation[:, 0:1], translation[:, 1:2], translation[:, 2:3]
    o = torch.zeros_like(x)
    return torch.cat([o, -z, y, z, o, -x, -y, x, o], dim=1).view(translation.size(0), 3, 3)


def compute_essential_matrix(vec, rotation_mode="euler"):
    """E = [t]x R  [B,3,3]."""
    if rotation_mode == "quat":
        return skewsymmetric(vec[:, :3]) @ quat2mat(vec[:, 3:])
    if rotation_mode != "euler":
        raise ValueError("rotation_mode must be 'euler' or 'quat', got %r" % (rotation_mode,))
    return ops.PoseMatsFn.apply(vec)[1]


def compute_projection_matrix(vec, K, rotation_mode="euler"):
    b = K.shape[0]
    iden = torch.cat([torch.eye(3), torch.zeros([3, 1])], -1).unsqueeze(0).repeat(b, 1, 1).to(K.device)
    return K.bmm(iden), K @ pose_vec2mat(vec, rotation_mode)


def inverse_warp2(img, depth, ref_depth, pose, intrinsics, padding_mode="zeros", align_corners=None):
    """Inverse-warp a source image into the target view.

    Returns (projected_img [B,3,H,W], valid_mask [B,1,H,W], projected_depth [B,1,H,W],
    computed_depth [B,1,H,W]).  Differentiable wrt depth, ref_depth and pose."""
    check_sizes(img, "img", "B3HW")
    check_sizes(depth, "depth", "B1HW")
    check_sizes(ref_depth, "ref_depth", "B1HW")
    check_sizes(pose, "pose", "B6")
    check_sizes(intrinsics, "intrinsics", "B33")
    if padding_mode != "zeros":
        raise NotImplementedError("only padding_mode='zeros' is on the hot path")
    ac = ops.get_align_corners() if align_corners is None else bool(align_corners)
    return ops.InverseWarp2Fn.apply(img, depth, ref_depth, pose, intrinsics, int(ac))


def calculate_rigid_flow(depth, pose, intrinsics):
    """Rigid flow [B,2,H,W] of depth [B,1,H,W] under pose [B,6]: (X/Z, Y/Z) - (x, y)."""
    return ops.RigidFlowFn.apply(depth, pose, intrinsics)


# ------------------------------------------------------------------ legacy signatures (off the path; see the module docstring)
def meshgrid(h, w):
    """[2,h,w] integer pixel coordinates (x, y) (:305-309)."""
    xx, yy = np.meshgrid(np.arange(0, w), np.arange(0, h))
    return torch.from_numpy(np.transpose(np.stack([xx, yy], axis=-1), [2, 0, 1]))


def _homogeneous_pixels(depth):
    b, h, w = depth.size()
    ys = torch.arange(0, h, device=depth.device).view(1, h, 1).expand(1, h, w).type_as(depth)
    xs = torch.arange(0, w, device=depth.device).view(1, 1, w).expand(1, h, w).type_as(depth)
    return torch.stack((xs, ys, torch.ones_like(xs)), dim=1)        # [1,3,H,W]; no module-global cache


def pixel2cam(depth, intrinsics_inv):
    """depth [B,H,W], K^-1 [B,3,3] -> camera-frame points [B,3,H,W] (:30-45)."""
    b, h, w = depth.size()
    pix = _homogeneous_pixels(depth).expand(b, 3, h, w).reshape(b, 3, -1)
    return (intrinsics_inv @ pix).reshape(b, 3, h, w) * depth.unsqueeze(1)


def _project_flat(cam_coords, proj_c2p_rot, proj_c2p_tr):
    b, _, h, w = cam_coords.size()
    p = cam_coords.reshape(b, 3, -1)
    if proj_c2p_rot is not None:
        p = proj_c2p_rot @ p
    if proj_c2p_tr is not None:
        p = p + proj_c2p_tr
    z = p[:, 2].clamp(min=1e-3)
    return p[:, 0] / z, p[:, 1] / z, z


def cam2pixel_change_shape(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode=None):
    """Pixel coordinates [B,2,H,W], not normalised (:47-78)."""
    b, _, h, w = cam_coords.size()
    u, v, _ = _project_flat(cam_coords, proj_c2p_rot, proj_c2p_tr)
    return torch.cat([u.reshape(b, h, w).unsqueeze(1), v.reshape(b, h, w).unsqueeze(1)], dim=1)


def cam2pixel(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode=None):
    """Normalised [-1,1] sampling grid [B,H,W,2] (:80-107)."""
    b, _, h, w = cam_coords.size()
    u, v, _ = _project_flat(cam_coords, proj_c2p_rot, proj_c2p_tr)
    return torch.stack([2 * u / (w - 1) - 1, 2 * v / (h - 1) - 1], dim=2).reshape(b, h, w, 2)


def cam2pixel2(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode):
    """Normalised grid [B,H,W,2] with out-of-range coordinates moved to 2 ('zeros'), and the depth Z [B,1,H,W] (:227-260).
    On the path this is fused into the HIP projection (csrc/dfe_camera.h)."""
    b, _, h, w = cam_coords.size()
    u, v, z = _project_flat(cam_coords, proj_c2p_rot, proj_c2p_tr)
    xn, yn = 2 * u / (w - 1) - 1, 2 * v / (h - 1) - 1
    if padding_mode == "zeros":
        xn = torch.where(((xn > 1) | (xn < -1)).detach(), torch.full_like(xn, 2.0), xn)
        yn = torch.where(((yn > 1) | (yn < -1)).detach(), torch.full_like(yn, 2.0), yn)
    return torch.stack([xn, yn], dim=2).reshape(b, h, w, 2), z.reshape(b, 1, h, w)


def inverse_warp(img, depth, pose, intrinsics, rotation_mode="euler", padding_mode="zeros", align_corners=None):
    """The first-generation warp (:190-224): depth is [B,H,W], no out-of-range rule, returns (projected_img,
    valid_points [B,H,W] bool).  Euler mode = the HIP operators: the rigid flow of (depth, pose, K) followed by the bilinear
    warp at x + flow (``warp_flow`` normalises its grid exactly as ``cam2pixel`` does, net_utils.py:34-41).  Quaternion mode and
    non-zero padding modes are tensor expressions + ``grid_sample``."""
    check_sizes(img, "img", "B3HW")
    check_sizes(depth, "depth", "BHW")
    check_sizes(pose, "pose", "B6")
    check_sizes(intrinsics, "intrinsics", "B33")
    ac = ops.get_align_corners() if align_corners is None else bool(align_corners)
    b, _, h, w = img.size()
    if rotation_mode == "euler" and padding_mode == "zeros" and img.is_cuda:
        flow = calculate_rigid_flow(depth.unsqueeze(1), pose, intrinsics)
        projected = ops.warp_flow(img, flow, use_mask=False, align_corners=ac)
        xs = torch.arange(w, device=img.device, dtype=flow.dtype).view(1, 1, w)
        ys = torch.arange(h, device=img.device, dtype=flow.dtype).view(1, h, 1)
        xn, yn = 2 * (flow[:, 0] + xs) / (w - 1) - 1, 2 * (flow[:, 1] + ys) / (h - 1) - 1
        return projected, torch.maximum(xn.abs(), yn.abs()) <= 1
    cam = pixel2cam(depth, intrinsics.inverse())
    proj = intrinsics @ pose_vec2mat(pose, rotation_mode)
    grid = cam2pixel(cam, proj[:, :, :3], proj[:, :, -1:], padding_mode)
    projected = F.grid_sample(img, grid, padding_mode=padding_mode, align_corners=ac)
    return projected, grid.abs().max(dim=-1)[0] <= 1
