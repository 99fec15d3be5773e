"""Reference-signature shims for core/networks/structures/inverse_warp.py (HIP-backed).

pose_vec2mat :172-187, euler2mat :110-145, inverse_warp2 :263-303, calculate_rigid_flow
:311-342, compute_essential_matrix :354-364, compute_projection_matrix :366-374.
No module-global pixel grid (the reference's racy cache, :6-18): kernels derive pixel
coordinates from thread indices."""
import torch

from .. import ops


def check_sizes(inp, name, expected):
    cond = [inp.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            cond.append(inp.size(i) == int(size))
    assert all(cond), "wrong size for {}, expected {}, got  {}".format(name, "x".join(expected), list(inp.size()))


def pose_vec2mat(vec, rotation_mode="euler"):
    """(tx,ty,tz,rx,ry,rz) [B,6] -> [R|t] [B,3,4], R = Rx Ry Rz."""
    if rotation_mode != "euler":
        raise NotImplementedError("only rotation_mode='euler' is on the hot path")
    return ops.PoseMatsFn.apply(vec)[0]


def euler2mat(angle):
    vec = torch.cat([torch.zeros_like(angle), angle], dim=1)
    return ops.PoseMatsFn.apply(vec)[0][:, :, :3]


def compute_essential_matrix(vec, rotation_mode="euler"):
    """E = [t]x R  [B,3,3]."""
    if rotation_mode != "euler":
        raise NotImplementedError("only rotation_mode='euler' is on the hot path")
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
