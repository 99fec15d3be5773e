"""ORACLE -- test infrastructure only, never the product path.

CPU (PyTorch fp32, ATen) restatement of the reference's photometric-warping loss
stack.  The arithmetic of this path lives in ATen (``grid_sample``, ``avg_pool2d``,
``interpolate``, ``softmax``, ``bmm``, ``inverse``), so the oracle calls the same
ATen operators in the same association order as the reference and is therefore a
torch program rather than C/numpy.  It is pinned against golden vectors captured
from the real reference in the build container (``tests/golden/make_golden.py``);
the reference itself has no tests for this path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``unsupervised_depth_opticalflow_egomotion_amd``)
never does and fails loudly when its HIP library is missing.

Every function cites the reference lines it follows (paths relative to
/root/reference).  ``align_corners`` is explicit everywhere: the reference leaves
it to the installed torch (SURVEY.md "three things", item 2); ``False`` is what
torch 2.10 does.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- helpers


def _pixel_grid(b, h, w, like):
    """[B,2,H,W] float grid, channel 0 = x, 1 = y (net_utils.py:28-32)."""
    xs = torch.arange(0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    ys = torch.arange(0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    return torch.cat((xs, ys), 1).float().to(like.device)


def _gs(img, grid, align_corners):
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


# --------------------------------------------------------------------------- a2 warp_flow


def _flow_grid(x, flow):
    """Normalised sampling grid of warp_flow (net_utils.py:28-44)."""
    b, _, h, w = x.shape
    pos = _pixel_grid(b, h, w, x) + flow
    gx = 2.0 * pos[:, 0] / max(w - 1, 1) - 1.0
    gy = 2.0 * pos[:, 1] / max(h - 1, 1) - 1.0
    return torch.stack((gx, gy), dim=3)


def warp_flow(x, flow, use_mask=False, align_corners=False):
    """Backward bilinear warp of ``x`` by ``flow`` (net_utils.py:16-54).

    gx = 2*(x+u)/max(W-1,1) - 1 (:42-43); zeros padding; with ``use_mask`` the output is
    zeroed wherever the in-bounds bilinear weight sum is < 0.9999 (:47-52)."""
    b, c, h, w = x.shape
    if (b, 2, h, w) != tuple(flow.shape):
        raise ValueError("the shape of grid {0} is not equal to the shape of flow {1}.".format(
            torch.Size((b, 2, h, w)), flow.shape))
    grid = _flow_grid(x, flow)
    out = _gs(x, grid, align_corners)
    if not use_mask:
        return out
    cover = _gs(torch.ones_like(x), grid, align_corners).detach()
    keep = torch.where(cover < 0.9999, torch.zeros_like(cover), cover)
    keep = torch.where(keep > 0, torch.ones_like(keep), keep)
    return out * keep


# --------------------------------------------------------------------------- a3 pose


class trig:
    """How ``euler2mat`` evaluates cos / sin (test-only switch; default = the reference's own call).

    ``host``: ``torch.cos`` / ``torch.sin`` of the machine the oracle runs on -- what the reference does
    (inverse_warp.py:122-139).  That is a vendor libm (MKL VML / SLEEF, <= 1 ulp, not reproducible across hosts).
    ``cr``: the correctly rounded value (float64 evaluation rounded once to fp32) -- what the HIP kernels compute
    (``cos_cr`` / ``sin_cr``), i.e. the statement "bit-exact given R" of DESIGN.md section 2.
    ``ulp_shift``: optional [6] integer offsets added, in units of fp32 ulps, to (cos rx, sin rx, cos ry, sin ry, cos rz,
    sin rz) after evaluation: the parity tests use it to measure how far a <= 1-ulp libm can move a decision.
    Gradients are those of torch.cos / torch.sin in every mode."""
    mode = "host"
    ulp_shift = None

    def __init__(self, mode="host", ulp_shift=None):
        self.new = (mode, ulp_shift)

    def __enter__(self):
        self.old = (trig.mode, trig.ulp_shift)
        trig.mode, trig.ulp_shift = self.new
        return self

    def __exit__(self, *exc):
        trig.mode, trig.ulp_shift = self.old
        return False

    @staticmethod
    def _eval(fn, x, slot):
        y = fn(x)
        v = y.detach()
        if trig.mode == "cr":
            v = fn(x.detach().double()).float()
        elif trig.mode != "host":
            raise ValueError(trig.mode)
        if trig.ulp_shift is not None and int(trig.ulp_shift[slot]) != 0:
            k = int(trig.ulp_shift[slot])
            tgt = torch.full_like(v, float("inf") if k > 0 else float("-inf"))
            for _ in range(abs(k)):
                v = torch.nextafter(v, tgt)
        return y + (v - y.detach())


def euler2mat(angle):
    """R = Rx(rx) @ Ry(ry) @ Rz(rz) (inverse_warp.py:110-145)."""
    n = angle.shape[0]
    rx, ry, rz = angle[:, 0], angle[:, 1], angle[:, 2]
    zero = rz.detach() * 0
    one = zero.detach() + 1
    cz, sz = trig._eval(torch.cos, rz, 4), trig._eval(torch.sin, rz, 5)
    cy, sy = trig._eval(torch.cos, ry, 2), trig._eval(torch.sin, ry, 3)
    cx, sx = trig._eval(torch.cos, rx, 0), trig._eval(torch.sin, rx, 1)
    zm = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).reshape(n, 3, 3)
    ym = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).reshape(n, 3, 3)
    xm = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).reshape(n, 3, 3)
    return xm @ ym @ zm


def pose_vec2mat(vec, rotation_mode="euler"):
    """(tx,ty,tz,rx,ry,rz) -> [R|t] [B,3,4] (inverse_warp.py:172-187)."""
    if rotation_mode != "euler":
        raise ValueError("only the euler parameterisation is on the hot path")
    return torch.cat([euler2mat(vec[:, 3:]), vec[:, :3].unsqueeze(-1)], dim=2)


def compute_essential_matrix(vec):
    """E = [t]x @ R (inverse_warp.py:344-364)."""
    t = vec[:, :3]
    n = t.shape[0]
    x, y, z = t[:, 0:1], t[:, 1:2], t[:, 2:3]
    zero = torch.zeros_like(x)
    skew = torch.cat([zero, -z, y, z, zero, -x, -y, x, zero], dim=1).view(n, 3, 3)
    return skew.bmm(euler2mat(vec[:, 3:]))


# --------------------------------------------------------------------------- a4-a7 rigid projection


def _fma(a, b, c):
    """fp32 fused multiply-add a*b + c with one rounding (the product of two fp32 numbers is exact in fp64)."""
    return (a.double() * b.double() + c.double()).float()


def _bmm3(a, b):
    """[B,3,3] @ [B,3,N] with the accumulation order of the reference run that produced the golden vectors.

    The reference writes ``a @ b`` (inverse_warp.py:41,238; model_geometry.py:393).  For N in the thousands that is
    MKL sgemm, whose order of the three products is not part of torch's contract and differs between hosts (measured:
    2e-6 in the synthesised views between the build container and the GPU box's host).  In the container that
    generated tests/golden it is fma(a2, b2, fma(a1, b1, a0 * b0)) -- probed bit for bit on 1e6 elements -- and that
    order is stated here explicitly so that the oracle is the same function on every host
    (tests/test_oracle_golden.py holds it to the captured reference outputs with array_equal)."""
    rows = []
    for i in range(3):
        acc = a[:, i, 0:1] * b[:, 0]
        acc = _fma(a[:, i, 1:2], b[:, 1], acc)
        acc = _fma(a[:, i, 2:3], b[:, 2], acc)
        rows.append(acc)
    return torch.stack(rows, dim=1)


def _inverse3(k):
    """``intrinsics.inverse()`` (inverse_warp.py:284,329) as LAPACK getrf/getrs evaluate it on a 3x3 fp32 matrix
    stored row-major (ATen factors the transposed storage): partial pivoting, first column scaled by the reciprocal
    of the pivot, second column divided, FMA Schur updates, reciprocal diagonal in the triangular solve.  Stated
    explicitly for the same reason as ``_bmm3``; bit-identical to ``torch.inverse`` in the build container for
    intrinsics that need no row exchange (|fx| >= |cx|, |fy| >= |cy|), checked in tests/test_oracle_golden.py.
    Intrinsics are constants of the path (no gradient flows into K)."""
    import numpy as np
    f32 = np.float32
    out = np.zeros(tuple(k.shape), f32)
    kk = k.detach().cpu().numpy().astype(f32)

    def fma(a, b, c):
        return f32(np.float64(a) * np.float64(b) + np.float64(c))
    for n in range(kk.shape[0]):
        m = kk[n].T.copy()
        perm = [0, 1, 2]
        for j in range(2):
            p = j + int(np.argmax(np.abs(m[j:, j])))
            if p != j:
                m[[j, p]] = m[[p, j]]
                perm[j], perm[p] = perm[p], perm[j]
            if j == 0:
                r = f32(1) / m[0, 0]
                m[1, 0] = m[1, 0] * r
                m[2, 0] = m[2, 0] * r
            else:
                m[2, 1] = m[2, 1] / m[1, 1]
            for i in range(j + 1, 3):
                for c in range(j + 1, 3):
                    m[i, c] = fma(-m[i, j], m[j, c], m[i, c])
        y = np.zeros((3, 3), f32)
        for c in range(3):
            for i in range(3):
                acc = f32(1.0 if i == c else 0.0)
                for q in range(i):
                    acc = f32(acc - f32(m[q, i] * y[q, c]))
                y[i, c] = f32(acc * (f32(1) / m[i, i]))
        wm = np.zeros((3, 3), f32)
        for c in range(3):
            for i in (2, 1, 0):
                acc = y[i, c]
                for q in range(i + 1, 3):
                    acc = f32(acc - f32(m[q, i] * wm[q, c]))
                wm[i, c] = acc
        for i in range(3):
            out[n, perm[i]] = wm[i]
    return torch.from_numpy(out).to(k.device)


def _project(depth, pose, intrinsics):
    """cam = depth * K^-1 (x,y,1); p = (K R) cam + K t (inverse_warp.py:30-45,284-292,329-338).
    Returns X, Y, Z(clamped at 1e-3) as [B, H*W]."""
    b, _, h, w = depth.shape
    xs = torch.arange(0, w).view(1, 1, w).expand(1, h, w).type_as(depth)
    ys = torch.arange(0, h).view(1, h, 1).expand(1, h, w).type_as(depth)
    pix = torch.stack((xs, ys, torch.ones(1, h, w).type_as(depth)), dim=1)
    pix = pix.expand(b, 3, h, w).reshape(b, 3, -1)
    cam = _bmm3(_inverse3(intrinsics), pix).reshape(b, 3, h, w) * depth.squeeze(1).unsqueeze(1)
    proj = intrinsics @ pose_vec2mat(pose)          # [B,3,3] @ [B,3,4]: ATen's own scalar loop (no BLAS), host independent
    p = _bmm3(proj[:, :, :3], cam.reshape(b, 3, -1)) + proj[:, :, -1:]
    return p[:, 0], p[:, 1], p[:, 2].clamp(min=1e-3)


def _rigid_grid(depth, pose, intrinsics, padding_mode="zeros"):
    """Normalised sampling grid [B,H,W,2] of cam2pixel2 (inverse_warp.py:227-260) and the clamped depth Z."""
    b, _, h, w = depth.shape
    X, Y, Z = _project(depth, pose, intrinsics)
    xn = 2 * (X / Z) / (w - 1) - 1
    yn = 2 * (Y / Z) / (h - 1) - 1
    if padding_mode == "zeros":
        xn = torch.where(((xn > 1) + (xn < -1)).detach(), torch.full_like(xn, 2.0), xn)
        yn = torch.where(((yn > 1) + (yn < -1)).detach(), torch.full_like(yn, 2.0), yn)
    return torch.stack([xn, yn], dim=2).reshape(b, h, w, 2), Z


def inverse_warp2(img, depth, ref_depth, pose, intrinsics, padding_mode="zeros", align_corners=False):
    """Rigid view synthesis (inverse_warp.py:263-303 with cam2pixel2 :227-260)."""
    assert img.dim() == 4 and img.shape[1] == 3, "wrong size for img"
    assert depth.dim() == 4 and depth.shape[1] == 1, "wrong size for depth"
    assert ref_depth.dim() == 4 and ref_depth.shape[1] == 1, "wrong size for ref_depth"
    assert pose.dim() == 2 and pose.shape[1] == 6, "wrong size for pose"
    assert intrinsics.dim() == 3 and tuple(intrinsics.shape[1:]) == (3, 3), "wrong size for intrinsics"
    b, _, h, w = img.shape
    grid, Z = _rigid_grid(depth, pose, intrinsics, padding_mode)
    projected_img = F.grid_sample(img, grid, mode="bilinear", padding_mode=padding_mode,
                                  align_corners=align_corners)
    valid = (grid.abs().max(dim=-1)[0] <= 1).unsqueeze(1).float()
    projected_depth = F.grid_sample(ref_depth, grid, mode="bilinear", padding_mode=padding_mode,
                                    align_corners=align_corners).clamp(min=1e-3)
    return projected_img, valid, projected_depth, Z.reshape(b, 1, h, w)


def calculate_rigid_flow(depth, pose, intrinsics):
    """(X/Z, Y/Z) - (x, y) (inverse_warp.py:311-342, cam2pixel_change_shape :47-78)."""
    b, _, h, w = depth.shape
    X, Y, Z = _project(depth, pose, intrinsics)
    px = torch.cat([(X / Z).reshape(b, h, w).unsqueeze(1), (Y / Z).reshape(b, h, w).unsqueeze(1)], dim=1)
    return px - _pixel_grid(b, h, w, depth)


# --------------------------------------------------------------------------- a8 SSIM, a20 correlation


def SSIM(x, y):
    """3x3 zero-padded box SSIM, divisor always 9 (pytorch_ssim/ssim.py:4-19)."""
    c1, c2 = 0.01 ** 2, 0.03 ** 2

    def box(t):
        return F.avg_pool2d(t, 3, 1, padding=1)
    mu_x, mu_y = box(x), box(y)
    var_x = box(x ** 2) - mu_x ** 2
    var_y = box(y ** 2) - mu_y ** 2
    cov = box(x * y) - mu_x * mu_y
    num = (2 * mu_x * mu_y + c1) * (2 * cov + c2)
    den = (mu_x ** 2 + mu_y ** 2 + c1) * (var_x + var_y + c2)
    return num / den


def corr_naive(input1, input2, d=4):
    """81-tap channel-mean cost volume, dy outer / dx inner (pwc_tf.py:97-106)."""
    assert input1.shape == input2.shape
    h, w = input1.shape[2:4]
    padded = F.pad(input2, (d, d, d, d), value=0)
    planes = []
    for dy in range(2 * d + 1):
        for dx in range(2 * d + 1):
            planes.append((input1 * padded[:, :, dy:dy + h, dx:dx + w]).mean(1).unsqueeze(1))
    return torch.cat(planes, 1)


# --------------------------------------------------------------------------- loss-stack composition


def _l2norm(flow):
    """||flow||_2 over the channel + 1e-12 (model_geometry.py:46-52)."""
    return torch.norm(flow, p=2, dim=1).unsqueeze(1) + 1e-12


def _unit(flow):
    return flow / _l2norm(flow).repeat(1, 2, 1, 1)


class occ_exp:
    """How the occlusion weights' 2-way softmax (model_geometry.py:119-130) evaluates its exponential (test-only switch;
    default = the reference's own call).

    ``host``: ``F.softmax`` of the machine the oracle runs on -- what the reference does; its ``exp`` is a vendor routine
    (<= 1 ulp, not reproducible across hosts).
    ``cr``: the softmax written out -- m = max(dl, dr); e_i = exp(d_i - m); w_i = 1 - e_i / (e_l + e_r), fp32 operations in
    that order -- with the exponential CORRECTLY ROUNDED (float64 evaluation rounded once to fp32).  This is the function the
    HIP kernels decide the occlusion bits on (dfe_device.h ``occ_exp``): in this mode the bits are compared for EQUALITY,
    no noise floor.  The weights themselves differ from ``host`` by <= 2 ulp; the hard decisions only where |w - 0.48| is inside
    that."""
    mode = "host"

    def __init__(self, mode="host"):
        self.new = mode

    def __enter__(self):
        self.old = occ_exp.mode
        occ_exp.mode = self.new
        return self

    def __exit__(self, *exc):
        occ_exp.mode = self.old
        return False

    @staticmethod
    def weights(dl, dr):
        """1 - softmax([dl, dr]) over the channel dimension, [B, 2, H, W]."""
        if occ_exp.mode == "host":
            return 1 - F.softmax(torch.cat((dl, dr), 1), 1)
        if occ_exp.mode != "cr":
            raise ValueError(occ_exp.mode)
        m = torch.maximum(dl, dr)
        el, er = torch.exp((dl - m).double()).float(), torch.exp((dr - m).double()).float()
        ssum = el + er
        w = torch.cat((1 - el / ssum, 1 - er / ssum), 1)
        soft = 1 - F.softmax(torch.cat((dl, dr), 1), 1)        # (gradients, where a caller wants them: the reference's graph)
        return soft + (w - soft).detach()


class GeomLossOracle:
    """The ``compute_*`` / ``fusion_*`` methods of the reference models, stateless.

    Method names follow ``Model_geometry`` (model_geometry.py) so the parity tests read
    like calls into the reference."""

    def __init__(self, num_scales=3, flow_consist_alpha=0.01, flow_consist_beta=0.5,
                 rigid_thres=0.5, inlier_thres=0.1, align_corners=False):
        self.num_scales = num_scales
        self.flow_consist_alpha = flow_consist_alpha
        self.flow_consist_beta = flow_consist_beta
        self.rigid_thres = rigid_thres
        self.inlier_thres = inlier_thres
        self.align_corners = align_corners

    # a1 ---------------------------------------------------------------
    def generate_img_pyramid(self, img, num_pyramid):
        """Bilinear (align_corners=False) resize to int(H/2^s) x int(W/2^s) (model_geometry.py:65-72)."""
        h, w = img.shape[2], img.shape[3]
        return [F.interpolate(img, (int(h / (2 ** s)), int(w / (2 ** s))), mode="bilinear", align_corners=False)
                for s in range(num_pyramid)]

    def generate_img_pyramid_avgpool(self, img, num_pyramid):
        """Model_flow's box-mean pyramid (model_flow.py:58-64)."""
        h, w = img.shape[2], img.shape[3]
        return [F.adaptive_avg_pool2d(img, [int(h / (2 ** s)), int(w / (2 ** s))]).data
                for s in range(num_pyramid)]

    def warp_flow_pyramid(self, img_pyramid, flow_pyramid):
        """zip() truncates to the shorter list (model_geometry.py:74-78)."""
        return [warp_flow(i, f, use_mask=True, align_corners=self.align_corners)
                for i, f in zip(img_pyramid, flow_pyramid)]

    # a6 caller ----------------------------------------------------------
    def reconstruction(self, ref_img, intrinsics, depth, depth_ref, pose):
        """Area-downsample the source, scale K rows 0-1, inverse_warp2 (model_geometry.py:80-103)."""
        outs = ([], [], [], [])
        for s in range(self.num_scales):
            b, _, h, w = depth[s].shape
            src = F.interpolate(ref_img, (h, w), mode="area")
            down = ref_img.size(2) / h
            k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
            res = inverse_warp2(src, depth[s], depth_ref[s], pose, k_s, align_corners=self.align_corners)
            for lst, r in zip(outs, res):
                lst.append(r)
        return outs

    # a11 -----------------------------------------------------------------
    def compute_occ_weight(self, from_l, tgt, from_r):
        """Hard occlusion weights from a 2-way softmax of the L1 errors (model_geometry.py:105-132)."""
        w_bwd, w_fwd, v_bwd, v_fwd = [], [], [], []
        for s in range(self.num_scales):
            il, it, ir = from_l[s], tgt[s], from_r[s]
            v_fwd.append(1 - (ir == 0).prod(1, keepdim=True).type_as(ir))
            v_bwd.append(1 - (il == 0).prod(1, keepdim=True).type_as(il))
            dl = torch.abs(it - il).mean(1, True)
            dr = torch.abs(it - ir).mean(1, True)
            wgt = occ_exp.weights(dl, dr)
            with torch.no_grad():
                hard = (wgt > 0.48).float()
                w_bwd.append(hard[:, 0:1])
                w_fwd.append(hard[:, 1:2])
        return w_bwd, w_fwd, v_bwd, v_fwd

    def compute_diff_weight(self, from_l, tgt, from_r):
        """Model_flow's soft gaussian weights (model_flow.py:105-138)."""
        d_bwd, d_fwd, w_bwd, w_fwd = [], [], [], []
        for s in range(self.num_scales):
            il, it, ir = from_l[s], tgt[s], from_r[s]
            vf = 1 - (ir == 0).prod(1, keepdim=True).type_as(ir)
            vb = 1 - (il == 0).prod(1, keepdim=True).type_as(il)
            dl = torch.abs(it - il).mean(1, True)
            dr = torch.abs(it - ir).mean(1, True)
            wgt = (1 - F.softmax(torch.cat((dl, dr), 1), 1)).detach()
            wgt = 2 * torch.exp(-(wgt - 0.5) ** 2 / 0.03)
            w_bwd.append(wgt[:, 0:1] * vb)
            w_fwd.append(wgt[:, 1:2] * vf)
            d_fwd.append(dr)
            d_bwd.append(dl)
        return d_bwd, d_fwd, w_bwd, w_fwd

    # a12 -----------------------------------------------------------------
    def compute_texture_mask(self, img_list, warped_list, source_list):
        """(mean_c|I-recon| < mean_c|I-I_src|) (model_geometry.py:134-140)."""
        return [(torch.abs(img_list[s] - warped_list[s]).mean(1, keepdim=True)
                 < torch.abs(img_list[s] - source_list[s]).mean(1, keepdim=True)).float()
                for s in range(self.num_scales)]

    # a10 / a9 ------------------------------------------------------------
    def compute_photometric_loss(self, img_list, warped_list, mask_list):
        """Masked L1, normalised by the mask mean (model_geometry.py:143-153)."""
        terms = []
        for s in range(self.num_scales):
            img, warped, mask = img_list[s], warped_list[s], mask_list[s]
            div = mask.mean((1, 2, 3))
            diff = torch.abs(img - warped) * mask.repeat(1, 3, 1, 1)
            terms.append((diff.mean((1, 2, 3)) / (div + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    def compute_loss_with_mask(self, diff_list, mask_list):
        """Model_flow pixel loss on the 1-channel diff (model_flow.py:94-103)."""
        terms = []
        for s in range(self.num_scales):
            diff, mask = diff_list[s], mask_list[s]
            div = mask.mean((1, 2, 3))
            terms.append(((diff * mask.repeat(1, 3, 1, 1)).mean((1, 2, 3)) / (div + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    def compute_ssim_loss(self, img_list, warped_list, mask_list):
        """clamp((1-SSIM(I*m, W*m))/2) mean over mask mean (model_geometry.py:212-223)."""
        terms = []
        for s in range(self.num_scales):
            img, warped, mask = img_list[s], warped_list[s], mask_list[s]
            div = mask.mean((1, 2, 3))
            m3 = mask.repeat(1, 3, 1, 1)
            val = torch.clamp((1.0 - SSIM(img * m3, warped * m3)) / 2.0, 0, 1).mean((1, 2, 3))
            terms.append((val / (div + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    def compute_consis_loss(self, predicted_depth_list, computed_depth_list, mask_list=None):
        """|c - p| / |c + p| clamped to [0,1], masked mean (model_geometry.py:182-193; model_depth.py:154-163 is the
        same without a mask).  Disabled in the reference's forward (:897-899) -- SURVEY.md 8(f) rank 3."""
        terms = []
        for s in range(self.num_scales):
            p, c = predicted_depth_list[s], computed_depth_list[s]
            diff = ((c - p).abs() / (c + p).abs()).clamp(0, 1)
            if mask_list is None:
                terms.append(diff.mean((1, 2, 3))[:, None])
            else:
                m = mask_list[s]
                terms.append(((diff * m).mean((1, 2, 3)) / (m.mean((1, 2, 3)) + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    # a16 / a17 / a18 -------------------------------------------------------
    def compute_smooth_loss(self, img, disps):
        """First-order edge-aware disparity smoothness at full resolution (model_geometry.py:225-252)."""
        h, w = img.shape[2], img.shape[3]
        terms = []
        for s in range(self.num_scales):
            d = F.interpolate(disps[s], size=(h, w), mode="bilinear", align_corners=False)
            gdx = torch.abs(d[:, :, :, :-1] - d[:, :, :, 1:])
            gdy = torch.abs(d[:, :, :-1, :] - d[:, :, 1:, :])
            gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
            giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
            gdx = gdx * torch.exp(-gix)
            gdy = gdy * torch.exp(-giy)
            terms.append((gdx.mean((1, 2, 3)) + gdy.mean((1, 2, 3)))[:, None])
        return torch.cat(terms, 1).sum(1)

    def cal_grad2_error(self, flow, img):
        """Second-order flow smoothness with exp(-10|dI|) weights (model_geometry.py:254-270)."""
        idx = img[:, :, :, 1:] - img[:, :, :, :-1]
        idy = img[:, :, 1:, :] - img[:, :, :-1, :]
        wx = torch.exp(-10.0 * torch.abs(idx).mean(1).unsqueeze(1))
        wy = torch.exp(-10.0 * torch.abs(idy).mean(1).unsqueeze(1))
        fx = flow[:, :, :, 1:] - flow[:, :, :, :-1]
        fy = flow[:, :, 1:, :] - flow[:, :, :-1, :]
        fxx = fx[:, :, :, 1:] - fx[:, :, :, :-1]
        fyy = fy[:, :, 1:, :] - fy[:, :, :-1, :]
        err = (wx[:, :, :, 1:] * torch.abs(fxx)).mean((1, 2, 3)) + (wy[:, :, 1:, :] * torch.abs(fyy)).mean((1, 2, 3))
        return err / 2.0

    def compute_loss_flow_smooth(self, flows, img_pyramid):
        """flow/20 per scale (model_geometry.py:272-279)."""
        terms = [self.cal_grad2_error(flows[s] / 20.0, img_pyramid[s])[:, None] for s in range(self.num_scales)]
        return torch.cat(terms, 1).sum(1)

    def compute_loss_flow_consis(self, fwd_flows, bwd_flows, occ_list):
        """|unit(fwd)+unit(bwd).detach| on (1-occ) (model_geometry.py:195-210)."""
        terms = []
        for s in range(self.num_scales):
            uf = _unit(fwd_flows[s])
            ub = _unit(bwd_flows[s]).float().detach()
            inv = 1 - occ_list[s]
            div = inv.mean((1, 2, 3))
            val = (torch.abs(uf + ub) * inv).mean((1, 2, 3))
            terms.append((val / (div + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    # a13 / a14 -------------------------------------------------------------
    def compute_dynamic_mask(self, intrinsics, depth, pose, flow):
        """|rigid-flow|, dynamic mask and score per scale (model_geometry.py:685-713)."""
        diffs, masks, scores = [], [], []
        h0 = depth[0].size(2)
        for s in range(self.num_scales):
            h = depth[s].size(2)
            down = h0 / h
            k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
            rigid = calculate_rigid_flow(depth[s], pose, k_s)
            bound = self.flow_consist_alpha * (torch.pow(_l2norm(flow[s]), 2) + torch.pow(_l2norm(rigid), 2)) \
                + self.flow_consist_beta
            diff = torch.abs(rigid - flow[s])
            diffs.append(diff)
            with torch.no_grad():
                masks.append((torch.pow(_l2norm(diff), 2) < bound).float())
                scores.append(1.0 / (1e-4 + _l2norm(diff)))
        return diffs, masks, scores

    def compute_depth_flow_consis_loss(self, flow_diffs, masks=None, scales=3):
        """Masked mean of |rigid-flow| (model_geometry.py:716-732)."""
        terms = []
        for s in range(scales):
            diff = flow_diffs[s]
            b, _, hh, ww = diff.shape
            mask = torch.ones(b, 1, hh, ww).to(diff.device) if masks is None else masks[s]
            div = mask.mean((1, 2, 3))
            terms.append(((diff * mask.repeat(1, 2, 1, 1)).mean((1, 2, 3)) / (div + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    # a15 -------------------------------------------------------------------
    def compute_epipolar_map(self, pose, flow, intrinsics, intrinsics_inverse):
        """Point-to-epipolar-line distance at one scale (model_geometry.py:355-403)."""
        b, _, h, w = flow.shape
        grid = _pixel_grid(b, h, w, flow)
        p1 = torch.cat([grid.view(b, 2, -1), torch.ones(b, 1, h * w).to(flow.device)], 1)
        p2 = torch.cat([(grid + flow).view(b, 2, -1), torch.ones(b, 1, h * w).to(flow.device)], 1)
        e_mat = compute_essential_matrix(pose)
        f_mat = intrinsics_inverse.transpose(1, 2).bmm(e_mat.bmm(intrinsics_inverse))
        line = _bmm3(f_mat, p1)
        la, lb = line[:, 0:1], line[:, 1:2]
        div = torch.sqrt(la * la + lb * lb) + 1e-6
        dist = torch.abs(torch.sum(p2 * line, dim=1, keepdim=True)) / div
        return dist.view(b, 1, h, w)

    def compute_epipolar_loss(self, dist_map, rigid_mask):
        """The masked mean is overwritten by the plain mean (model_geometry.py:413-418)."""
        return dist_map.mean((1, 2, 3))

    def get_rigid_mask(self, dist_map):
        """(model_geometry.py:420-425)."""
        with torch.no_grad():
            rigid = (dist_map < self.rigid_thres).float()
            inlier = (dist_map < self.inlier_thres).float()
            score = rigid * 1.0 / (1.0 + dist_map)
        return rigid, inlier, score

    # a19 -------------------------------------------------------------------
    def fusion_mask(self, valid, occ, dyna):
        return [valid[s] * occ[s] * dyna[s] for s in range(self.num_scales)]

    def fusion_mask_2item(self, a, b):
        return [a[s] * b[s] for s in range(self.num_scales)]

    # A.5 decision margins -------------------------------------------------------
    def decision_margins(self, img_l, img, img_r, disp_list, pose_vectors, flows_bwd, flows_fwd, K, signed=False):
        """|lhs - rhs| of every thresholded mask decision of ``geom_losses``, per pixel
        (``signed=True``: lhs - rhs itself for the dyna / texture / valid_to families, so that the effect of a perturbed
        input on a decision can be measured as a difference of margins).

        Test-only: the parity tests demand that a HIP mask may differ from the oracle's only at pixels whose
        margin is below the mask's fp32 noise floor, and that margin-checked seeds have no such pixel (so their
        masks are compared for exact equality).  Keys follow ``geom_losses``' mask dict; values are lists per
        scale of [B,1,H,W] tensors.  ``dyna`` margins are relative to the bound."""
        S = self.num_scales
        out = {k: [] for k in ("valid_bwd", "valid_fwd", "occ_bwd", "occ_fwd", "dyna_bwd", "dyna_fwd",
                               "texture_bwd", "texture_fwd", "valid_to_l", "valid_to_r")}
        sgn = (lambda t: t) if signed else torch.abs
        with torch.no_grad():
            pose_d = (pose_vectors[:, 0, :], pose_vectors[:, 1, :])
            pyr_t = self.generate_img_pyramid(img, S)
            pyr_src = (self.generate_img_pyramid(img_l, S), self.generate_img_pyramid(img_r, S))
            inf = torch.tensor(float("inf"))
            for s in range(S):
                b, _, h, w = disp_list[s].shape
                it = pyr_t[s]
                warped = []
                for d, (tag, flows) in enumerate((("bwd", flows_bwd), ("fwd", flows_fwd))):
                    grid = _flow_grid(pyr_src[d][s], flows[s])
                    cover = _gs(torch.ones_like(pyr_src[d][s]), grid, self.align_corners)[:, 0:1]
                    keep = (cover >= 0.9999).float()
                    wv = _gs(pyr_src[d][s], grid, self.align_corners) * keep
                    warped.append(wv)
                    nz = wv.abs().max(1, keepdim=True)[0]
                    out["valid_" + tag].append(torch.minimum((cover - 0.9999).abs(), torch.where(keep > 0, nz, inf)))
                dl = torch.abs(it - warped[0]).mean(1, True)
                dr = torch.abs(it - warped[1]).mean(1, True)
                wgt = occ_exp.weights(dl, dr)
                out["occ_bwd"].append((wgt[:, 0:1] - 0.48).abs())
                out["occ_fwd"].append((wgt[:, 1:2] - 0.48).abs())
                down = disp_list[0].size(2) / h
                k_s = torch.cat((K[:, 0:2] / down, K[:, 2:]), dim=1)
                for d, (tag, flows, src_img) in enumerate((("bwd", flows_bwd, img_l), ("fwd", flows_fwd, img_r))):
                    rigid = calculate_rigid_flow(disp_list[s], pose_d[d], k_s)
                    bound = self.flow_consist_alpha * (torch.pow(_l2norm(flows[s]), 2) + torch.pow(_l2norm(rigid), 2)) \
                        + self.flow_consist_beta
                    nd2 = torch.pow(_l2norm(torch.abs(rigid - flows[s])), 2)
                    out["dyna_" + tag].append(sgn(nd2 - bound) / bound)
                    src = F.interpolate(src_img, (h, w), mode="area")
                    grid, _ = _rigid_grid(disp_list[s], pose_d[d], k_s)
                    rec = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=self.align_corners)
                    e_rec = torch.abs(it - rec).mean(1, keepdim=True)
                    e_src = torch.abs(it - pyr_src[d][s]).mean(1, keepdim=True)
                    out["texture_" + tag].append(sgn(e_rec - e_src))
                    out["valid_to_" + ("l" if d == 0 else "r")].append(
                        sgn(grid.abs().max(dim=-1)[0] - 1).unsqueeze(1))
        return out

    # a21 geom ----------------------------------------------------------------
    def geom_losses(self, img_l, img, img_r, disp_l_list, disp_list, disp_r_list, pose_vectors,
                    flows_bwd, flows_fwd, K, K_inv, enable_depth_ssim=False, enable_depth_consis=False):
        """Everything from model_geometry.py:797 to :951 given the nets' outputs.

        Returns ``(loss_pack, masks)`` where ``masks`` holds the full-batch float masks the
        reference keeps as locals (the reference's ``mask_pack`` takes sample 0 of some of them)."""
        S = self.num_scales
        pose_bwd, pose_fwd = pose_vectors[:, 0, :], pose_vectors[:, 1, :]
        img_list = self.generate_img_pyramid(img, S)
        img_l_list = self.generate_img_pyramid(img_l, S)
        img_r_list = self.generate_img_pyramid(img_r, S)
        rec_l, valid_to_l, pd_l, cd_l = self.reconstruction(img_l, K, disp_list, disp_l_list, pose_bwd)
        rec_r, valid_to_r, pd_r, cd_r = self.reconstruction(img_r, K, disp_list, disp_r_list, pose_fwd)
        tex_bwd = self.compute_texture_mask(img_list, rec_l, img_l_list)
        tex_fwd = self.compute_texture_mask(img_list, rec_r, img_r_list)
        warp_l = self.warp_flow_pyramid(img_l_list, flows_bwd)
        warp_r = self.warp_flow_pyramid(img_r_list, flows_fwd)
        occ_bwd, occ_fwd, val_bwd, val_fwd = self.compute_occ_weight(warp_l, img_list, warp_r)
        diff_bwd, dyn_bwd, _ = self.compute_dynamic_mask(K, disp_list, pose_bwd, flows_bwd)
        diff_fwd, dyn_fwd, _ = self.compute_dynamic_mask(K, disp_list, pose_fwd, flows_fwd)
        dist_bwd = self.compute_epipolar_map(pose_bwd, flows_bwd[0], K, K_inv)
        dist_fwd = self.compute_epipolar_map(pose_fwd, flows_fwd[0], K, K_inv)
        rigid_bwd, inlier_bwd, _ = self.get_rigid_mask(dist_bwd)
        rigid_fwd, inlier_fwd, _ = self.get_rigid_mask(dist_fwd)
        fwd_mask = self.fusion_mask(val_fwd, occ_fwd, dyn_fwd)
        bwd_mask = self.fusion_mask(val_bwd, occ_bwd, dyn_bwd)
        fwd_tex = self.fusion_mask_2item(fwd_mask, tex_fwd)
        bwd_tex = self.fusion_mask_2item(bwd_mask, tex_bwd)
        fwd_vo = self.fusion_mask_2item(val_fwd, occ_fwd)
        bwd_vo = self.fusion_mask_2item(val_bwd, occ_bwd)
        fwd_vo_rigid = self.fusion_mask_2item(fwd_vo, dyn_fwd)
        bwd_vo_rigid = self.fusion_mask_2item(bwd_vo, dyn_bwd)
        fwd_vo_dyna = self.fusion_mask_2item(fwd_vo, [1 - m for m in dyn_fwd])
        bwd_vo_dyna = self.fusion_mask_2item(bwd_vo, [1 - m for m in dyn_bwd])

        dev = img_l.device
        lp = {}
        lp["loss_depth_pixel"] = self.compute_photometric_loss(img_list, rec_l, bwd_tex) + \
            self.compute_photometric_loss(img_list, rec_r, fwd_tex)
        # the two terms the reference keeps commented (model_geometry.py:889-891,897-899), as written there
        lp["loss_depth_ssim"] = (self.compute_ssim_loss(img_list, rec_l, bwd_tex) +
                                 self.compute_ssim_loss(img_list, rec_r, fwd_tex)) if enable_depth_ssim \
            else torch.zeros([2]).to(dev).requires_grad_()
        lp["loss_depth_smooth"] = self.compute_smooth_loss(img, disp_list) + \
            self.compute_smooth_loss(img_l, disp_l_list) + self.compute_smooth_loss(img_r, disp_r_list)
        lp["loss_depth_consis"] = (self.compute_consis_loss(pd_l, cd_l, bwd_tex) +
                                   self.compute_consis_loss(pd_r, cd_r, fwd_tex)) if enable_depth_consis \
            else torch.zeros([2]).to(dev).requires_grad_()
        lp["loss_flow_pixel"] = self.compute_photometric_loss(img_list, warp_l, bwd_vo_rigid) + \
            self.compute_photometric_loss(img_list, warp_r, fwd_vo_rigid) + \
            2 * self.compute_photometric_loss(img_list, warp_l, bwd_vo_dyna) + \
            2 * self.compute_photometric_loss(img_list, warp_r, fwd_vo_dyna)
        lp["loss_flow_ssim"] = self.compute_ssim_loss(img_list, warp_l, bwd_vo) + \
            self.compute_ssim_loss(img_list, warp_r, fwd_vo)
        lp["loss_flow_smooth"] = self.compute_loss_flow_smooth(flows_fwd, img_list) + \
            self.compute_loss_flow_smooth(flows_bwd, img_list)
        lp["loss_flow_consis"] = self.compute_loss_flow_consis(flows_fwd, flows_bwd, occ_fwd)
        lp["loss_depth_flow_consis"] = self.compute_depth_flow_consis_loss(diff_bwd, bwd_mask, 1) + \
            self.compute_depth_flow_consis_loss(diff_fwd, fwd_mask, 1)
        lp["loss_epipolar"] = self.compute_epipolar_loss(dist_bwd, dyn_bwd[0]) + \
            self.compute_epipolar_loss(dist_fwd, dyn_fwd[0])
        lp["loss_triangle"] = torch.zeros([2]).to(dev).requires_grad_()
        lp["loss_pnp"] = torch.zeros([2]).to(dev).requires_grad_()
        lp["loss_eight_point"] = torch.zeros([2]).to(dev).requires_grad_()
        masks = dict(occ_bwd=occ_bwd, occ_fwd=occ_fwd, valid_bwd=val_bwd, valid_fwd=val_fwd,
                     dyna_bwd=dyn_bwd, dyna_fwd=dyn_fwd, texture_bwd=tex_bwd, texture_fwd=tex_fwd,
                     valid_to_l=valid_to_l, valid_to_r=valid_to_r, rigid_bwd=rigid_bwd, rigid_fwd=rigid_fwd,
                     inlier_bwd=inlier_bwd, inlier_fwd=inlier_fwd, fwd_mask=fwd_mask, bwd_mask=bwd_mask,
                     dist_bwd=dist_bwd, dist_fwd=dist_fwd)
        return lp, masks

    # Model_depth.forward loss stack (model_depth.py:272-337) ----------------------
    def depth_losses(self, img_l, img, img_r, depth_l_list, depth_list, depth_r_list, pose_vectors, K,
                     enable_depth_ssim=False, enable_depth_consis=False):
        S = self.num_scales
        pose_bwd, pose_fwd = pose_vectors[:, 0, :], pose_vectors[:, 1, :]
        img_list = self.generate_img_pyramid(img, S)
        img_l_list = self.generate_img_pyramid(img_l, S)
        img_r_list = self.generate_img_pyramid(img_r, S)
        rec_l, valid_l, pd_l, cd_l = self.reconstruction(img_l, K, depth_list, depth_l_list, pose_bwd)
        rec_r, valid_r, pd_r, cd_r = self.reconstruction(img_r, K, depth_list, depth_r_list, pose_fwd)
        tex_bwd = self.compute_texture_mask(img_list, rec_l, img_l_list)
        tex_fwd = self.compute_texture_mask(img_list, rec_r, img_r_list)
        m_bwd = self.fusion_mask_2item(valid_l, tex_bwd)
        m_fwd = self.fusion_mask_2item(valid_r, tex_fwd)
        dev = img_l.device
        lp = {}
        lp["loss_depth_pixel"] = self.compute_photometric_loss(img_list, rec_l, m_bwd) + \
            self.compute_photometric_loss(img_list, rec_r, m_fwd)
        # the two terms the reference keeps commented (model_depth.py:326-327,332-333), as written there
        lp["loss_depth_ssim"] = (self.compute_ssim_loss(img_list, rec_l, m_bwd) +
                                 self.compute_ssim_loss(img_list, rec_r, m_fwd)) if enable_depth_ssim \
            else torch.zeros([2]).to(dev).requires_grad_()
        lp["loss_depth_smooth"] = self.compute_smooth_loss(img, depth_list) + \
            self.compute_smooth_loss(img_l, depth_l_list) + self.compute_smooth_loss(img_r, depth_r_list)
        lp["loss_depth_consis"] = (self.compute_consis_loss(pd_l, cd_l) + self.compute_consis_loss(pd_r, cd_r)) \
            if enable_depth_consis else torch.zeros([2]).to(dev).requires_grad_()
        return lp, dict(mask_bwd=m_bwd, mask_fwd=m_fwd, texture_bwd=tex_bwd, texture_fwd=tex_fwd,
                        valid_to_l=valid_l, valid_to_r=valid_r)

    # Model_flow.forward loss stack (model_flow.py:209-261, with the fixes of SURVEY.md) -----
    def flow_losses(self, img_l, img, img_r, flows_bwd, flows_fwd):
        n = len(flows_fwd)
        il = self.generate_img_pyramid_avgpool(img_l, n)
        it = self.generate_img_pyramid_avgpool(img, n)
        ir = self.generate_img_pyramid_avgpool(img_r, n)
        warp_l = self.warp_flow_pyramid(il, flows_bwd)
        warp_r = self.warp_flow_pyramid(ir, flows_fwd)
        d_bwd, d_fwd, w_bwd, w_fwd = self.compute_diff_weight(warp_l, it, warp_r)
        lp = {}
        lp["loss_flow_pixel"] = self.compute_loss_with_mask(d_fwd, w_fwd) + self.compute_loss_with_mask(d_bwd, w_bwd)
        lp["loss_flow_ssim"] = self.compute_ssim_loss(it, warp_r, w_fwd) + self.compute_ssim_loss(it, warp_l, w_bwd)
        lp["loss_flow_smooth"] = self.compute_loss_flow_smooth(flows_fwd, it) + \
            self.compute_loss_flow_smooth(flows_bwd, it)
        lp["loss_flow_consis"] = self.compute_loss_flow_consis(flows_fwd, flows_bwd, w_fwd)
        return lp, dict(weight_bwd=w_bwd, weight_fwd=w_fwd)
