#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Runs only in the build container, where /root/reference exists (it does not exist on
the GPU box; nothing under tests/ reads it at test time).  The reference has no tests
of its own for this path, so these captured outputs are what pins
the oracle (oracle/loss_stack_oracle.py), and through it the HIP path.

Harness-only shims, none of which is shipped as product code:
  * empty ``cv2`` module and a ``torchvision.models`` stand-in (``tests/golden/tv_stub.py``: torchvision's ResNet
    structure from stock ``torch.nn`` layers, independent of the product's own encoder);
  * ``torch.Tensor.get_device = lambda t: t.device`` so ``.to(x.get_device())`` works on CPU;
  * ``align_corners`` pinned per run by wrapping ``F.grid_sample`` (the reference does not
    pass the argument; False is torch 2.10's default, True is the torch<=1.2 behaviour).

Inputs come from ``synthetic.py`` / seeded numpy generators, so fixtures hold mostly
outputs.  Usage:  python tests/golden/make_golden.py [--only G1,G2,...]
"""
import argparse
import functools
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from unsupervised_depth_opticalflow_egomotion_amd import synthetic  # noqa: E402

synthetic.CONDITION_POSE = True   # golden inputs use conditioned poses (cos / sin unambiguous on any host), as the tests do

_ORIG_GRID_SAMPLE = F.grid_sample


def install_shims():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "torchvision" not in sys.modules:
        # an independent restatement of torchvision's ResNet (NOT the product's networks/resnet.py): G7 then pins
        # the product's encoder against a second implementation
        from tests.golden import tv_stub
        tv_stub.install(sys.modules)
    torch.Tensor.get_device = lambda self: self.device


CURRENT_AC = [False]


def set_align_corners(ac: bool):
    CURRENT_AC[0] = bool(ac)
    F.grid_sample = functools.partial(_ORIG_GRID_SAMPLE, align_corners=ac)
    nn.functional.grid_sample = F.grid_sample


def load_reference():
    install_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import core.networks as ref_networks  # noqa: F401  (runs the reference's own sys.path glue)
    import core.networks.structures.inverse_warp as iw
    import core.networks.structures.net_utils as nu
    import core.networks.structures.pwc_tf as pwc
    import core.networks.pytorch_ssim.ssim as ssim
    from core.networks.model_geometry import Model_geometry
    from core.networks.model_depth import Model_depth
    import core.networks.model_flow as mflow
    return dict(iw=iw, nu=nu, pwc=pwc, ssim=ssim, Model_geometry=Model_geometry, Model_depth=Model_depth,
                mflow=mflow)


def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).float()
    return t.requires_grad_(True) if grad else t


def N(t):
    return t.detach().cpu().numpy()


def packmask(t):
    a = N(t)
    assert np.all((a == 0) | (a == 1)), "mask is not {0,1}"
    return np.packbits(a.astype(np.uint8).reshape(-1))


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def kmat(b, h, w):
    return np.broadcast_to(synthetic.kitti_like_intrinsics(h, w).astype(np.float32), (b, 3, 3)).copy()


# ------------------------------------------------------------------------------------ G1
def g1_inputs():
    r = rng(101)
    b, c, h, w = 2, 3, 16, 24
    x = (r.random((b, c, h, w)) + 0.1).astype(np.float32)
    flows = {
        "zero": np.zeros((b, 2, h, w), np.float32),
        "const": np.stack([np.full((b, h, w), 3.0), np.zeros((b, h, w))], 1).astype(np.float32),
        "small": (2.0 * r.standard_normal((b, 2, h, w))).astype(np.float32),
        "large": (12.0 * r.standard_normal((b, 2, h, w))).astype(np.float32),
    }
    wgt = r.standard_normal((b, c, h, w)).astype(np.float32)
    return x, flows, wgt


def gen_g1(ref, out):
    x, flows, wgt = g1_inputs()
    for name, fl in flows.items():
        for um in (False, True):
            xt, ft = T(x, True), T(fl, True)
            y = ref["nu"].warp_flow(xt, ft, use_mask=um)
            (y * T(wgt)).sum().backward()
            key = "%s_mask%d" % (name, int(um))
            out[key + "_out"] = N(y)
            out[key + "_gflow"] = N(ft.grad)
            out[key + "_gx"] = N(xt.grad)


# ------------------------------------------------------------------------------------ G2
def g2_inputs(h, w, seed, case):
    r = rng(seed)
    b = 2
    img = (r.random((b, 3, h, w))).astype(np.float32)
    depth = (0.1 + 0.9 * r.random((b, 1, h, w))).astype(np.float32)
    ref_depth = (0.1 + 0.9 * r.random((b, 1, h, w))).astype(np.float32)
    pose = (0.05 * r.standard_normal((b, 6))).astype(np.float32)
    if case == "identity":
        pose[:] = 0
    elif case == "oob":
        pose[:, 0] = 0.6
        pose[:, 4] = 0.3
    elif case == "clamp":  # behind the camera: Z <= 1e-3 for every pixel
        pose[:, 2] = -2.0
    pose = synthetic.robust_pose(pose)   # cos / sin unambiguous across <= 0.6-ulp implementations
    wi = r.standard_normal((b, 3, h, w)).astype(np.float32)
    wd = r.standard_normal((b, 1, h, w)).astype(np.float32)
    wf = r.standard_normal((b, 2, h, w)).astype(np.float32)
    return img, depth, ref_depth, pose, kmat(b, h, w), wi, wd, wf


G2_CASES = [(16, 24, "rand"), (16, 24, "identity"), (16, 24, "oob"), (16, 24, "clamp"), (32, 104, "rand")]


def gen_g2(ref, out):
    iw = ref["iw"]
    r = rng(202)
    vec = synthetic.robust_pose((0.3 * r.standard_normal((5, 6))).astype(np.float32))
    out["vec"] = vec
    out["pose_mat"] = N(iw.pose_vec2mat(T(vec)))
    out["essential"] = N(iw.compute_essential_matrix(T(vec)))
    for i, (h, w, case) in enumerate(G2_CASES):
        img, depth, ref_depth, pose, k, wi, wd, wf = g2_inputs(h, w, 210 + i, case)
        key = "%dx%d_%s" % (h, w, case)
        dt, rdt, pt = T(depth, True), T(ref_depth, True), T(pose, True)
        pi, valid, pd, cd = iw.inverse_warp2(T(img), dt, rdt, pt, T(k))
        ((pi * T(wi)).sum() + (pd * T(wd)).sum() + (cd * T(wd)).sum() * 0.5).backward()
        out[key + "_img"], out[key + "_valid"], out[key + "_pdepth"], out[key + "_cdepth"] = N(pi), N(valid), N(pd), N(cd)
        out[key + "_gdepth"], out[key + "_grefdepth"], out[key + "_gpose"] = N(dt.grad), N(rdt.grad), N(pt.grad)
        # image-only loss (what the active depth pixel loss back-propagates)
        dt2, pt2 = T(depth, True), T(pose, True)
        pi2 = iw.inverse_warp2(T(img), dt2, T(ref_depth), pt2, T(k))[0]
        (pi2 * T(wi)).sum().backward()
        out[key + "_gdepth_img"], out[key + "_gpose_img"] = N(dt2.grad), N(pt2.grad)
        dt3, pt3 = T(depth, True), T(pose, True)
        rf = iw.calculate_rigid_flow(dt3, pt3, T(k))
        (rf * T(wf)).sum().backward()
        out[key + "_rflow"], out[key + "_rflow_gdepth"], out[key + "_rflow_gpose"] = N(rf), N(dt3.grad), N(pt3.grad)


# ------------------------------------------------------------------------------------ G3
def g3_inputs():
    r = rng(303)
    x = r.random((2, 3, 12, 20)).astype(np.float32)
    y = np.clip(x + 0.1 * r.standard_normal(x.shape), 0, 1).astype(np.float32)
    m = (r.random((2, 1, 12, 20)) > 0.3).astype(np.float32)
    c = np.full((1, 3, 8, 8), 0.5, np.float32)
    wgt = r.standard_normal(x.shape).astype(np.float32)
    return x, y, m, c, wgt


def gen_g3(ref, out):
    x, y, m, c, wgt = g3_inputs()
    S = ref["ssim"].SSIM
    xt, yt = T(x, True), T(y, True)
    s = S(xt, yt)
    (s * T(wgt)).sum().backward()
    out["rand"], out["rand_gx"], out["rand_gy"] = N(s), N(xt.grad), N(yt.grad)
    out["masked"] = N(S(T(x) * T(m), T(y) * T(m)))
    out["const"] = N(S(T(c), T(c)))
    out["const_vs_rand"] = N(S(T(c), T(x[:1, :, :8, :8])))


# ------------------------------------------------------------------------------------ G4
G4_CASES = [(2, 32, 4, 13), (1, 196, 4, 13), (1, 32, 8, 26), (1, 7, 5, 9)]


def g4_inputs(i):
    b, c, h, w = G4_CASES[i]
    r = rng(404 + i)
    return (r.standard_normal((b, c, h, w)).astype(np.float32), r.standard_normal((b, c, h, w)).astype(np.float32),
            r.standard_normal((b, 81, h, w)).astype(np.float32))


def gen_g4(ref, out):
    corr = ref["pwc"].PWC_tf.corr_naive
    for i in range(len(G4_CASES)):
        f1, f2, wgt = g4_inputs(i)
        a, bq = T(f1, True), T(f2, True)
        cv = corr(None, a, bq)
        (cv * T(wgt)).sum().backward()
        out["c%d_out" % i], out["c%d_g1" % i], out["c%d_g2" % i] = N(cv), N(a.grad), N(bq.grad)


# ------------------------------------------------------------------------------------ G5 / G6 / G8
def bare_geometry(ref, num_scales=3):
    m = ref["Model_geometry"].__new__(ref["Model_geometry"])
    nn.Module.__init__(m)
    m.num_scales, m.flow_consist_alpha, m.flow_consist_beta = num_scales, 0.01, 0.5
    m.rigid_thres, m.inlier_thres = 0.5, 0.1
    m.ratio, m.num, m.dataset = 0.3, 6000, "kitti_depth"
    return m


def lists_to_t(inp, grad):
    disps = [[T(a, grad) for a in lst] for lst in inp.disps]
    pose = T(inp.pose, grad)
    fb = [T(a, grad) for a in inp.flows_bwd]
    ff = [T(a, grad) for a in inp.flows_fwd]
    return disps, pose, fb, ff


G5_SHAPE, G5_SEED = (2, 48, 128), 505


def gen_g5(ref, out):
    """Every compute_* / fusion_* / get_rigid_mask called on a bare Model_geometry (48x128 base, B=2).

    H + W > 128 on purpose: below that ATen's bilinear resize switches to a differently associated kernel
    (UpSampleKernel.cpp _use_vectorized_kernel_cond_2d); every image the real configs resize is far above it."""
    inp = synthetic.make_loss_stack_inputs(*G5_SHAPE, 3, seed=G5_SEED)
    store_margins(out, "g5", inp, CURRENT_AC[0])
    m = bare_geometry(ref)
    il, it, ir = [T(a) for a in inp.imgs]
    disps, pose, fb, ff = lists_to_t(inp, False)
    K, Ki = T(inp.K), T(inp.K_inv)
    pyr_l, pyr_t, pyr_r = (m.generate_img_pyramid(x, 3) for x in (il, it, ir))
    for s in range(3):
        out["pyr_t_%d" % s] = N(pyr_t[s])
    rec_l, vl, pdl, cdl = m.reconstruction(il, K, disps[1], disps[0], pose[:, 0])
    rec_r, vr, pdr, cdr = m.reconstruction(ir, K, disps[1], disps[2], pose[:, 1])
    wl = m.warp_flow_pyramid(pyr_l, fb)
    wr = m.warp_flow_pyramid(pyr_r, ff)
    assert len(wl) == 3
    occ_b, occ_f, val_b, val_f = m.compute_occ_weight(wl, pyr_t, wr)
    tex_b = m.compute_texture_mask(pyr_t, rec_l, pyr_l)
    tex_f = m.compute_texture_mask(pyr_t, rec_r, pyr_r)
    diff_b, dyn_b, sc_b = m.compute_dynamic_mask(K, disps[1], pose[:, 0], fb)
    diff_f, dyn_f, sc_f = m.compute_dynamic_mask(K, disps[1], pose[:, 1], ff)
    dist_b = m.compute_epipolar_map(pose[:, 0], fb[0], K, Ki)
    dist_f = m.compute_epipolar_map(pose[:, 1], ff[0], K, Ki)
    rig_b, inl_b, rsc_b = m.get_rigid_mask(dist_b)
    fm = m.fusion_mask(val_f, occ_f, dyn_f)
    bm = m.fusion_mask(val_b, occ_b, dyn_b)
    vo_f = m.fusion_mask_2item(val_f, occ_f)
    for s in range(3):
        out["rec_l_%d" % s], out["rec_r_%d" % s] = N(rec_l[s]), N(rec_r[s])
        out["warp_l_%d" % s], out["warp_r_%d" % s] = N(wl[s]), N(wr[s])
        out["diff_b_%d" % s], out["diff_f_%d" % s] = N(diff_b[s]), N(diff_f[s])
        out["score_b_%d" % s] = N(sc_b[s])
        for nm, lst in (("valid_to_l", vl), ("valid_to_r", vr), ("occ_b", occ_b), ("occ_f", occ_f),
                        ("val_b", val_b), ("val_f", val_f), ("tex_b", tex_b), ("tex_f", tex_f),
                        ("dyn_b", dyn_b), ("dyn_f", dyn_f), ("fwd_mask", fm), ("bwd_mask", bm)):
            out["%s_%d" % (nm, s)] = packmask(lst[s])
    out["dist_b"], out["dist_f"] = N(dist_b), N(dist_f)
    out["rigid_b"], out["inlier_b"], out["rigid_score_b"] = packmask(rig_b), packmask(inl_b), N(rsc_b)
    out["photometric_rec_l"] = N(m.compute_photometric_loss(pyr_t, rec_l, m.fusion_mask_2item(bm, tex_b)))
    out["photometric_warp_r"] = N(m.compute_photometric_loss(pyr_t, wr, vo_f))
    out["ssim_warp_r"] = N(m.compute_ssim_loss(pyr_t, wr, vo_f))
    out["ssim_warp_l"] = N(m.compute_ssim_loss(pyr_t, wl, m.fusion_mask_2item(val_b, occ_b)))
    out["smooth_t"] = N(m.compute_smooth_loss(it, disps[1]))
    out["flow_smooth_f"] = N(m.compute_loss_flow_smooth(ff, pyr_t))
    out["flow_consis"] = N(m.compute_loss_flow_consis(ff, fb, occ_f))
    out["depth_flow_consis_1"] = N(m.compute_depth_flow_consis_loss(diff_f, fm, 1))
    out["depth_flow_consis_3"] = N(m.compute_depth_flow_consis_loss(diff_f, fm, 3))
    out["depth_flow_consis_nomask"] = N(m.compute_depth_flow_consis_loss(diff_b, None, 2))
    out["epipolar_loss"] = N(m.compute_epipolar_loss(dist_f, dyn_f[0]))


class _Seq:
    """Callable standing in for a network: returns the queued outputs in call order."""

    def __init__(self, outs):
        self.outs, self.i = list(outs), 0

    def __call__(self, *a, **k):
        o = self.outs[self.i % len(self.outs)]
        self.i += 1
        return o


def run_ref_geom(ref, inp, grad=True):
    m = bare_geometry(ref, inp.num_scales)
    disps, pose, fb, ff = lists_to_t(inp, grad)
    object.__setattr__(m, "depth_net", _Seq(disps))            # called for img_l, img, img_r
    object.__setattr__(m, "pose_net", _Seq([pose]))
    object.__setattr__(m, "fpyramid", _Seq([None]))
    object.__setattr__(m, "pwc_model", _Seq([fb, ff]))         # bwd first, then fwd (model_geometry.py:794-795)
    images = torch.cat([T(a) for a in inp.imgs], dim=2)
    b = images.shape[0]
    k_ms = T(inp.K).unsqueeze(1)
    ki_ms = T(inp.K_inv).unsqueeze(1)
    torch.manual_seed(0)
    loss_pack, mask_pack = m.forward([images, k_ms, ki_ms])
    return loss_pack, mask_pack, (disps, pose, fb, ff)


GEOM_WEIGHTS = dict(loss_flow_pixel=0.15, loss_flow_ssim=0.85, loss_flow_smooth=10.0, loss_flow_consis=0.01,
                    loss_depth_pixel=1.0, loss_depth_ssim=0.85, loss_depth_smooth=0.5, loss_depth_consis=0.1,
                    loss_depth_flow_consis=1.0, loss_epipolar=0.1, loss_triangle=0.001, loss_pnp=0.1,
                    loss_eight_point=0.1)


def grad_summary(out, prefix, t, stride=97):
    g = N(t.grad) if t.grad is not None else np.zeros(tuple(t.shape), np.float32)
    flat = g.reshape(-1).astype(np.float64)
    out[prefix + "_sum"] = np.array([flat.sum(), np.abs(flat).sum(), (flat * flat).sum()])
    out[prefix + "_sub"] = g.reshape(-1)[::stride].copy()


# seeds are margin-checked (tests/_margins.py): no pixel of any mask lies inside its fp32 noise floor in either
# align_corners mode, so the HIP masks must EQUAL the reference's (tests/test_api_cpu.py re-checks this on CPU)
G6_CASES = [(2, 128, 448, 600), (1, 256, 832, 1219)]


def store_margins(out, key, inp, ac):
    """Minimum decision margin per mask and scale, from the oracle's restatement of the decisions
    (the oracle is held bit-identical to the reference by tests/test_oracle_golden.py)."""
    from tests import _margins
    mg = _margins.geom_margins(inp, ac, inp.num_scales)
    for k, v in _margins.min_margins(mg).items():
        out[key + "_margin_" + k] = v
    out[key + "_within"] = np.array([sum(_margins.within_counts(mg).values())])


def gen_g6(ref, out):
    for (b, h, w, seed) in G6_CASES:
        inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=seed)
        lp, mp, (disps, pose, fb, ff) = run_ref_geom(ref, inp)
        key = "%dx%dx%d" % (b, h, w)
        store_margins(out, key, inp, CURRENT_AC[0])
        total = sum(GEOM_WEIGHTS[k] * v.mean() for k, v in lp.items())
        total.backward()
        for k, v in lp.items():
            out[key + "_" + k] = N(v)
        out[key + "_total"] = N(total)
        for k in ("occ_fwd_mask", "rigid_fwd_mask", "inlier_fwd_mask", "dyna_fwd_mask", "valid_fwd_mask", "fwd_mask",
                  "texture_mask_fwd"):
            out[key + "_mp_" + k] = np.packbits((np.asarray(mp[k]) // 255).astype(np.uint8).reshape(-1))
        out[key + "_gpose"] = N(pose.grad)
        for f in range(3):
            for s in range(3):
                grad_summary(out, key + "_gdisp_%d_%d" % (f, s), disps[f][s])
        for s in range(4):
            grad_summary(out, key + "_gflow_b_%d" % s, fb[s])
            grad_summary(out, key + "_gflow_f_%d" % s, ff[s])


def gen_g8(ref, out):
    """Model_depth.forward (B=2, 64x208) and the patched Model_flow.forward (B=1, 64x192) loss packs."""
    inp = synthetic.make_loss_stack_inputs(2, 64, 208, 3, seed=808)
    md = ref["Model_depth"].__new__(ref["Model_depth"])
    nn.Module.__init__(md)
    md.num_scales, md.dataset = 3, "kitti_depth"
    disps, pose, fb, ff = lists_to_t(inp, True)
    object.__setattr__(md, "depth_net", _Seq(disps))
    object.__setattr__(md, "pose_net", _Seq([pose]))
    images = torch.cat([T(a) for a in inp.imgs], dim=2)
    lp, _ = md.forward([images, T(inp.K).unsqueeze(1), T(inp.K_inv).unsqueeze(1)])
    (lp["loss_depth_pixel"].mean() + 0.5 * lp["loss_depth_smooth"].mean()).backward()
    for k, v in lp.items():
        out["depth_" + k] = N(v)
    out["depth_gpose"] = N(pose.grad)
    for f in range(3):
        for s in range(3):
            grad_summary(out, "depth_gdisp_%d_%d" % (f, s), disps[f][s], stride=31)
    # Model_flow: the shipped forward has a NameError (output_flow) -> define the module global (harness fix)
    mflow = ref["mflow"]
    mflow.output_flow = False
    inp = synthetic.make_loss_stack_inputs(1, 64, 192, 3, seed=809, num_flow_scales=4)
    mf = mflow.Model_flow.__new__(mflow.Model_flow)
    nn.Module.__init__(mf)
    mf.num_scales, mf.dataset = 3, "kitti_depth"
    _, _, fb, ff = lists_to_t(inp, True)
    object.__setattr__(mf, "fpyramid", _Seq([None]))
    object.__setattr__(mf, "pwc_model", _Seq([fb, ff]))
    images = torch.cat([T(a) for a in inp.imgs], dim=2)
    lp = mf.forward([images, None, None])
    (0.15 * lp["loss_flow_pixel"].mean() + 0.85 * lp["loss_flow_ssim"].mean()
     + 10 * lp["loss_flow_smooth"].mean() + 0.01 * lp["loss_flow_consis"].mean()).backward()
    for k, v in lp.items():
        out["flow_" + k] = N(v)
    for s in range(4):
        grad_summary(out, "flow_gflow_b_%d" % s, fb[s], stride=31)
        grad_summary(out, "flow_gflow_f_%d" % s, ff[s], stride=31)


# ------------------------------------------------------------------------------------ G9 (disabled depth terms)
G9_SHAPE, G9_SEED = (2, 64, 208), 77     # margin-checked (tests/test_hip_loss_stack.STRICT)


def gen_g9(ref, out):
    """The depth SSIM / depth consistency terms the reference keeps commented (model_geometry.py:889-891,897-899),
    evaluated exactly as those lines read, by calling the reference's own methods on a bare Model_geometry."""
    inp = synthetic.make_loss_stack_inputs(*G9_SHAPE, 3, seed=G9_SEED)
    m = bare_geometry(ref)
    il, it, ir = [T(a) for a in inp.imgs]
    disps, pose, fb, ff = lists_to_t(inp, True)
    K = T(inp.K)
    pyr_l, pyr_t, pyr_r = (m.generate_img_pyramid(x, 3) for x in (il, it, ir))
    rec_l, vl, pdl, cdl = m.reconstruction(il, K, disps[1], disps[0], pose[:, 0])
    rec_r, vr, pdr, cdr = m.reconstruction(ir, K, disps[1], disps[2], pose[:, 1])
    with torch.no_grad():
        wl, wr = m.warp_flow_pyramid(pyr_l, fb), m.warp_flow_pyramid(pyr_r, ff)
        occ_b, occ_f, val_b, val_f = m.compute_occ_weight(wl, pyr_t, wr)
        tex_b, tex_f = m.compute_texture_mask(pyr_t, rec_l, pyr_l), m.compute_texture_mask(pyr_t, rec_r, pyr_r)
        _, dyn_b, _ = m.compute_dynamic_mask(K, disps[1], pose[:, 0], fb)
        _, dyn_f, _ = m.compute_dynamic_mask(K, disps[1], pose[:, 1], ff)
        bwd_tex = m.fusion_mask_2item(m.fusion_mask(val_b, occ_b, dyn_b), tex_b)
        fwd_tex = m.fusion_mask_2item(m.fusion_mask(val_f, occ_f, dyn_f), tex_f)
    l_ssim = m.compute_ssim_loss(pyr_t, rec_l, bwd_tex) + m.compute_ssim_loss(pyr_t, rec_r, fwd_tex)
    l_cons = m.compute_consis_loss(pdl, cdl, bwd_tex) + m.compute_consis_loss(pdr, cdr, fwd_tex)
    (0.85 * l_ssim.mean() + 0.1 * l_cons.mean()).backward()
    out["loss_depth_ssim"], out["loss_depth_consis"] = N(l_ssim), N(l_cons)
    out["gpose"] = N(pose.grad)
    for f in range(3):
        for s in range(3):
            out["gdisp_%d_%d" % (f, s)] = N(disps[f][s].grad) if disps[f][s].grad is not None else np.zeros(tuple(disps[f][s].shape), np.float32)
    store_margins(out, "g9", inp, CURRENT_AC[0])
    # Model_depth's commented terms (model_depth.py:326-327,332-333): SSIM on validity x texture, the consistency term
    # without a mask (model_depth.py:154-163) -- the reference's own methods on a bare Model_depth, same inputs
    md = ref["Model_depth"].__new__(ref["Model_depth"])
    nn.Module.__init__(md)
    md.num_scales, md.dataset = 3, "kitti_depth"
    disps, pose, _, _ = lists_to_t(inp, True)
    pyr_l, pyr_t, pyr_r = (md.generate_img_pyramid(x, 3) for x in (il, it, ir))
    rec_l, vl, pdl, cdl = md.reconstruction(il, K, disps[1], disps[0], pose[:, 0])
    rec_r, vr, pdr, cdr = md.reconstruction(ir, K, disps[1], disps[2], pose[:, 1])
    with torch.no_grad():
        m_b = md.fusion_mask(vl, md.compute_texture_mask(pyr_t, rec_l, pyr_l))
        m_f = md.fusion_mask(vr, md.compute_texture_mask(pyr_t, rec_r, pyr_r))
    l_ssim = md.compute_ssim_loss(pyr_t, rec_l, m_b) + md.compute_ssim_loss(pyr_t, rec_r, m_f)
    l_cons = md.compute_consis_loss(pdl, cdl) + md.compute_consis_loss(pdr, cdr)
    (0.85 * l_ssim.mean() + 0.1 * l_cons.mean()).backward()
    out["md_loss_depth_ssim"], out["md_loss_depth_consis"] = N(l_ssim), N(l_cons)
    out["md_gpose"] = N(pose.grad)
    for f in range(3):
        for s in range(3):
            out["md_gdisp_%d_%d" % (f, s)] = N(disps[f][s].grad) if disps[f][s].grad is not None else np.zeros(tuple(disps[f][s].shape), np.float32)


# ------------------------------------------------------------------------------------ G7 (real nets)
def closed_form_state(model, scale=0.02):
    """Fill every parameter/buffer with a closed-form function of (key, flat index) so that the reference
    model here and the build's model on the GPU box hold identical weights without shipping 86 MB."""
    import zlib
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_mean"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = torch.ones_like(v)
        elif ".bn" in k or "downsample.1" in k:
            out[k] = torch.ones_like(v) if k.endswith("weight") else torch.zeros_like(v)
        else:
            n = v.numel()
            phase = float(zlib.crc32(k.encode()) % 97)
            idx = torch.arange(n, dtype=torch.float64)
            fan = max(n // max(v.shape[0], 1), 1)
            amp = scale if k.endswith("bias") else min(1.0, 1.7 / np.sqrt(fan))
            out[k] = (amp * torch.sin(0.37 * idx + phase)).float().view_as(v)
    model.load_state_dict(out)
    return out


def g7_cfg():
    return types.SimpleNamespace(dataset="kitti_depth", num_scales=3, flow_consist_alpha=0.01, flow_consist_beta=0.5,
                                 num_input_frames=3, geometric_ratio=0.3, geometric_num=6000, pose_beta=1, mode="geom")


def g7_inputs():
    images, k_ms, ki_ms = synthetic.make_triplet_batch(1, 256, 832, 3, seed=707)
    return T(images), T(k_ms), T(ki_ms)


def gen_g7(ref, out):
    m = ref["Model_geometry"](g7_cfg())
    closed_form_state(m)
    out["state_keys"] = np.array(list(m.state_dict().keys()))
    out["state_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    images, k_ms, ki_ms = g7_inputs()
    m.train()
    torch.manual_seed(0)
    lp, mp = m([images, k_ms, ki_ms])
    for k, v in lp.items():
        out["train_" + k] = N(v)
    m.eval()
    h = 256
    img_l, img, img_r = images[:, :, :h], images[:, :, h:2 * h], images[:, :, 2 * h:]
    with torch.no_grad():
        d = m.infer_depth(img)
        p = m.infer_pose(torch.cat([img_l, img, img_r], 1))
        f = m.inference_flow(img, img_r)
    out["eval_depth_stats"] = np.array([float(d.mean()), float(d.std()), float(d.min()), float(d.max())])
    out["eval_depth_crop"] = N(d[0, 0, 100:108, 400:408])
    out["eval_pose"] = N(p)
    out["eval_flow_stats"] = np.array([float(f.mean()), float(f.std()), float(f.abs().max())])
    out["eval_flow_crop"] = N(f[0, :, 100:108, 400:408])


# ------------------------------------------------------------------------------------ G10
def g10_inputs():
    """Seeded flow / depth evaluation cases.  Predictions have the ground truth's size and cfg.img_hw equals it, so the
    reference's cv2.resize is the identity and only the metric arithmetic is pinned (cv2 itself cannot be imported)."""
    r = rng(1010)
    H, W, n = 40, 64, 3
    gt_flows, nocs, movs, preds = [], [], [], []
    for _ in range(n):
        valid = (r.random((H, W)) > 0.3).astype(np.float64)
        gt = np.zeros((H, W, 3), np.float64)
        gt[:, :, 0:2] = r.normal(0, 12, (H, W, 2)) * valid[:, :, None]
        gt[:, :, 2] = valid
        noc = valid * (r.random((H, W)) > 0.2)
        mov = (r.random((H, W)) > 0.6).astype(np.float64)
        pred = (gt[:, :, 0:2] + r.normal(0, 3.0, (H, W, 2)) * (r.random((H, W, 1)) > 0.5)).astype(np.float32)
        gt_flows.append(gt); nocs.append(noc); movs.append(mov); preds.append(pred)
    gt_depths, pred_depths = [], []
    for _ in range(n):
        g = (r.random((60, 200)) * 90.0).astype(np.float32)
        g[r.random((60, 200)) > 0.4] = 0.0                      # sparse LiDAR ground truth
        p = (np.abs(g + r.normal(0, 4, g.shape)) * 0.37 + 0.5).astype(np.float32)
        gt_depths.append(g); pred_depths.append(p)
    return dict(hw=(H, W), gt_flows=gt_flows, nocs=nocs, movs=movs, preds=preds, gt_depths=gt_depths, pred_depths=pred_depths)


def load_reference_evaluation():
    """core/evaluation/{evaluate_flow, evaluate_depth, evaluation_utils}.py loaded by path; cv2 / png / skimage / imageio
    are stand-in modules (cv2.resize = identity for an unchanged size, anything else raises)."""
    import importlib.util
    ev = os.path.join(REF, "core", "evaluation")
    cv2 = sys.modules.get("cv2") or types.ModuleType("cv2")

    def resize(img, size, interpolation=None):
        if (img.shape[1], img.shape[0]) != tuple(size):
            raise RuntimeError("cv2 stand-in: only the identity resize exists here")
        return img
    cv2.resize, cv2.INTER_LINEAR = resize, 1
    sys.modules["cv2"] = cv2
    for name in ("png", "skimage", "skimage.io", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if ev not in sys.path:
        sys.path.insert(0, ev)
    mods = {}
    for name in ("evaluation_utils", "evaluate_depth", "evaluate_flow"):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(ev, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def gen_g10(ref, out):
    ev = load_reference_evaluation()
    c = g10_inputs()
    cfg = types.SimpleNamespace(img_hw=c["hw"], model_dir=None)
    copy = lambda lst: [np.array(a, copy=True) for a in lst]   # noqa: E731  (the reference's eval_depth writes into its inputs)
    out["flow_table"] = np.array(ev["evaluate_flow"].eval_flow_avg(copy(c["gt_flows"]), copy(c["nocs"]), copy(c["preds"]), cfg))
    out["flow_table_moving"] = np.array(ev["evaluate_flow"].eval_flow_avg(copy(c["gt_flows"]), copy(c["nocs"]), copy(c["preds"]), cfg,
                                                                          moving_masks=copy(c["movs"])))
    rates = []
    for gt, pred in zip(c["gt_flows"], c["preds"]):
        epe = np.sqrt(np.sum(np.square(pred - gt[:, :, 0:2]), axis=2))
        rates.append(ev["evaluate_flow"].calculate_error_rate(epe, gt[:, :, 0:2], gt[:, :, 2]))
    out["error_rates"] = np.array(rates, np.float64)
    out["depth_metrics"] = np.array(ev["evaluate_depth"].eval_depth(copy(c["gt_depths"]), copy(c["pred_depths"])), np.float64)
    errs = []
    for g, p in zip(c["gt_depths"], c["pred_depths"]):
        m = g > 0
        errs.append(ev["evaluation_utils"].compute_errors(g[m].astype(np.float64), p[m].astype(np.float64)))
    out["compute_errors"] = np.array(errs, np.float64)


# ------------------------------------------------------------------------------------ G11
def g11_inputs():
    """Matches, cameras and depth maps for the triangulation family (model_geometry.py:427-470, 569-683)."""
    r = rng(1111)
    b, n, h, w = 2, 200, 48, 160
    K = kmat(b, h, w)
    pose = (0.05 * r.standard_normal((b, 6))).astype(np.float32)
    pose[:, 3:] *= 0.3
    pose[:, 0] += 0.4                                             # a baseline, so that the rays are not parallel
    xy = np.stack([r.uniform(2, w - 3, (b, n)), r.uniform(2, h - 3, (b, n))], 1).astype(np.float32)
    match = np.concatenate([xy, xy + r.normal(0, 3, (b, 2, n)).astype(np.float32)], 1)
    depth1 = r.uniform(2, 30, (b, 1, h, w)).astype(np.float32)
    depth2 = r.uniform(2, 30, (b, 1, h, w)).astype(np.float32)
    flow = r.normal(0, 3, (b, 2, h, w)).astype(np.float32)
    score = r.random((b, 1, h, w)).astype(np.float32)
    return dict(K=K, pose=pose, match=match, depth1=depth1, depth2=depth2, flow=flow, score=score, hw=(h, w))


def gen_g11(ref, out):
    m = bare_geometry(ref)
    m.ratio, m.num, m.dataset = 0.3, 50, "kitti_depth"
    c = g11_inputs()
    K, pose, match = T(c["K"]), T(c["pose"]), T(c["match"])
    Ki = torch.inverse(K)
    P1, P2 = ref["iw"].compute_projection_matrix(pose, K)
    out["P2"] = N(P2)
    pts = m.midpoint_triangulate(match, K, Ki, P1, P2)
    out["points"] = N(pts)
    c1, z1 = m.reproject(P1, pts)
    c2, z2 = m.reproject(P2, pts)
    out["coord1"], out["depth1"], out["coord2"], out["depth2"] = N(c1), N(z1), N(c2), N(z2)
    d1 = T(c["depth1"])
    r1, i1 = m.register_depth(d1, c1, z1)
    out["reg_pred1"], out["reg_inter1"] = N(r1), N(i1)
    a, bb = m.affine_adapt(i1, z1.abs() + 0.5, use_translation=True)
    out["affine_a"], out["affine_b"], out["scale_a"] = N(a), N(bb), N(m.scale_adapt(i1, z1.abs() + 0.5))
    out["trian_loss"] = N(m.compute_triangulate_loss(match, pose, K, Ki, [d1], [T(c["depth2"])]))
    flow, score = T(c["flow"]), T(c["score"])
    b, _, h, w = flow.shape
    grid = m.meshgrid(b, h, w)
    full = torch.cat([grid, grid + flow], 1).view(b, 4, -1)
    tm, td, ts = m.top_ratio_sample(full, d1.view(b, 1, -1), score.view(b, 1, -1), 0.3)
    out["top_match"], out["top_depth"], out["top_score"] = N(tm), N(td), N(ts)
    torch.manual_seed(1111)
    sm, sd = m.sample_match(flow, d1, score)
    out["sample_match"], out["sample_depth"] = N(sm), N(sd)


GENERATORS = dict(G1=gen_g1, G2=gen_g2, G3=gen_g3, G4=gen_g4, G5=gen_g5, G6=gen_g6, G7=gen_g7, G8=gen_g8, G9=gen_g9, G10=gen_g10, G11=gen_g11)
AC_INDEPENDENT = {"G3", "G4", "G10"}   # no grid_sample inside (G11: register_depth samples with grid_sample)


# ------------------------------------------------------------------------------------ G12
def g12_inputs():
    """Inputs for the legacy signatures of inverse_warp.py (:30-107,148-224): first-generation warp, quaternion poses."""
    r = rng(1212)
    b, h, w = 2, 24, 40
    img = r.random((b, 3, h, w)).astype(np.float32)
    depth = (0.3 + 0.7 * r.random((b, h, w))).astype(np.float32)
    pose = synthetic.robust_pose((0.04 * r.standard_normal((b, 6))).astype(np.float32))
    pose[:, 0] += 0.05
    wgt = r.standard_normal((b, 3, h, w)).astype(np.float32)
    return dict(img=img, depth=depth, pose=pose, K=kmat(b, h, w), wgt=wgt)


def gen_g12(ref, out):
    iw = ref["iw"]
    c = g12_inputs()
    K = T(c["K"])
    out["quat2mat"] = N(iw.quat2mat(T(c["pose"])[:, 3:]))
    out["pose_mat_quat"] = N(iw.pose_vec2mat(T(c["pose"]), "quat"))
    iw.pixel_coords = None
    cam = iw.pixel2cam(T(c["depth"]), K.inverse())
    out["pixel2cam"] = N(cam)
    proj = K @ iw.pose_vec2mat(T(c["pose"]))
    out["cam2pixel"] = N(iw.cam2pixel(cam, proj[:, :, :3], proj[:, :, -1:]))
    out["cam2pixel_change_shape"] = N(iw.cam2pixel_change_shape(cam, proj[:, :, :3], proj[:, :, -1:]))
    g2, z2 = iw.cam2pixel2(cam, proj[:, :, :3], proj[:, :, -1:], "zeros")
    out["cam2pixel2_grid"], out["cam2pixel2_z"] = N(g2), N(z2)
    out["skew"] = N(iw.skewsymmetric(T(c["pose"])[:, :3].to("cpu")))
    out["meshgrid"] = N(iw.meshgrid(5, 7))
    for mode in ("euler", "quat"):
        d, p = T(c["depth"], True), T(c["pose"], True)
        y, valid = iw.inverse_warp(T(c["img"]), d, p, K, rotation_mode=mode)
        (y * T(c["wgt"])).sum().backward()
        out["iw_%s_img" % mode], out["iw_%s_valid" % mode] = N(y), packmask(valid.float())
        out["iw_%s_gdepth" % mode], out["iw_%s_gpose" % mode] = N(d.grad), N(p.grad)


GENERATORS["G12"] = gen_g12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = load_reference()
    names = [n for n in GENERATORS if not args.only or n in args.only.split(",")]
    for name in names:
        for ac in (False, True):
            if name in AC_INDEPENDENT and ac:
                continue
            set_align_corners(ac)
            out = {}
            GENERATORS[name](ref, out)
            suffix = "" if name in AC_INDEPENDENT else "_ac%d" % int(ac)
            path = os.path.join(HERE, "%s%s.npz" % (name, suffix))
            np.savez_compressed(path, **out)
            print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")
    F.grid_sample = _ORIG_GRID_SAMPLE


if __name__ == "__main__":
    main()
