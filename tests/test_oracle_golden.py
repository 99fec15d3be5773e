"""Pins the oracle (oracle/loss_stack_oracle.py) to the golden vectors captured from the real
reference by tests/golden/make_golden.py, in both align_corners modes.  CPU only.

Tolerances: element-wise maps 1e-6 abs (same ATen kernels, same association order -> normally
bit-identical), masks bit-exact, mean-reduced losses 1e-6 rel, grads 1e-5 rel."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_stack_oracle as O
from tests.golden import make_golden as MG
from unsupervised_depth_opticalflow_egomotion_amd import synthetic

T, N = MG.T, MG.N
ACS = [False, True]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def close(a, b, atol=1e-6, rtol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), atol=atol, rtol=rtol)


def bits(t):
    return np.packbits(N(t).astype(np.uint8).reshape(-1))


@pytest.mark.parametrize("ac", ACS)
def test_g1_warp_flow(golden_dir, ac):
    g = load(golden_dir, "G1_ac%d" % ac)
    x, flows, wgt = MG.g1_inputs()
    for name, fl in flows.items():
        for um in (False, True):
            xt, ft = T(x, True), T(fl, True)
            y = O.warp_flow(xt, ft, use_mask=um, align_corners=ac)
            (y * T(wgt)).sum().backward()
            key = "%s_mask%d" % (name, int(um))
            close(N(y), g[key + "_out"])
            close(N(ft.grad), g[key + "_gflow"], atol=1e-5, rtol=1e-5)
            close(N(xt.grad), g[key + "_gx"], atol=1e-6)
    # border-ring behaviour called out in SURVEY.md: zero flow is the identity only with align_corners=True
    y0 = N(O.warp_flow(T(x), T(flows["zero"]), use_mask=True, align_corners=ac))
    if ac:
        close(y0, x, atol=1e-6)
    else:
        assert np.all(y0[:, :, 0, :] == 0) and np.all(y0[:, :, :, -1] == 0)


def test_warp_flow_shape_error():
    with pytest.raises(ValueError):
        O.warp_flow(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 9))


@pytest.mark.parametrize("ac", ACS)
def test_g2_rigid(golden_dir, ac):
    g = load(golden_dir, "G2_ac%d" % ac)
    vec = T(g["vec"])
    assert np.array_equal(N(O.pose_vec2mat(vec)), g["pose_mat"])
    assert np.array_equal(N(O.compute_essential_matrix(vec)), g["essential"])
    for i, (h, w, case) in enumerate(MG.G2_CASES):
        img, depth, ref_depth, pose, k, wi, wd, wf = MG.g2_inputs(h, w, 210 + i, case)
        key = "%dx%d_%s" % (h, w, case)
        dt, rdt, pt = T(depth, True), T(ref_depth, True), T(pose, True)
        pi, valid, pd, cd = O.inverse_warp2(T(img), dt, rdt, pt, T(k), align_corners=ac)
        ((pi * T(wi)).sum() + (pd * T(wd)).sum() + (cd * T(wd)).sum() * 0.5).backward()
        # forward: the oracle states the reference's accumulation orders explicitly (oracle._bmm3, _inverse3), so it
        # reproduces the captured reference outputs bit for bit on any host
        for got, name in ((pi, "_img"), (valid, "_valid"), (pd, "_pdepth"), (cd, "_cdepth")):
            assert np.array_equal(N(got), g[key + name]), (key, name)
        # backward: autograd runs through the oracle's fp64-emulated FMA chain instead of sgemm's own backward --
        # same mathematics, rounding differs: 5e-4 of the gradient scale
        gscale(N(dt.grad), g[key + "_gdepth"])
        close(N(rdt.grad), g[key + "_grefdepth"], atol=1e-6)
        gscale(N(pt.grad), g[key + "_gpose"])
        dt3, pt3 = T(depth, True), T(pose, True)
        rf = O.calculate_rigid_flow(dt3, pt3, T(k))
        (rf * T(wf)).sum().backward()
        assert np.array_equal(N(rf), g[key + "_rflow"]), key
        gscale(N(dt3.grad), g[key + "_rflow_gdepth"])
        gscale(N(pt3.grad), g[key + "_rflow_gpose"])


def gscale(a, b, rel=5e-4, atol=1e-4):
    """atol: gradients that are analytically zero (identity pose: d rigid_flow / d depth) are pure cancellation noise
    of ~3e-5 in the reference itself."""
    assert np.abs(a - b).max() <= rel * np.abs(b).max() + atol, (np.abs(a - b).max(), np.abs(b).max())


def test_inverse3_matches_torch_inverse():
    """oracle._inverse3 (explicit LAPACK arithmetic) == torch.inverse in this container for every pyramid level of the
    KITTI-like intrinsics the configs use (no row exchange: fx >= cx, fy >= cy)."""
    from unsupervised_depth_opticalflow_egomotion_amd import synthetic
    for (h, w) in [(256, 832), (375, 1242), (128, 448), (64, 208), (70, 100), (32, 96)]:
        for ds in [1, 2, 4, 8, 375 / 187, 375 / 93, 375 / 46, 375 / 23, 375 / 11]:
            K = torch.from_numpy(synthetic.kitti_like_intrinsics(h, w).astype(np.float32))[None].clone()
            K[:, :2] = K[:, :2] / ds
            assert torch.equal(O._inverse3(K), K.inverse()), (h, w, ds)


def test_bmm3_is_the_golden_run_order():
    """oracle._bmm3 == the reference's ``a @ b`` as evaluated in the container that generated the goldens."""
    torch.manual_seed(3)
    a, b = torch.randn(4, 3, 3), torch.randn(4, 3, 53248)
    assert torch.equal(O._bmm3(a, b), a @ b)


def test_g3_ssim(golden_dir):
    g = load(golden_dir, "G3")
    x, y, m, c, wgt = MG.g3_inputs()
    xt, yt = T(x, True), T(y, True)
    s = O.SSIM(xt, yt)
    (s * T(wgt)).sum().backward()
    close(N(s), g["rand"])
    close(N(xt.grad), g["rand_gx"], atol=1e-5)
    close(N(yt.grad), g["rand_gy"], atol=1e-5)
    close(N(O.SSIM(T(x) * T(m), T(y) * T(m))), g["masked"])
    close(N(O.SSIM(T(c), T(c))), g["const"])
    close(N(O.SSIM(T(c), T(x[:1, :, :8, :8]))), g["const_vs_rand"])


def test_g4_corr(golden_dir):
    g = load(golden_dir, "G4")
    for i in range(len(MG.G4_CASES)):
        f1, f2, wgt = MG.g4_inputs(i)
        a, b = T(f1, True), T(f2, True)
        cv = O.corr_naive(a, b)
        (cv * T(wgt)).sum().backward()
        close(N(cv), g["c%d_out" % i])
        close(N(a.grad), g["c%d_g1" % i], atol=1e-5)
        close(N(b.grad), g["c%d_g2" % i], atol=1e-5)


@pytest.mark.parametrize("ac", ACS)
def test_g5_methods(golden_dir, ac):
    g = load(golden_dir, "G5_ac%d" % ac)
    inp = synthetic.make_loss_stack_inputs(*MG.G5_SHAPE, 3, seed=MG.G5_SEED)
    m = O.GeomLossOracle(align_corners=ac)
    il, it, ir = [T(a) for a in inp.imgs]
    disps, pose, fb, ff = MG.lists_to_t(inp, False)
    K, Ki = T(inp.K), T(inp.K_inv)
    pyr_l, pyr_t, pyr_r = (m.generate_img_pyramid(x, 3) for x in (il, it, ir))
    rec_l, vl, _, _ = m.reconstruction(il, K, disps[1], disps[0], pose[:, 0])
    rec_r, vr, _, _ = m.reconstruction(ir, K, disps[1], disps[2], pose[:, 1])
    wl = m.warp_flow_pyramid(pyr_l, fb)
    wr = m.warp_flow_pyramid(pyr_r, ff)
    assert len(wl) == 3  # the 1/8-scale flow is dropped by zip()
    occ_b, occ_f, val_b, val_f = m.compute_occ_weight(wl, pyr_t, wr)
    tex_b = m.compute_texture_mask(pyr_t, rec_l, pyr_l)
    tex_f = m.compute_texture_mask(pyr_t, rec_r, pyr_r)
    diff_b, dyn_b, sc_b = m.compute_dynamic_mask(K, disps[1], pose[:, 0], fb)
    diff_f, dyn_f, _ = m.compute_dynamic_mask(K, disps[1], pose[:, 1], ff)
    dist_b = m.compute_epipolar_map(pose[:, 0], fb[0], K, Ki)
    dist_f = m.compute_epipolar_map(pose[:, 1], ff[0], K, Ki)
    rig_b, inl_b, rsc_b = m.get_rigid_mask(dist_b)
    fm = m.fusion_mask(val_f, occ_f, dyn_f)
    bm = m.fusion_mask(val_b, occ_b, dyn_b)
    vo_f = m.fusion_mask_2item(val_f, occ_f)
    for s in range(3):
        close(N(pyr_t[s]), g["pyr_t_%d" % s])
        close(N(rec_l[s]), g["rec_l_%d" % s])
        close(N(rec_r[s]), g["rec_r_%d" % s])
        close(N(wl[s]), g["warp_l_%d" % s])
        close(N(wr[s]), g["warp_r_%d" % s])
        close(N(diff_b[s]), g["diff_b_%d" % s], atol=1e-5)
        close(N(diff_f[s]), g["diff_f_%d" % s], atol=1e-5)
        close(N(sc_b[s]), g["score_b_%d" % s], rtol=1e-5)
        for nm, lst in (("valid_to_l", vl), ("valid_to_r", vr), ("occ_b", occ_b), ("occ_f", occ_f),
                        ("val_b", val_b), ("val_f", val_f), ("tex_b", tex_b), ("tex_f", tex_f),
                        ("dyn_b", dyn_b), ("dyn_f", dyn_f), ("fwd_mask", fm), ("bwd_mask", bm)):
            assert np.array_equal(bits(lst[s]), g["%s_%d" % (nm, s)]), (nm, s)
    close(N(dist_b), g["dist_b"], atol=1e-5, rtol=1e-5)
    close(N(dist_f), g["dist_f"], atol=1e-5, rtol=1e-5)
    assert np.array_equal(bits(rig_b), g["rigid_b"]) and np.array_equal(bits(inl_b), g["inlier_b"])
    close(N(rsc_b), g["rigid_score_b"], atol=1e-6)
    close(N(m.compute_photometric_loss(pyr_t, rec_l, m.fusion_mask_2item(bm, tex_b))), g["photometric_rec_l"])
    close(N(m.compute_photometric_loss(pyr_t, wr, vo_f)), g["photometric_warp_r"])
    close(N(m.compute_ssim_loss(pyr_t, wr, vo_f)), g["ssim_warp_r"])
    close(N(m.compute_ssim_loss(pyr_t, wl, m.fusion_mask_2item(val_b, occ_b))), g["ssim_warp_l"])
    close(N(m.compute_smooth_loss(it, disps[1])), g["smooth_t"])
    close(N(m.compute_loss_flow_smooth(ff, pyr_t)), g["flow_smooth_f"])
    close(N(m.compute_loss_flow_consis(ff, fb, occ_f)), g["flow_consis"])
    close(N(m.compute_depth_flow_consis_loss(diff_f, fm, 1)), g["depth_flow_consis_1"])
    close(N(m.compute_depth_flow_consis_loss(diff_f, fm, 3)), g["depth_flow_consis_3"])
    close(N(m.compute_depth_flow_consis_loss(diff_b, None, 2)), g["depth_flow_consis_nomask"])
    close(N(m.compute_epipolar_loss(dist_f, dyn_f[0])), g["epipolar_loss"])


def run_oracle_geom(inp, ac, grad=True):
    m = O.GeomLossOracle(num_scales=inp.num_scales, align_corners=ac)
    disps, pose, fb, ff = MG.lists_to_t(inp, grad)
    il, it, ir = [T(a) for a in inp.imgs]
    lp, masks = m.geom_losses(il, it, ir, disps[0], disps[1], disps[2], pose, fb, ff, T(inp.K), T(inp.K_inv))
    return lp, masks, (disps, pose, fb, ff)


def check_grad_summary(g, prefix, t, stride=97, rtol=2e-4):
    gr = N(t.grad) if t.grad is not None else np.zeros(tuple(t.shape), np.float32)
    flat = gr.reshape(-1).astype(np.float64)
    ref = g[prefix + "_sum"]
    scale = max(ref[1], 1e-12)
    assert abs(flat.sum() - ref[0]) <= rtol * scale, prefix
    assert abs(np.abs(flat).sum() - ref[1]) <= rtol * scale, prefix
    sub = g[prefix + "_sub"]
    np.testing.assert_allclose(gr.reshape(-1)[::stride], sub, atol=rtol * max(np.abs(sub).max(), 1e-12), rtol=1e-4)


@pytest.mark.parametrize("ac", ACS)
@pytest.mark.parametrize("case", [0, 1])
def test_g6_end_to_end(golden_dir, ac, case):
    g = load(golden_dir, "G6_ac%d" % ac)
    b, h, w, seed = MG.G6_CASES[case]
    inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=seed)
    lp, masks, (disps, pose, fb, ff) = run_oracle_geom(inp, ac)
    key = "%dx%dx%d" % (b, h, w)
    assert list(lp.keys()) == [k[len(key) + 1:] for k in g.files if k.startswith(key + "_loss_")]
    for k, v in lp.items():
        close(N(v), g[key + "_" + k], atol=1e-6, rtol=2e-6)
    total = sum(MG.GEOM_WEIGHTS[k] * v.mean() for k, v in lp.items())
    close(N(total), g[key + "_total"], rtol=2e-6)
    total.backward()
    for nm, t in (("occ_fwd_mask", masks["occ_fwd"][0][0]), ("rigid_fwd_mask", masks["rigid_fwd"][0]),
                  ("inlier_fwd_mask", masks["inlier_fwd"][0]), ("dyna_fwd_mask", masks["dyna_fwd"][0][0]),
                  ("valid_fwd_mask", masks["valid_to_r"][0][0]), ("fwd_mask", masks["fwd_mask"][0][0]),
                  ("texture_mask_fwd", masks["texture_fwd"][0][0])):
        assert np.array_equal(bits(t), g[key + "_mp_" + nm]), nm
    close(N(pose.grad), g[key + "_gpose"], atol=1e-4, rtol=1e-4)
    for f in range(3):
        for s in range(3):
            check_grad_summary(g, key + "_gdisp_%d_%d" % (f, s), disps[f][s])
    for s in range(4):
        check_grad_summary(g, key + "_gflow_b_%d" % s, fb[s])
        check_grad_summary(g, key + "_gflow_f_%d" % s, ff[s])


@pytest.mark.parametrize("ac", ACS)
def test_g8_depth_and_flow_modes(golden_dir, ac):
    g = load(golden_dir, "G8_ac%d" % ac)
    inp = synthetic.make_loss_stack_inputs(2, 64, 208, 3, seed=808)
    m = O.GeomLossOracle(align_corners=ac)
    disps, pose, fb, ff = MG.lists_to_t(inp, True)
    il, it, ir = [T(a) for a in inp.imgs]
    lp, _ = m.depth_losses(il, it, ir, disps[0], disps[1], disps[2], pose, T(inp.K))
    (lp["loss_depth_pixel"].mean() + 0.5 * lp["loss_depth_smooth"].mean()).backward()
    for k, v in lp.items():
        close(N(v), g["depth_" + k], rtol=2e-6)
    close(N(pose.grad), g["depth_gpose"], atol=1e-4, rtol=1e-4)
    for f in range(3):
        for s in range(3):
            check_grad_summary(g, "depth_gdisp_%d_%d" % (f, s), disps[f][s], stride=31)
    inp = synthetic.make_loss_stack_inputs(1, 64, 192, 3, seed=809, num_flow_scales=4)
    _, _, fb, ff = MG.lists_to_t(inp, True)
    il, it, ir = [T(a) for a in inp.imgs]
    lp, _ = m.flow_losses(il, it, ir, fb, ff)
    (0.15 * lp["loss_flow_pixel"].mean() + 0.85 * lp["loss_flow_ssim"].mean()
     + 10 * lp["loss_flow_smooth"].mean() + 0.01 * lp["loss_flow_consis"].mean()).backward()
    for k, v in lp.items():
        close(N(v), g["flow_" + k], rtol=2e-6)
    for s in range(4):
        check_grad_summary(g, "flow_gflow_b_%d" % s, fb[s], stride=31)
        check_grad_summary(g, "flow_gflow_f_%d" % s, ff[s], stride=31)


@pytest.mark.parametrize("ac", ACS)
def test_g9_disabled_depth_terms(golden_dir, ac):
    """The oracle's restatement of the commented depth SSIM / depth consistency terms against the reference's own methods."""
    g = load(golden_dir, "G9_ac%d" % ac)
    inp = synthetic.make_loss_stack_inputs(*MG.G9_SHAPE, 3, seed=MG.G9_SEED)
    m = O.GeomLossOracle(3, align_corners=ac)
    d, p, fb, ff = MG.lists_to_t(inp, True)
    lp, _ = m.geom_losses(*[T(a) for a in inp.imgs], d[0], d[1], d[2], p, fb, ff, T(inp.K), T(inp.K_inv),
                          enable_depth_ssim=True, enable_depth_consis=True)
    (0.85 * lp["loss_depth_ssim"].mean() + 0.1 * lp["loss_depth_consis"].mean()).backward()
    close(N(lp["loss_depth_ssim"]), g["loss_depth_ssim"])
    close(N(lp["loss_depth_consis"]), g["loss_depth_consis"])
    gscale(N(p.grad), g["gpose"], rel=5e-5, atol=1e-6)
    for f in range(3):
        for s in range(3):
            got = N(d[f][s].grad) if d[f][s].grad is not None else np.zeros_like(g["gdisp_%d_%d" % (f, s)])
            gscale(got, g["gdisp_%d_%d" % (f, s)], rel=5e-5, atol=1e-9)
    # Model_depth's variants (model_depth.py:326-327,332-333: validity x texture mask; consistency without a mask)
    d, p, _, _ = MG.lists_to_t(inp, True)
    lp, _ = m.depth_losses(*[T(a) for a in inp.imgs], d[0], d[1], d[2], p, T(inp.K), enable_depth_ssim=True,
                           enable_depth_consis=True)
    (0.85 * lp["loss_depth_ssim"].mean() + 0.1 * lp["loss_depth_consis"].mean()).backward()
    close(N(lp["loss_depth_ssim"]), g["md_loss_depth_ssim"])
    close(N(lp["loss_depth_consis"]), g["md_loss_depth_consis"])
    gscale(N(p.grad), g["md_gpose"], rel=5e-5, atol=1e-6)
    for f in range(3):
        for s in range(3):
            got = N(d[f][s].grad) if d[f][s].grad is not None else np.zeros_like(g["md_gdisp_%d_%d" % (f, s)])
            gscale(got, g["md_gdisp_%d_%d" % (f, s)], rel=5e-5, atol=1e-9)


def _table_values(table):
    return [float(v) for v in str(table).strip().split("\n")[1].split(",")]


def test_g10_evaluation_metrics(golden_dir):
    """oracle/eval_oracle.py against the reference's own evaluate_flow.eval_flow_avg / calculate_error_rate and
    evaluate_depth.eval_depth / compute_errors (G10: identity-size cases, so cv2.resize plays no part): the metric
    arithmetic of the evaluation oracle is pinned; its resize_linear stays a statement of cv2's published definition."""
    from oracle import eval_oracle as EO
    g = load(golden_dir, "G10")
    c = MG.g10_inputs()
    acc = EO.eval_flow_avg(c["gt_flows"], c["nocs"], c["preds"], c["hw"], c["movs"])
    # table order of the reference: epe, noc, occ, move, static, move_rate, static_rate, err_rate
    np.testing.assert_allclose([acc[0], acc[1], acc[2], acc[4], acc[5], acc[6], acc[7], acc[3]], _table_values(g["flow_table_moving"]), atol=5.1e-5)
    acc4 = EO.eval_flow_avg(c["gt_flows"], c["nocs"], c["preds"], c["hw"])
    np.testing.assert_allclose(acc4[:4], _table_values(g["flow_table"]), atol=5.1e-5)
    for gt, pred, want in zip(c["gt_flows"], c["preds"], g["error_rates"]):
        epe = np.sqrt(np.sum(np.square(pred - gt[:, :, 0:2]), axis=2))
        assert EO.calculate_error_rate(epe, gt[:, :, 0:2], gt[:, :, 2]) == want
    np.testing.assert_allclose(EO.eval_depth(c["gt_depths"], c["pred_depths"]), g["depth_metrics"], rtol=1e-6)
    for gd, pd, want in zip(c["gt_depths"], c["pred_depths"], g["compute_errors"]):
        m = gd > 0
        np.testing.assert_allclose(EO.compute_errors(gd[m].astype(np.float64), pd[m].astype(np.float64)), want, rtol=1e-12)


def test_oracle_cr_exp_mode_moves_decisions_only_inside_the_exp_floor():
    """oracle.occ_exp("cr") (round 6): the occlusion softmax written out with a correctly rounded exponential.  Against the
    host's own F.softmax the WEIGHTS move by a few ulp; a hard decision may only move where |w - 0.48| lies inside the exp
    noise floor of tests/_margins.py -- the canonical function is a statement of the reference's, not a different one."""
    import torch
    from oracle import loss_stack_oracle as O
    from tests import _margins as M
    g = torch.Generator().manual_seed(5)
    dl = 0.3 * torch.rand(4, 1, 128, 416, generator=g)
    dr = dl + 0.08 + 0.02 * torch.randn(4, 1, 128, 416, generator=g)      # |dl - dr| around the flip point 0.08
    host = O.occ_exp.weights(dl, dr)
    with O.occ_exp("cr"):
        cr = O.occ_exp.weights(dl, dr)
    assert float((host - cr).abs().max()) <= 3e-7
    flips = (host > 0.48) != (cr > 0.48)
    assert bool(((host - 0.48).abs()[flips] < M.TAU["occ"]).all())
    assert int(((host > 0.48) & (host < 0.52)).sum()) > 1000            # the test does sit on the threshold
