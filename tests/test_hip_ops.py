"""GPU parity of the per-operator HIP kernels (through the C ABI) against the oracle and the
golden vectors.  Tolerances (fp32): maps 1e-5 abs for bilinear outputs on [0,1] images, 2e-5
for SSIM maps, pixel coordinates 2e-3 px (one fp32 ulp at x~800 is 6e-5 and the projection
chains ~10 roundings), grads 1e-4 relative to the gradient scale, masks bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_stack_oracle as O
from tests.golden import make_golden as MG

pytestmark = pytest.mark.gpu
T, N = MG.T, MG.N


def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).float().to(dev())
    return t.requires_grad_(True) if grad else t


def close(a, b, atol=1e-5, rtol=1e-5):
    np.testing.assert_allclose(N(a) if isinstance(a, torch.Tensor) else a,
                               N(b) if isinstance(b, torch.Tensor) else b, atol=atol, rtol=rtol)


def gclose(a, b, rel=1e-4, max_outliers=0, atol=0.0):
    """max |a-b| <= rel * max|b| + atol; ``max_outliers`` elements may exceed it (pixels whose
    {0,1} validity decision sits within fp32 noise of its threshold differ legitimately).
    ``atol`` is the cancellation-noise floor for gradients that are analytically ~0."""
    a, b = (N(a) if isinstance(a, torch.Tensor) else a), (N(b) if isinstance(b, torch.Tensor) else b)
    scale = max(np.abs(b).max(), 1e-12)
    bad = int((np.abs(a - b) > rel * scale + atol).sum())
    assert bad <= max_outliers, "grad mismatch: %d elements (allowed %d), max %g vs scale %g" % (
        bad, max_outliers, np.abs(a - b).max(), scale)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("ac", [False, True])
def test_warp_flow_golden(golden_dir, ac):
    from unsupervised_depth_opticalflow_egomotion_amd.structures import warp_flow
    g = load(golden_dir, "G1_ac%d" % ac)
    x, flows, wgt = MG.g1_inputs()
    for name, fl in flows.items():
        for um in (False, True):
            xt, ft = G(x, True), G(fl, True)
            y = warp_flow(xt, ft, use_mask=um, align_corners=ac)
            (y * G(wgt)).sum().backward()
            key = "%s_mask%d" % (name, int(um))
            close(y, g[key + "_out"], atol=2e-6)
            # exact agreement of the zeroed (masked) pixels
            assert np.array_equal(N(y) == 0, g[key + "_out"] == 0), key
            gclose(ft.grad, g[key + "_gflow"])
            gclose(xt.grad, g[key + "_gx"])


def test_warp_flow_errors():
    from unsupervised_depth_opticalflow_egomotion_amd.structures import warp_flow
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    with pytest.raises(ValueError):
        warp_flow(torch.zeros(1, 3, 8, 8, device=dev()), torch.zeros(1, 2, 8, 9, device=dev()))
    with pytest.raises(DfeError):  # no CPU fallback
        warp_flow(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))


@pytest.mark.parametrize("shape", [(2, 3, 64, 208), (1, 32, 64, 208), (2, 128, 8, 26), (1, 3, 375, 1242)])
@pytest.mark.parametrize("ac", [False, True])
def test_warp_flow_oracle(shape, ac):
    from unsupervised_depth_opticalflow_egomotion_amd.structures import warp_flow
    r = MG.rng(7 + shape[1])
    b, c, h, w = shape
    x = r.random(shape).astype(np.float32)
    fl = (3.0 * r.standard_normal((b, 2, h, w))).astype(np.float32)
    wgt = r.standard_normal(shape).astype(np.float32)
    for um in (False, True):
        xo, fo = T(x, True), T(fl, True)
        yo = O.warp_flow(xo, fo, use_mask=um, align_corners=ac)
        (yo * T(wgt)).sum().backward()
        xt, ft = G(x, True), G(fl, True)
        y = warp_flow(xt, ft, use_mask=um, align_corners=ac)
        (y * G(wgt)).sum().backward()
        close(y, yo, atol=2e-6)
        assert np.array_equal(N(y) == 0, N(yo) == 0)
        gclose(ft.grad, fo.grad)
        gclose(xt.grad, xo.grad)


def test_forward_splat_occlusion_properties():
    """Model_flow.get_occlusion_mask_from_flow (model_flow.py:33-39; no oracle exists -- the reference's transformerFwd is
    undefined): zero flow -> ones; an integer shift -> shifted ones with the vacated columns at 0; mass conservation
    before the clamp while every footprint stays inside; clamp to [0,1]; agreement with a numpy scatter."""
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    from unsupervised_depth_opticalflow_egomotion_amd.models import Model_flow
    B, H, W = 2, 24, 40
    z = torch.zeros(B, 2, H, W, device=dev())
    assert torch.equal(ops.forward_splat_ones(z), torch.ones(B, 1, H, W, device=dev()))
    sh = z.clone(); sh[:, 0] = 3.0
    o = N(ops.forward_splat_ones(sh))
    assert (o[:, :, :, :3] == 0).all() and (o[:, :, :, 3:] == 1).all()
    r = MG.rng(12)
    fl = (0.4 * r.standard_normal((B, 2, H, W))).astype(np.float32)
    fl[:, :, :2] = 0; fl[:, :, -2:] = 0; fl[:, :, :, :2] = 0; fl[:, :, :, -2:] = 0      # footprints stay inside
    raw = N(ops.forward_splat_ones(G(fl), clamp=False))
    np.testing.assert_allclose(raw.sum((1, 2, 3)), H * W, rtol=1e-5)
    ref = np.zeros((B, H, W), np.float64)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                tx, ty = x + float(fl[b, 0, y, x]), y + float(fl[b, 1, y, x])
                x0, y0 = int(np.floor(tx)), int(np.floor(ty)); wx, wy = tx - x0, ty - y0
                for (yy, xx, wgt) in ((y0, x0, (1 - wx) * (1 - wy)), (y0, x0 + 1, wx * (1 - wy)), (y0 + 1, x0, (1 - wx) * wy), (y0 + 1, x0 + 1, wx * wy)):
                    if 0 <= yy < H and 0 <= xx < W:
                        ref[b, yy, xx] += wgt
    np.testing.assert_allclose(raw[:, 0], ref, atol=2e-5)     # the device forms x + u in fp32 (ulp(40) = 3.8e-6)
    mf = Model_flow.__new__(Model_flow)
    m = mf.get_occlusion_mask_from_flow((B, 3, H, W), G(fl))
    assert m.shape == (B, 3, H, W) and float(m.max()) <= 1.0 and float(m.min()) >= 0.0


def test_device_input_pipeline():
    """ops.prepare_triplets against a numpy statement of KITTI_Prepared's image path
    (kitti_prepared.py:63-90: per-frame cv2.resize INTER_LINEAR geometry, flip, / 255, HWC -> CHW).  Tolerance 1e-5 of
    the [0,1] range: the device forms the source coordinate in fp32 (ATen's fma(scale, dst + 0.5, -0.5)), the numpy
    statement in double -- at W0 = 122 that is 7e-6 px, times a slope of up to 255 grey levels per pixel, / 255."""
    from oracle import eval_oracle as EO
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    r = MG.rng(31)
    B, H0, W0, H, W = 3, 37, 122, 32, 96
    raw = r.integers(0, 256, (B, 3 * H0, W0, 3), dtype=np.uint8)
    flip = np.array([0, 1, 0], np.uint8)
    got = N(ops.prepare_triplets(torch.from_numpy(raw).to(dev()), (H, W), torch.from_numpy(flip)))
    assert got.shape == (B, 3, 3 * H, W)
    for b in range(B):
        frames = [EO.resize_linear(raw[b, f * H0:(f + 1) * H0].astype(np.float32), (H, W)) for f in range(3)]
        img = np.concatenate(frames, 0)
        if flip[b]:
            img = img[:, ::-1]
        ref = (img / 255.0).transpose(2, 0, 1)
        np.testing.assert_allclose(got[b], ref, atol=1e-5)
    ks, kis = ops.rescale_intrinsics(np.array([[721.5, 0, 609.6], [0, 721.5, 172.9], [0, 0, 1]]), (375, 1242), (256, 832), 3)
    assert ks.shape == (3, 3, 3) and abs(float(ks[1, 0, 0]) - 721.5 * 832 / 1242 / 2) < 1e-3
    np.testing.assert_allclose((ks[2] @ kis[2]).numpy(), np.eye(3), atol=1e-5)


def test_exact_math_sequences_exhaustive():
    """The 3 / 5-instruction reciprocal, division and square-root sequences the mask-deciding expressions use return the
    IEEE results bit for bit: every fp32 reciprocal and square root, 2^32 random quotients + 2^32 with all-ones divisors."""
    from unsupervised_depth_opticalflow_egomotion_amd import _lib
    lib = _lib.get_lib()
    counts = torch.full((4,), -1, dtype=torch.int64, device=dev())
    _lib.check(lib.dfe_exact_math_selftest(_lib.ptr(counts), 1 << 32, _lib.stream_ptr()), "dfe_exact_math_selftest")
    torch.cuda.synchronize()
    assert counts.tolist() == [0, 0, 0, 0], counts.tolist()


def test_warp_flow_backward_is_deterministic():
    """Both gradients are bitwise equal from run to run at the PWC feature-warp shapes: grad wrt flow sums over all
    channels in a fixed order (per-group register sums met in LDS); grad wrt x is a scatter-add done in 64-bit fixed
    point with integer atomics (csrc/dfe_scatter.h: integer addition is associative, the order the atomics retire in
    cannot matter).  The last case sends every pixel to the same spot (H*W-fold collisions)."""
    from unsupervised_depth_opticalflow_egomotion_amd.structures import warp_flow
    r = MG.rng(99)
    for shape, pile in [((2, 128, 8, 26), False), ((2, 64, 32, 104), False), ((1, 196, 4, 13), False), ((2, 32, 64, 208), False),
                        ((1, 4, 64, 208), True)]:
        b, c, h, w = shape
        x, fl = r.standard_normal(shape).astype(np.float32), (2.0 * r.standard_normal((b, 2, h, w))).astype(np.float32)
        if pile:
            yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
            fl[:, 0], fl[:, 1] = 100.3 - xx, 30.6 - yy
        wgt = r.standard_normal(shape).astype(np.float32)
        grads = []
        for _ in range(3):
            xt, ft = G(x, True), G(fl, True)
            (warp_flow(xt, ft) * G(wgt)).sum().backward()
            grads.append((ft.grad.clone(), xt.grad.clone()))
        for k in (0, 1):
            assert torch.equal(grads[0][k], grads[1][k]) and torch.equal(grads[0][k], grads[2][k]), (shape, k)


def test_warp_flow_backward_scatter_accuracy_and_edge_cases():
    """The fixed-point scatter against a float64 scatter of the same contributions: closer than an fp32 accumulation
    could promise (quantum 2^-36 of the largest gradient's binade; one fp32 rounding at the end), including gradients
    spanning 30 binades, a heavy pile-up, an all-zero gradient (exact zeros) and a non-finite one (NaN everywhere: fails
    loudly instead of wrapping an integer)."""
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    r = MG.rng(5)
    # align_corners=True with W-1, H-1 powers of two and dyadic flows: the sampling coordinates and the tap weights are
    # exact in fp32, so the float64 oracle scatters exactly the kernel's contributions and only the accumulation differs
    B, C, H, W = 2, 8, 33, 129
    x = r.standard_normal((B, C, H, W)).astype(np.float32)
    fl = (r.integers(-6, 7, (B, 2, H, W)) + r.integers(0, 4, (B, 2, H, W)) / 4.0).astype(np.float32)
    fl[1, 0] = 50.25 - np.arange(W)[None, :]; fl[1, 1] = 10.5 - np.arange(H)[:, None]       # sample 1: everything to one spot
    g = (r.standard_normal((B, C, H, W)) * np.exp2(r.integers(-30, 1, (B, C, H, W)))).astype(np.float32)

    def run(gout):
        xt = G(x, True)
        y = ops.warp_flow(xt, G(fl), align_corners=True)
        y.backward(G(gout))
        return N(xt.grad)

    xo = torch.from_numpy(x).double().requires_grad_(True)
    yo = O.warp_flow(xo, torch.from_numpy(fl).double(), use_mask=False, align_corners=True)
    yo.backward(torch.from_numpy(g).double())
    want = xo.grad.numpy()
    xo.grad = None
    O.warp_flow(xo, torch.from_numpy(fl).double(), use_mask=False, align_corners=True).backward(torch.from_numpy(np.abs(g)).double())
    mass = xo.grad.numpy()                                 # sum of |contribution| per element
    got = run(g)
    quantum = 2.0 ** -36 * 2.0 * np.abs(g).max()           # of the largest gradient's binade
    for b in range(B):
        err = np.abs(got[b] - want[b])
        # each contribution g * w is one fp32 product (2^-24 relative), then a half-quantum per term, then one fp32 rounding
        bound = 2.0 ** -24 * mass[b] + 0.5 * quantum * 4 * H * W + 2.0 ** -24 * np.abs(want[b])
        assert (err <= bound).all(), (b, float((err - bound).max()))
    assert np.array_equal(run(np.zeros_like(g)), np.zeros_like(g))
    gbad = g.copy(); gbad[0, 0, 0, 0] = np.inf
    assert np.isnan(run(gbad)).all()


@pytest.mark.parametrize("ac", [False, True])
def test_rigid_golden(golden_dir, ac):
    from unsupervised_depth_opticalflow_egomotion_amd.structures import (
        inverse_warp2, calculate_rigid_flow, pose_vec2mat, compute_essential_matrix)
    g = load(golden_dir, "G2_ac%d" % ac)
    # the kernels evaluate the reference's own fp32 arithmetic (k_prepare_cameras, project(): ATen's small-bmm loops,
    # LAPACK's 3x3 inverse, sgemm's FMA order, grid_sample's FMA chain) -> forward outputs are BIT-IDENTICAL to the
    # values the reference produced (poses are robust_pose'd: cos / sin unambiguous)
    assert np.array_equal(N(pose_vec2mat(G(g["vec"]))), g["pose_mat"])
    assert np.array_equal(N(compute_essential_matrix(G(g["vec"]))), g["essential"])
    for i, (h, w, case) in enumerate(MG.G2_CASES):
        img, depth, ref_depth, pose, k, wi, wd, wf = MG.g2_inputs(h, w, 210 + i, case)
        key = "%dx%d_%s" % (h, w, case)
        dt, rdt, pt = G(depth, True), G(ref_depth, True), G(pose, True)
        pi, valid, pd, cd = inverse_warp2(G(img), dt, rdt, pt, G(k), align_corners=ac)
        ((pi * G(wi)).sum() + (pd * G(wd)).sum() + (cd * G(wd)).sum() * 0.5).backward()
        for got, name in ((valid, "_valid"), (pi, "_img"), (pd, "_pdepth"), (cd, "_cdepth")):
            assert np.array_equal(N(got), g[key + name]), (key, name, float(np.abs(N(got) - g[key + name]).max()))
        # gradients: 2e-4 of the gradient's scale; atol covers analytically-zero gradients that are cancellation noise
        # in the reference itself (identity pose)
        gclose(dt.grad, g[key + "_gdepth"], rel=2e-4, atol=1e-4)
        gclose(rdt.grad, g[key + "_grefdepth"], rel=2e-4)
        gclose(pt.grad, g[key + "_gpose"], rel=2e-4, atol=1e-3)
        dt3, pt3 = G(depth, True), G(pose, True)
        rf = calculate_rigid_flow(dt3, pt3, G(k))
        (rf * G(wf)).sum().backward()
        assert np.array_equal(N(rf), g[key + "_rflow"]), key
        gclose(dt3.grad, g[key + "_rflow_gdepth"], rel=2e-4, atol=1e-4)
        gclose(pt3.grad, g[key + "_rflow_gpose"], rel=2e-4, atol=1e-3)


def test_scatters_of_the_per_operator_api_are_reproducible():
    """inverse_warp2's reference-depth gradient and the forward splat are scatter-adds too: both go through the
    fixed-point accumulators of csrc/dfe_scatter.h and come out bitwise equal from run to run."""
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    from unsupervised_depth_opticalflow_egomotion_amd.structures import inverse_warp2
    img, depth, ref_depth, pose, k, wi, wd, wf = MG.g2_inputs(128, 416, 31, "rand")
    outs = []
    for _ in range(3):
        rdt = G(ref_depth, True)
        pi, valid, pd, cd = inverse_warp2(G(img), G(depth), rdt, G(pose), G(k), align_corners=False)
        (pd * G(wd)).sum().backward()
        outs.append(rdt.grad.clone())
    assert float(outs[0].abs().max()) > 0 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    r = MG.rng(77)
    fl = G((3.0 * r.standard_normal((2, 2, 128, 416))).astype(np.float32))
    sp = [ops.forward_splat_ones(fl, clamp=False) for _ in range(3)]
    assert torch.equal(sp[0], sp[1]) and torch.equal(sp[0], sp[2])
    # mass: every deposited weight lands somewhere or falls off the border -- at most one per source pixel
    assert float(sp[0].double().sum()) <= 2 * 128 * 416 * (1 + 1e-6)


@pytest.mark.parametrize("ac", [False, True])
def test_rigid_ops_full_size_vs_oracle(ac):
    """inverse_warp2 / calculate_rigid_flow at 256x832 (BASELINE size) against the oracle: every output EQUAL."""
    from unsupervised_depth_opticalflow_egomotion_amd.structures import inverse_warp2, calculate_rigid_flow
    img, depth, ref_depth, pose, k, wi, wd, wf = MG.g2_inputs(256, 832, 777, "rand")
    pi, valid, pd, cd = inverse_warp2(G(img), G(depth), G(ref_depth), G(pose), G(k), align_corners=ac)
    po, vo, pdo, cdo = O.inverse_warp2(T(img), T(depth), T(ref_depth), T(pose), T(k), align_corners=ac)
    for got, ref, name in ((valid, vo, "valid"), (pi, po, "img"), (pd, pdo, "pdepth"), (cd, cdo, "cdepth")):
        assert np.array_equal(N(got), N(ref)), (name, float(np.abs(N(got) - N(ref)).max()))
    assert np.array_equal(N(calculate_rigid_flow(G(depth), G(pose), G(k))), N(O.calculate_rigid_flow(T(depth), T(pose), T(k))))


def test_pose_mats_backward():
    from unsupervised_depth_opticalflow_egomotion_amd.structures import pose_vec2mat, compute_essential_matrix
    r = MG.rng(11)
    vec = (0.3 * r.standard_normal((7, 6))).astype(np.float32)
    wt, we = r.standard_normal((7, 3, 4)).astype(np.float32), r.standard_normal((7, 3, 3)).astype(np.float32)
    vo = T(vec, True)
    ((O.pose_vec2mat(vo) * T(wt)).sum() + (O.compute_essential_matrix(vo) * T(we)).sum()).backward()
    vg = G(vec, True)
    ((pose_vec2mat(vg) * G(wt)).sum() + (compute_essential_matrix(vg) * G(we)).sum()).backward()
    gclose(vg.grad, vo.grad, rel=1e-5)


def test_ssim_golden(golden_dir):
    from unsupervised_depth_opticalflow_egomotion_amd.pytorch_ssim import SSIM
    g = load(golden_dir, "G3")
    x, y, m, c, wgt = MG.g3_inputs()
    xt, yt = G(x, True), G(y, True)
    s = SSIM(xt, yt)
    (s * G(wgt)).sum().backward()
    close(s, g["rand"], atol=2e-5)
    gclose(xt.grad, g["rand_gx"], rel=2e-4)
    gclose(yt.grad, g["rand_gy"], rel=2e-4)
    close(SSIM(G(x) * G(m), G(y) * G(m)), g["masked"], atol=2e-5)
    close(SSIM(G(c), G(c)), g["const"], atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 3, 64, 208), (1, 3, 33, 70), (1, 3, 256, 832)])
def test_ssim_oracle(shape):
    from unsupervised_depth_opticalflow_egomotion_amd.pytorch_ssim import SSIM
    r = MG.rng(21)
    x = r.random(shape).astype(np.float32)
    y = np.clip(x + 0.1 * r.standard_normal(shape), 0, 1).astype(np.float32)
    wgt = r.standard_normal(shape).astype(np.float32)
    xo, yo = T(x, True), T(y, True)
    so = O.SSIM(xo, yo)
    (so * T(wgt)).sum().backward()
    xt, yt = G(x, True), G(y, True)
    s = SSIM(xt, yt)
    (s * G(wgt)).sum().backward()
    close(s, so, atol=2e-5)
    gclose(xt.grad, xo.grad, rel=2e-4)
    gclose(yt.grad, yo.grad, rel=2e-4)


def test_corr_golden(golden_dir):
    from unsupervised_depth_opticalflow_egomotion_amd.ops import corr81
    g = load(golden_dir, "G4")
    for i in range(len(MG.G4_CASES)):
        f1, f2, wgt = MG.g4_inputs(i)
        a, b = G(f1, True), G(f2, True)
        cv = corr81(a, b)
        (cv * G(wgt)).sum().backward()
        close(cv, g["c%d_out" % i], atol=1e-5)
        gclose(a.grad, g["c%d_g1" % i])
        gclose(b.grad, g["c%d_g2" % i])


@pytest.mark.parametrize("shape", [(4, 196, 4, 13), (2, 128, 8, 26), (2, 96, 16, 52), (2, 64, 32, 104), (1, 32, 64, 208)])
def test_corr_oracle(shape):
    from unsupervised_depth_opticalflow_egomotion_amd.ops import corr81
    r = MG.rng(31)
    f1 = r.standard_normal(shape).astype(np.float32)
    f2 = r.standard_normal(shape).astype(np.float32)
    wgt = r.standard_normal((shape[0], 81, shape[2], shape[3])).astype(np.float32)
    ao, bo = T(f1, True), T(f2, True)
    co = O.corr_naive(ao, bo)
    (co * T(wgt)).sum().backward()
    a, b = G(f1, True), G(f2, True)
    cv = corr81(a, b)
    (cv * G(wgt)).sum().backward()
    close(cv, co, atol=1e-5)
    gclose(a.grad, ao.grad)
    gclose(b.grad, bo.grad)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 3, 2, 3), (1, 5, 3, 70), (1, 8, 5, 67), (3, 17, 9, 33), (2, 7, 12, 260), (1, 40, 6, 4)])
def test_corr_ragged_shapes(shape):
    """The LDS-staged correlation kernels (csrc/ops_corr.hip) on shapes no PWC level has: single pixels, widths below one
    quad, widths that are not multiples of 4 (the dword staging path), more than one tile across (W > 64), channel counts
    below the channel split and not multiples of the backward's 8-channel groups -- forward and both gradients against the
    oracle's corr_naive on the host, each gradient also requested alone (one-sided launches)."""
    from unsupervised_depth_opticalflow_egomotion_amd.ops import corr81
    r = MG.rng(77)
    f1 = r.standard_normal(shape).astype(np.float32)
    f2 = r.standard_normal(shape).astype(np.float32)
    wgt = r.standard_normal((shape[0], 81, shape[2], shape[3])).astype(np.float32)
    ao, bo = T(f1, True), T(f2, True)
    co = O.corr_naive(ao, bo)
    (co * T(wgt)).sum().backward()
    a, b = G(f1, True), G(f2, True)
    cv = corr81(a, b)
    (cv * G(wgt)).sum().backward()
    close(cv, co, atol=1e-5)
    gclose(a.grad, ao.grad)
    gclose(b.grad, bo.grad)
    a1 = G(f1, True)
    (corr81(a1, G(f2)) * G(wgt)).sum().backward()
    gclose(a1.grad, ao.grad)          # (the one-sided launch may pick another tiling: same sums, another order)
    b1 = G(f2, True)
    (corr81(G(f1), b1) * G(wgt)).sum().backward()
    gclose(b1.grad, bo.grad)


@pytest.mark.parametrize("shape", [(3, 64, 128, 416), (2, 5, 37, 131), (1, 3, 8, 9), (2, 2, 1, 1), (1, 4, 2, 130)])
def test_stem_maxpool_is_aten_bit_for_bit(shape):
    """ops.maxpool3x3s2 (ResNet stem, depth_model.py:60-95) against F.max_pool2d(x, 3, 2, 1) on the host: values and
    gradients EQUAL, on post-ReLU inputs (whole windows of tied zeros: ATen keeps the first maximum in window scan
    order), odd sizes, windows cut by every border, -inf and NaN entries."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd.ops import maxpool3x3s2
    r = MG.rng(91)
    x = np.maximum(r.standard_normal(shape), 0).astype(np.float32)          # ~half the entries are exactly 0
    x[r.random(shape) < 0.01] = -np.inf
    if x.size > 50:
        x.reshape(-1)[r.integers(0, x.size, 5)] = np.nan
    wgt = r.standard_normal((shape[0], shape[1], (shape[2] - 1) // 2 + 1, (shape[3] - 1) // 2 + 1)).astype(np.float32)
    a, b = G(x, True), T(x, True)
    ya, yb = maxpool3x3s2(a), F.max_pool2d(b, 3, 2, 1)
    assert ya.shape == yb.shape
    np.testing.assert_array_equal(N(ya), N(yb))
    (ya * G(wgt)).sum().backward()
    (yb * T(wgt)).sum().backward()
    np.testing.assert_array_equal(N(a.grad), N(b.grad))


@pytest.mark.parametrize("shape", [(2, 128, 8, 26), (2, 96, 16, 52), (3, 32, 64, 208), (1, 20, 7, 10)])
def test_pwc_level_input(shape):
    """One PWC decoder level's input (pwc_tf.py:119-121) as one operator: equal to the composition of the per-op HIP
    operators (forward bit for bit; g_c1 / g_flow bit for bit -- two-term sums; g_c2 is a 64-bit fixed-point integer scatter: reproducible, held to the gradient tolerance here) and to the
    oracle's warp_flow + corr_naive + cat on the host."""
    from unsupervised_depth_opticalflow_egomotion_amd.ops import corr81, pwc_level_input, warp_flow
    B, C, H, W = shape
    r = MG.rng(57)
    c1 = r.standard_normal(shape).astype(np.float32)
    c2 = r.standard_normal(shape).astype(np.float32)
    flow = (r.standard_normal((B, 2, H, W)) * 2.5).astype(np.float32)
    wgt = r.standard_normal((B, 81 + C + 2, H, W)).astype(np.float32)
    a, b, f = G(c1, True), G(c2, True), G(flow, True)
    x = pwc_level_input(a, b, f)
    assert tuple(x.shape) == (B, 81 + C + 2, H, W)
    (x * G(wgt)).sum().backward()
    a2, b2, f2 = G(c1, True), G(c2, True), G(flow, True)
    x2 = torch.cat((corr81(a2, warp_flow(b2, f2, use_mask=False)), a2, f2), 1)
    (x2 * G(wgt)).sum().backward()
    assert torch.equal(x, x2)
    assert torch.equal(a.grad, a2.grad)
    assert torch.equal(f.grad, f2.grad)
    gclose(b.grad, b2.grad)
    ao, bo, fo = T(c1, True), T(c2, True), T(flow, True)
    xo = torch.cat((O.corr_naive(ao, O.warp_flow(bo, fo, use_mask=False)), ao, fo), 1)
    (xo * T(wgt)).sum().backward()
    close(x, xo, atol=1e-5)
    gclose(a.grad, ao.grad)
    gclose(b.grad, bo.grad)
    gclose(f.grad, fo.grad)
    # flow only (features detached): the g_c2 scatter is skipped
    f3 = G(flow, True)
    x3 = pwc_level_input(G(c1), G(c2), f3)
    (x3 * G(wgt)).sum().backward()
    assert torch.equal(f3.grad, f.grad)


@pytest.mark.parametrize("shape,flow_kind", [((2, 128, 8, 26), "smooth"), ((2, 96, 16, 52), "rough"), ((3, 32, 64, 208), "smooth"),
                                             ((1, 20, 7, 10), "rough"), ((2, 8, 5, 9), "zero"), ((2, 16, 12, 20), "collapse"),
                                             ((1, 196, 4, 13), "out")])
def test_pwc_level_map_built_in_the_forward_pass(shape, flow_kind, monkeypatch):
    """dfe_pwc_level_fwd_map / _bwd_map (the inverse map of the feature warp counted in the forward pass, three backward launches
    at every level) against dfe_pwc_level_fwd / _bwd (the map per backward call on the large levels, the 64-bit scatter on the
    small ones): outputs and ALL gradients bit for bit -- every contribution is rounded as the scatter rounds it and integer sums
    do not depend on the order of a list -- on smooth, rough, zero, out-of-view flows and a flow that sends every pixel to one
    target.  A second backward pass through the same graph (the bound of the first pass is still in the map) stays within the
    gradient tolerance."""
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    B, C, H, W = shape
    r = MG.rng(63)
    c1 = r.standard_normal(shape).astype(np.float32)
    c2 = r.standard_normal(shape).astype(np.float32)
    if flow_kind == "smooth":
        flow = (np.repeat(np.repeat(r.standard_normal((B, 2, (H + 3) // 4, (W + 3) // 4)), 4, 2), 4, 3)[:, :, :H, :W] * 2.0).astype(np.float32)
    elif flow_kind == "rough":
        flow = (r.standard_normal((B, 2, H, W)) * 4.0).astype(np.float32)
    elif flow_kind == "zero":
        flow = np.zeros((B, 2, H, W), np.float32)
    elif flow_kind == "out":
        flow = (r.standard_normal((B, 2, H, W)) * 30.0).astype(np.float32)
    else:       # every pixel samples (almost) the same point: one target pixel collects H*W contributions
        ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
        flow = np.stack([W / 2 + 0.25 - xs, H / 2 + 0.25 - ys])[None].repeat(B, 0).astype(np.float32)
    wgt = r.standard_normal((B, 81 + C + 2, H, W)).astype(np.float32)
    outs = {}
    for use_map in (True, False):
        monkeypatch.setattr(ops, "PWC_LEVEL_MAP", use_map)
        a, b, f = G(c1, True), G(c2, True), G(flow, True)
        x = ops.pwc_level_input(a, b, f)
        (x * G(wgt)).sum().backward(retain_graph=use_map)
        outs[use_map] = (x.detach().clone(), a.grad.clone(), b.grad.clone(), f.grad.clone())
        if use_map:
            g1 = b.grad.clone()
            b.grad = None
            (x * G(wgt * 0.37)).sum().backward()           # a second pass through the same map
            gclose(b.grad, 0.37 * g1)
    for u, v in zip(outs[True], outs[False]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("shape,out_hw,mult,pre", [((8, 2, 4, 13), (8, 26), 2.0, False), ((8, 2, 32, 104), (64, 208), 2.0, False),
                                                   ((2, 2, 64, 208), (256, 832), 4.0, True), ((2, 2, 8, 26), (32, 104), 4.0, True),
                                                   ((1, 3, 7, 9), (10, 31), 1.5, True), ((2, 1, 16, 20), (16, 20), 4.0, False)])
def test_resize_bilinear_differentiable(shape, out_hw, mult, pre):
    """ops.resize_bilinear (PWC_tf's flow up-sampling, pwc_tf.py:118-119,175-178) against F.interpolate on the host with
    the scalar on the same side: forward EQUAL (ATen-CPU association incl. the scalar's rounding), gradient 1e-6 of its
    scale (a gather in fixed order here, autograd's scatter there), and reproducible bit for bit run to run."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd.ops import resize_bilinear
    r = MG.rng(61)
    x = r.standard_normal(shape).astype(np.float32)
    wgt = r.standard_normal((shape[0], shape[1]) + tuple(out_hw)).astype(np.float32)
    a, b = G(x, True), T(x, True)
    ya = resize_bilinear(a, out_hw, mult, pre)
    yb = F.interpolate(b * mult, list(out_hw), mode="bilinear", align_corners=False) if pre else \
        F.interpolate(b, list(out_hw), mode="bilinear", align_corners=False) * mult
    np.testing.assert_array_equal(N(ya), N(yb))
    (ya * G(wgt)).sum().backward()
    (yb * T(wgt)).sum().backward()
    gclose(a.grad, b.grad, rel=1e-6, atol=1e-7)
    a2 = G(x, True)
    (resize_bilinear(a2, out_hw, mult, pre) * G(wgt)).sum().backward()
    assert torch.equal(a2.grad, a.grad)


@pytest.mark.parametrize("hw", [(256, 832), (375, 1242), (64, 208)])
def test_resize_oracle(hw):
    from unsupervised_depth_opticalflow_egomotion_amd.ops import resize
    import torch.nn.functional as F
    r = MG.rng(41)
    h, w = hw
    img = r.random((2, 3, h, w)).astype(np.float32)
    for s in range(1, 4):
        oh, ow = int(h / 2 ** s), int(w / 2 ** s)
        close(resize(G(img), (oh, ow), "bilinear"), F.interpolate(T(img), (oh, ow), mode="bilinear", align_corners=False), atol=1e-6)
        close(resize(G(img), (oh, ow), "area"), F.interpolate(T(img), (oh, ow), mode="area"), atol=1e-6)


def test_degenerate_sizes():
    """W == 1 / H == 1 (max(W-1,1) normalisation, net_utils.py:42-43), 2-px planes, single channel."""
    from unsupervised_depth_opticalflow_egomotion_amd.structures import warp_flow
    from unsupervised_depth_opticalflow_egomotion_amd.pytorch_ssim import SSIM
    r = MG.rng(91)
    for shape in [(1, 1, 1, 9), (2, 2, 7, 1), (1, 3, 2, 2), (1, 1, 1, 1)]:
        b, c, h, w = shape
        x = r.random(shape).astype(np.float32)
        fl = (0.7 * r.standard_normal((b, 2, h, w))).astype(np.float32)
        for ac in (False, True):
            for um in (False, True):
                yo = O.warp_flow(T(x), T(fl), use_mask=um, align_corners=ac)
                y = warp_flow(G(x), G(fl), use_mask=um, align_corners=ac)
                close(y, yo, atol=2e-6)
        close(SSIM(G(x), G(x[::-1].copy())), O.SSIM(T(x), T(x[::-1].copy())), atol=2e-5)


# ---------------------------------------------------------------------------------------------- depth-decoder glue
# Reference = the plain PyTorch fp32 composition on the CPU (what depth_model.py's ConvBlock / decoder stage does
# between its convolutions).  Tolerance: 2e-6 abs on O(1) activations (expm1 / exp are within 1 ulp of ATen's),
# gradients 1e-5 of their scale (sums of <= 16 products in a different association order).
def _ref_elu_pad(x, bias, apply_elu):
    import torch.nn.functional as F
    if bias is not None:
        x = x + bias[None, :, None, None]
    return F.pad(F.elu(x) if apply_elu else x, (1, 1, 1, 1), mode="reflect")


def _ref_up2_cat_pad(x, bias, skip):
    import torch.nn.functional as F
    if bias is not None:
        x = x + bias[None, :, None, None]
    u = F.interpolate(F.elu(x), scale_factor=2, mode="bilinear", align_corners=False)
    if skip is not None:
        u = torch.cat([u, skip], 1)
    return F.pad(u, (1, 1, 1, 1), mode="reflect")


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("apply_elu", [True, False])
@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 4, 2, 2), (3, 2, 3, 64), (2, 16, 33, 129)])
def test_elu_pad(shape, apply_elu, with_bias):
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(*shape).astype(np.float32) * 2.0
    bias = rng.randn(shape[1]).astype(np.float32)
    r = rng.randn(shape[0], shape[1], shape[2] + 2, shape[3] + 2).astype(np.float32)
    xh, xo = G(x, True), T(x).requires_grad_(True)
    bh, bo = (G(bias, True), T(bias).requires_grad_(True)) if with_bias else (None, None)
    yh = ops.elu_pad(xh, bh, apply_elu)
    yo = _ref_elu_pad(xo, bo, apply_elu)
    (yh * G(r)).sum().backward()
    (yo * T(r)).sum().backward()
    close(yh, yo, atol=2e-6, rtol=2e-6)
    gclose(xh.grad, xo.grad, rel=1e-5)
    if with_bias:
        gclose(bh.grad, bo.grad, rel=2e-5, atol=1e-6)


@pytest.mark.parametrize("shape,c2", [((2, 3, 4, 6), 2), ((1, 2, 1, 1), 1), ((2, 5, 8, 26), 0), ((1, 16, 33, 65), 7),
                                      ((2, 3, 19, 300), 0), ((1, 2, 5, 257), 1)])   # w >= 256: the rolling-window backward
def test_elu_up2_cat_pad(shape, c2):
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    b, c1, h, w = shape
    rng = np.random.RandomState(sum(shape) + c2)
    x = rng.randn(*shape).astype(np.float32) * 2.0
    sk = rng.randn(b, c2, 2 * h, 2 * w).astype(np.float32) if c2 else None
    r = rng.randn(b, c1 + c2, 2 * h + 2, 2 * w + 2).astype(np.float32)
    xh, xo = G(x, True), T(x).requires_grad_(True)
    sh = G(sk, True) if c2 else None
    so = T(sk).requires_grad_(True) if c2 else None
    bias = rng.randn(c1).astype(np.float32)
    bh, bo = (G(bias, True), T(bias).requires_grad_(True)) if (c1 + c2) % 2 else (None, None)   # both variants over the cases
    yh = ops.elu_up2_cat_pad(xh, bh, sh)
    yo = _ref_up2_cat_pad(xo, bo, so)
    (yh * G(r)).sum().backward()
    (yo * T(r)).sum().backward()
    close(yh, yo, atol=2e-6, rtol=2e-6)
    gclose(xh.grad, xo.grad, rel=1e-5)
    if c2:
        gclose(sh.grad, so.grad, rel=1e-6)
    if bh is not None:
        gclose(bh.grad, bo.grad, rel=2e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(8, 96, 64, 208), (2, 32, 64, 208), (2, 96, 4, 13), (3, 8, 7, 70), (1, 16, 1, 5)])
def test_flow_head_matches_miopen(shape):
    """PWC's predict_flow = Conv2d(C, 2, 3, 1, 1) (pwc_tf.py:39-40) on the rolling-window head kernels against ATen's
    convolution: forward 1e-5 of the output scale, the three gradients 1e-4 of theirs (fp32 sums of C*9 / H*W*B terms in a
    different order); ragged sizes, one row, strips that end mid-wave."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    b, c, h, w = shape
    r = MG.rng(500 + c + h)
    x = r.standard_normal(shape).astype(np.float32)
    wt = (r.standard_normal((2, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    bs = r.standard_normal(2).astype(np.float32)
    gy = r.standard_normal((b, 2, h, w)).astype(np.float32)
    xt, wtt, bt = G(x, True), G(wt, True), G(bs, True)
    assert ops.flow_head_eligible(xt, wtt, bt)
    y = ops.FlowHeadFn.apply(xt, wtt, bt)
    y.backward(G(gy))
    xr, wr, br = G(x, True), G(wt, True), G(bs, True)
    yr = F.conv2d(xr, wr, br, 1, 1)
    yr.backward(G(gy))
    gclose(y, yr, rel=1e-5)
    gclose(xt.grad, xr.grad, rel=1e-4)
    gclose(wtt.grad, wr.grad, rel=1e-4)
    gclose(bt.grad, br.grad, rel=1e-4)
    # reproducible: fixed-order partial sums
    x2, w2, b2 = G(x, True), G(wt, True), G(bs, True)
    ops.FlowHeadFn.apply(x2, w2, b2).backward(G(gy))
    assert torch.equal(w2.grad, wtt.grad) and torch.equal(b2.grad, bt.grad) and torch.equal(x2.grad, xt.grad)


@pytest.mark.parametrize("cfg", [(2, 16, 24, 64, 64, 8), (1, 8, 12, 32, 48, 16), (2, 8, 8, 16, 32, 2)])
def test_dilated_conv_on_phase_images_matches_the_dilated_call(cfg):
    """convs.conv2d sends a dilated 3x3 'same' convolution (PWC's context network, pwc_tf.py:31-36) to MIOpen as a dense
    3x3 convolution on the d*d phase images: the same products and sums, another kernel -- values and gradients agree
    with the dilated call to fp32 rounding."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import convs
    b, ci, co, h, w, d = cfg
    r = MG.rng(900 + d)
    x = r.standard_normal((b, ci, h, w)).astype(np.float32)
    wt = (r.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    gy = r.standard_normal((b, co, h, w)).astype(np.float32)
    old = convs.PHASE_MIN_DILATION
    convs.PHASE_MIN_DILATION = 2
    try:
        xt, wtt = G(x, True), G(wt, True)
        assert convs._phase_eligible(xt, wtt, (1, 1), (d, d), (d, d), 1)
        y = convs.conv2d(xt, wtt, None, 1, d, d)
        y.backward(G(gy))
    finally:
        convs.PHASE_MIN_DILATION = old
    xr, wr = G(x, True), G(wt, True)
    yr = F.conv2d(xr, wr, None, 1, d, d)
    yr.backward(G(gy))
    gclose(y, yr, rel=2e-5)
    gclose(xt.grad, xr.grad, rel=1e-4)
    gclose(wtt.grad, wr.grad, rel=1e-4)


def test_decoder_glue_argument_errors():
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    x = torch.zeros(1, 2, 4, 4, device=dev())
    with pytest.raises(ValueError):          # skip must be exactly twice the size
        ops.elu_up2_cat_pad(x, None, torch.zeros(1, 2, 7, 8, device=dev()))
    with pytest.raises(DfeError):            # 1-pixel planes cannot be reflection-padded
        ops.elu_pad(torch.zeros(1, 1, 1, 5, device=dev()))
    with pytest.raises(DfeError):            # no CPU fallback
        ops.elu_pad(torch.zeros(1, 1, 4, 4))


def test_depth_decoder_fused_matches_aten_graph():
    """DepthDecoder.forward_fused (HIP glue) against the same module evaluated through ATen's
    elu / interpolate / cat / ReflectionPad2d on the same device and weights: disparities and every parameter
    gradient.  Both run MIOpen convolutions; tolerance 1e-5 abs on sigmoid outputs, 2e-4 of the gradient scale."""
    from unsupervised_depth_opticalflow_egomotion_amd.networks.depth_model import DepthDecoder
    torch.manual_seed(3)
    enc_ch = np.array([64, 64, 128, 256, 512])
    dec = DepthDecoder(enc_ch, scales=range(3)).to(dev())
    b, h, w = 2, 64, 96
    feats = [torch.randn(b, int(c), h // 2 ** (i + 1), w // 2 ** (i + 1), device=dev()).relu_() for i, c in enumerate(enc_ch)]
    res = {}
    for name in ("fused", "aten"):
        dec.zero_grad(set_to_none=True)
        fin = [f.clone().requires_grad_(True) for f in feats]
        out = dec.forward_fused(fin) if name == "fused" else dec.forward_aten(fin)
        loss = sum((out[s] * (s + 1.0)).mean() for s in out)
        loss.backward()
        res[name] = ({s: N(out[s]) for s in out}, {k: N(p.grad) for k, p in dec.named_parameters()}, [N(f.grad) for f in fin])
    for s in res["aten"][0]:
        close(res["fused"][0][s], res["aten"][0][s], atol=1e-5, rtol=1e-5)
    for k in res["aten"][1]:
        gclose(res["fused"][1][k], res["aten"][1][k], rel=2e-4, atol=1e-9)
    for a, c in zip(res["fused"][2], res["aten"][2]):
        gclose(a, c, rel=2e-4, atol=1e-9)



# ---------------------------------------------------------------------------------------------- conv epilogue
@pytest.mark.parametrize("slope", [0.1, 0.0, 1.0])
@pytest.mark.parametrize("shape", [(2, 5, 8, 26), (1, 3, 7, 9), (4, 16, 64, 208), (2, 2, 1, 1)])
def test_bias_act(shape, slope):
    """act(z + bias) in place and its backward (gz, gbias) against F.leaky_relu(z + bias) on the CPU: values are the
    same two IEEE operations (exact), gz exact, gbias a sum in a different order (1e-5 of its scale)."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    rng = np.random.RandomState(sum(shape))
    z = rng.randn(*shape).astype(np.float32)
    bias = rng.randn(shape[1]).astype(np.float32)
    r = rng.randn(*shape).astype(np.float32)
    zh, bh = G(z, True), G(bias, True)
    zo, bo = T(z).requires_grad_(True), T(bias).requires_grad_(True)
    yh = ops.bias_act(zh * 1.0, bh, slope)           # * 1.0: the op works in place on a non-leaf tensor
    yo = F.leaky_relu(zo + bo[None, :, None, None], slope)
    (yh * G(r)).sum().backward()
    (yo * T(r)).sum().backward()
    np.testing.assert_array_equal(N(yh), N(yo))
    np.testing.assert_array_equal(N(zh.grad), N(zo.grad))
    gclose(bh.grad, bo.grad, rel=1e-5, atol=1e-6)


def test_bias_act_reads_cat_gradient_slices_in_place():
    """The gradient arriving from a torch.cat is a channel slice of a wider tensor: same result as the contiguous path."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    rng = np.random.RandomState(5)
    z, other, bias = rng.randn(2, 4, 8, 12).astype(np.float32), rng.randn(2, 3, 8, 12).astype(np.float32), rng.randn(4).astype(np.float32)
    r = rng.randn(2, 7, 8, 12).astype(np.float32)
    zh, bh = G(z, True), G(bias, True)
    yh = torch.cat([G(other), ops.bias_act(zh * 1.0, bh, 0.1)], 1)
    (yh * G(r)).sum().backward()
    zo, bo = T(z).requires_grad_(True), T(bias).requires_grad_(True)
    yo = torch.cat([T(other), F.leaky_relu(zo + bo[None, :, None, None], 0.1)], 1)
    (yo * T(r)).sum().backward()
    np.testing.assert_array_equal(N(zh.grad), N(zo.grad))
    gclose(bh.grad, bo.grad, rel=1e-5, atol=1e-6)


def test_conv_act_module_matches_aten():
    """net_utils.conv() on the GPU (MIOpen conv without bias + HIP epilogue) against the same module through ATen's
    conv + LeakyReLU on the same device: output 1e-5, weight / bias / input gradients 2e-4 of their scale."""
    import torch.nn as nn
    from unsupervised_depth_opticalflow_egomotion_amd.structures.net_utils import conv
    torch.manual_seed(1)
    m = conv(12, 20, kernel_size=3, stride=1, padding=2, dilation=2).to(dev())
    x = torch.randn(2, 12, 32, 52, device=dev())
    res = []
    for fused in (True, False):
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = m(xi) if fused else nn.Sequential.forward(m, xi)
        (y * y).mean().backward()
        res.append((N(y), N(m[0].weight.grad), N(m[0].bias.grad), N(xi.grad)))
    close(res[0][0], res[1][0], atol=1e-5, rtol=1e-5)
    for a, b in zip(res[0][1:], res[1][1:]):
        gclose(a, b, rel=2e-4, atol=1e-9)


# ---------------------------------------------------------------------------------------------- grouped batch norm
def _ref_grouped_bn(x, res, bn, groups, relu):
    """``groups`` sequential nn.BatchNorm2d calls on the host (what the reference's per-frame calls do)."""
    import torch.nn.functional as F
    y = torch.cat([torch.nn.BatchNorm2d.forward(bn, c) for c in x.chunk(groups, 0)], 0)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("shape,groups", [((6, 5, 8, 26), 3), ((4, 3, 7, 9), 1), ((12, 16, 64, 208), 3), ((2, 4, 1, 1), 1),
                                          ((12, 8, 32, 104), 3), ((12, 6, 16, 52), 3), ((4, 7, 64, 64), 1),   # single-kernel small-plane path
                                          ((15, 4, 8, 28), 3), ((4, 3, 64, 68), 2)])                         # Bg = 5 / HW > 4096: three-kernel path
def test_grouped_batch_norm(shape, groups, relu, with_res):
    """Fused grouped BatchNorm (+ residual + ReLU) against sequential nn.BatchNorm2d calls on the CPU: outputs 2e-5
    (statistics merged in double vs ATen's accumulation), running statistics 1e-6, gradients 1e-4 of their scale."""
    from unsupervised_depth_opticalflow_egomotion_amd.networks.resnet import FrameBatchNorm2d
    rng = np.random.RandomState(sum(shape) + groups)
    x = (rng.randn(*shape) * 1.5 + 0.7).astype(np.float32)
    res = rng.randn(*shape).astype(np.float32) if with_res else None
    r = rng.randn(*shape).astype(np.float32)
    w, b = rng.rand(shape[1]).astype(np.float32) + 0.5, rng.randn(shape[1]).astype(np.float32)
    mods = []
    for device in ("cuda", "cpu"):
        m = FrameBatchNorm2d(shape[1])
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        m.groups = groups
        mods.append(m.to(device).train())
    mh, mo = mods
    xh, xo = G(x, True), T(x).requires_grad_(True)
    rh = G(res, True) if with_res else None
    ro = T(res).requires_grad_(True) if with_res else None
    yh = mh(xh, residual=rh, relu=relu)
    yo = _ref_grouped_bn(xo, ro, mo, groups, relu)
    (yh * G(r)).sum().backward()
    (yo * T(r)).sum().backward()
    close(yh, yo, atol=2e-5, rtol=2e-5)
    close(mh.running_mean, mo.running_mean, atol=1e-6, rtol=1e-6)
    close(mh.running_var, mo.running_var, atol=1e-6, rtol=1e-5)
    assert int(mh.num_batches_tracked) == int(mo.num_batches_tracked) == groups
    # the ReLU mask can differ where |y| is within rounding of 0 (expected ~1 of the 2.5 M elements of the largest
    # case): such a pixel moves its own gradient and its channel's weight / bias gradient by O(1) -- budget two
    flips = 2 if relu else 0
    gclose(xh.grad, xo.grad, rel=1e-4, max_outliers=flips, atol=1e-6)
    gclose(mh.weight.grad, mo.weight.grad, rel=1e-4, atol=1e-5, max_outliers=flips)
    gclose(mh.bias.grad, mo.bias.grad, rel=1e-4, atol=1e-5, max_outliers=flips)
    if with_res:
        gclose(rh.grad, ro.grad, rel=1e-6, max_outliers=2 if relu else 0)


def test_depth_net_frames_match_sequential_calls():
    """Depth_Model.forward_frames (one batch of 3B, grouped BatchNorm, fused glue) on the GPU against three sequential
    calls of the same net on the CPU (the reference's call pattern): disparities 1e-4, BatchNorm running statistics
    1e-5, parameter gradients 2e-3 of their scale (MIOpen vs host convolutions and a few ReLU decisions at rounding
    distance from 0, through ~30 layers)."""
    from unsupervised_depth_opticalflow_egomotion_amd.networks.depth_model import Depth_Model
    torch.manual_seed(11)
    net_c = Depth_Model(3).train()
    net_g = Depth_Model(3).train()
    net_g.load_state_dict(net_c.state_dict())
    net_g.to(dev())
    frames = [torch.rand(2, 3, 64, 128) for _ in range(3)]
    out_c = [net_c(f) for f in frames]
    out_g = net_g.forward_frames([f.to(dev()) for f in frames])
    wts = [1.0, 0.7, 1.3]
    sum(wts[i] * o.mean() for i in range(3) for o in out_c[i]).backward()
    sum(wts[i] * o.mean() for i in range(3) for o in out_g[i]).backward()
    for i in range(3):
        for a, c in zip(out_g[i], out_c[i]):
            close(a, c, atol=1e-4, rtol=1e-4)
    sd_c, sd_g = net_c.state_dict(), net_g.state_dict()
    for k in sd_c:
        if "running_" in k:
            close(sd_g[k], sd_c[k], atol=1e-5, rtol=1e-4)
        if "num_batches_tracked" in k:
            assert int(sd_g[k]) == int(sd_c[k]) == 3
    pc, pg = dict(net_c.named_parameters()), dict(net_g.named_parameters())
    for k in pc:
        if pc[k].grad is None:
            assert pg[k].grad is None, k
            continue
        gclose(pg[k].grad, pc[k].grad, rel=2e-3, atol=1e-6)


# ---------------------------------------------------------------------------------------------- disparity head
@pytest.mark.parametrize("shape", [(2, 16, 5, 7), (1, 32, 20, 130), (3, 16, 33, 61), (2, 64, 16, 52), (12, 16, 64, 208)])
def test_disp_head(shape):
    """sigmoid(conv3x3(p) + b) and its backward against F.conv2d + sigmoid on the CPU: disparities 2e-6, gradients
    1e-4 of their scale (sums of 9*C products, and of all pixels for the weight gradient, in another order)."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    B, C, H, W = shape
    rng = np.random.RandomState(sum(shape))
    p = rng.randn(B, C, H + 2, W + 2).astype(np.float32)
    w = (rng.randn(1, C, 3, 3) * 0.2).astype(np.float32)
    bias = rng.randn(1).astype(np.float32)
    r = rng.randn(B, 1, H, W).astype(np.float32)
    ph, wh, bh = G(p, True), G(w, True), G(bias, True)
    po, wo, bo = T(p).requires_grad_(True), T(w).requires_grad_(True), T(bias).requires_grad_(True)
    yh = ops.disp_head(ph, wh, bh)
    yo = torch.sigmoid(F.conv2d(po, wo, bo))
    (yh * G(r)).sum().backward()
    (yo * T(r)).sum().backward()
    close(yh, yo, atol=2e-6, rtol=2e-6)
    gclose(ph.grad, po.grad, rel=1e-4, atol=1e-7)
    gclose(wh.grad, wo.grad, rel=1e-4, atol=1e-5)
    gclose(bh.grad, bo.grad, rel=1e-4, atol=1e-5)


def test_disp_head_argument_errors():
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    with pytest.raises(DfeError):            # channel count must be a multiple of 16
        ops.disp_head(torch.zeros(1, 8, 6, 6, device=dev()), torch.zeros(1, 8, 3, 3, device=dev()), None)
    with pytest.raises(ValueError):
        ops.disp_head(torch.zeros(1, 16, 6, 6, device=dev()), torch.zeros(2, 16, 3, 3, device=dev()), None)


def test_net_glue_many_planes():
    """B*C above 65535 (the grid is (chunks, C, B), not (chunks, B*C)): bias_act, elu_pad and grouped batch norm on
    70 x 1024 planes of 2x3 pixels against the host composition."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    rng = np.random.RandomState(9)
    B, C = 70, 1024
    x = rng.randn(B, C, 2, 3).astype(np.float32)
    bias = rng.randn(C).astype(np.float32)
    y = ops.bias_act(G(x) * 1.0, G(bias), 0.1)
    np.testing.assert_array_equal(N(y), N(F.leaky_relu(T(x) + T(bias)[None, :, None, None], 0.1)))
    close(ops.elu_pad(G(x), G(bias), True), _ref_elu_pad(T(x), T(bias), True), atol=2e-6, rtol=2e-6)
    w = rng.rand(C).astype(np.float32) + 0.5
    rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    yb = ops.grouped_batch_norm(G(x), G(w), G(bias), rm, rv, groups=2, relu=True)
    ref = torch.cat([F.batch_norm(c, None, None, T(w), T(bias), True) for c in T(x).chunk(2, 0)], 0).relu()
    close(yb, ref, atol=2e-5, rtol=2e-5)


# ---------------------------------------------------------------------------------------------- MFMA weight gradient
@pytest.mark.parametrize("shape", [(2, 16, 16, 8, 32), (1, 32, 16, 5, 16), (2, 48, 32, 7, 48), (3, 96, 32, 16, 64), (2, 16, 32, 33, 80)])
def test_wgrad3x3_mfma(shape):
    """dfe_wgrad3x3_fwd (fp32 MFMA, exact fma chains) against the weight gradient of F.conv2d in fp64 on the CPU:
    1e-6 of the gradient scale (sums of up to B*H*W products in fp32)."""
    import ctypes
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib, ptr, stream_ptr, check
    lib = get_lib()
    B, Ci, Co, H, W = shape
    rng = np.random.RandomState(sum(shape))
    p = rng.randn(B, Ci, H + 2, W + 2).astype(np.float32)
    gy = rng.randn(B, Co, H, W).astype(np.float32)
    w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(torch.from_numpy(p).double(), w) * torch.from_numpy(gy).double()).sum().backward()
    ph, gh = G(p), G(gy)
    gw = torch.empty(Co, Ci, 3, 3, device=dev())
    part = torch.empty(lib.dfe_wgrad3x3_partials_floats(B, Ci, Co, H, W), device=dev())
    check(lib.dfe_wgrad3x3_fwd(ptr(ph), ptr(gh), ptr(gw), ptr(part), B, Ci, Co, H, W, stream_ptr()), "dfe_wgrad3x3_fwd")
    gclose(gw, w.grad.float(), rel=1e-6)
    gw2 = torch.empty_like(gw)
    check(lib.dfe_wgrad3x3_fwd(ptr(ph), ptr(gh), ptr(gw2), ptr(part), B, Ci, Co, H, W, stream_ptr()), "dfe_wgrad3x3_fwd")
    assert torch.equal(gw, gw2), "the weight gradient must be bitwise reproducible"


def test_thin_conv_function_and_eligibility():
    """ops.conv3x3_valid: the MFMA weight gradient is used for the thin full-resolution layers only; values and both
    gradients agree with F.conv2d on the same device (MIOpen) to 1e-5 / 2e-4 of the gradient scale."""
    import torch.nn.functional as F
    from unsupervised_depth_opticalflow_egomotion_amd import ops
    torch.manual_seed(2)
    p = torch.randn(2, 32, 130, 418, device=dev())
    w = torch.randn(16, 32, 3, 3, device=dev()) * 0.1
    assert ops.thin_conv3x3_eligible(p, w)
    assert not ops.thin_conv3x3_eligible(torch.zeros(2, 64, 34, 106, device=dev()), torch.zeros(32, 64, 3, 3, device=dev()))
    assert not ops.thin_conv3x3_eligible(p, torch.zeros(64, 32, 3, 3, device=dev()))
    for pp, ww in ((p, w), (torch.randn(2, 16, 66, 210, device=dev()), torch.randn(16, 16, 3, 3, device=dev()) * 0.1)):
        res = []                      # 32 -> 16: MFMA weight gradient only; 16 -> 16: all three passes on the matrix cores
        for fn in (ops.conv3x3_valid, F.conv2d):
            pi, wi = pp.clone().requires_grad_(True), ww.clone().requires_grad_(True)
            y = fn(pi, wi)
            (y * y).mean().backward()
            res.append((N(y), N(pi.grad), N(wi.grad)))
        close(res[0][0], res[1][0], atol=1e-5, rtol=1e-5)
        gclose(res[0][1], res[1][1], rel=2e-4, atol=1e-9)
        gclose(res[0][2], res[1][2], rel=2e-4, atol=1e-9)




@pytest.mark.gpu
@pytest.mark.parametrize("ac", [False, True])
def test_legacy_inverse_warp_euler_golden(golden_dir, ac):
    """The first-generation ``inverse_warp`` (inverse_warp.py:190-224) in Euler mode = rigid flow + bilinear warp on the HIP
    operators, against golden G12 (the reference's function): image 2e-5, validity equal away from the +-1 border (the
    reference thresholds 2U/(w-1)-1, the composition (U-x)+x: decisions within 1e-5 of the border may differ), gradients
    1e-4 of scale."""
    import os
    from tests.golden import make_golden as MG
    from unsupervised_depth_opticalflow_egomotion_amd.structures import inverse_warp as iw
    g = np.load(os.path.join(golden_dir, "G12_ac%d.npz" % ac))
    c = MG.g12_inputs()
    dev = torch.device("cuda:0")
    T = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)   # noqa: E731
    d, p = T(c["depth"], True), T(c["pose"], True)
    y, valid = iw.inverse_warp(T(c["img"]), d, p, T(c["K"]), rotation_mode="euler", align_corners=ac)
    (y * T(c["wgt"])).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["iw_euler_img"], rtol=0, atol=2e-5)
    want = np.unpackbits(g["iw_euler_valid"])[:valid.numel()].reshape(valid.shape).astype(bool)
    grid = np.abs(g["cam2pixel"]).max(-1)
    safe = np.abs(grid - 1.0) > 1e-5
    assert valid.dtype == torch.bool and np.array_equal(valid.cpu().numpy()[safe], want[safe]) and safe.mean() > 0.99
    for mine, key in ((d.grad, "iw_euler_gdepth"), (p.grad, "iw_euler_gpose")):
        np.testing.assert_allclose(mine.cpu().numpy(), g[key], rtol=1e-4, atol=1e-4 * np.abs(g[key]).max())
    # quaternion mode on the device (tensor expressions + grid_sample)
    d2, p2 = T(c["depth"], True), T(c["pose"], True)
    y2, v2 = iw.inverse_warp(T(c["img"]), d2, p2, T(c["K"]), rotation_mode="quat", align_corners=ac)
    np.testing.assert_allclose(y2.detach().cpu().numpy(), g["iw_quat_img"], rtol=0, atol=2e-5)
