"""GPU parity of the fused loss stack (dfe_geom_loss_fwd/bwd through the C ABI) against the oracle's
restatement of model_geometry.py:797-951 and against the golden vectors captured from the reference.

Contract (fp32), measured headroom in brackets (MI355X, EPYC host; tools/parity_report.py prints the flip counts and
largest errors per case -> profiles/r03_parity_report.txt):
  * masks: BIT-EXACT.  The kernels evaluate the reference's expressions in the reference's association order
    (incl. the FMA patterns of its BLAS / ATen kernels) on identical inputs, so every mask decision is taken on
    identical bits; the one transcendental feeding a decision is the softmax's exp (<= 1-2 ulp between libraries),
    so an occlusion bit may differ only at a pixel whose margin |w - 0.48| < 2e-7 (tests/_margins.py).  The seeds in
    STRICT are margin-checked (no pixel inside that floor, re-checked on the host in every run): for them all eight
    masks at every scale must EQUAL the oracle's.  Elsewhere a differing pixel must lie inside the floor.  [0 flips
    in every case tried, also at B=4 256x832 and B=2 375x1242 S=6]
  * loss vectors: 5e-6 relative on all eight rows (reduction order only) [<= 7e-7]
  * gradients: 1e-4 of the gradient's scale, element-wise, no outliers on STRICT seeds [<= 2e-5];
    pose gradient 2e-5 [<= 3e-6]."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_stack_oracle as O
from tests import _margins as M
from tests.golden import make_golden as MG
from unsupervised_depth_opticalflow_egomotion_amd import synthetic

pytestmark = pytest.mark.gpu
T, N = MG.T, MG.N

# margin-checked seeds per (B, H, W) at S = 3, valid for both align_corners modes (tests/test_api_cpu.py re-derives this)
STRICT = {(2, 32, 96): 932, (2, 64, 208): 77, (2, 128, 448): 600, (1, 256, 832): 1219}


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).float().to(dev())
    return t.requires_grad_(True) if grad else t


def to_dev(inp, grad=True):
    disps = [[G(a, grad) for a in lst] for lst in inp.disps]
    return disps, G(inp.pose, grad), [G(a, grad) for a in inp.flows_bwd], [G(a, grad) for a in inp.flows_fwd]


def run_hip(inp, ac, S, weights=None, depth_terms=False):
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import geom_loss_stack
    disps, pose, fb, ff = to_dev(inp)
    il, it, ir = [G(a) for a in inp.imgs]
    lp, masks = geom_loss_stack(il, it, ir, disps[0], disps[1], disps[2], pose, fb, ff, G(inp.K), G(inp.K_inv),
                                num_scales=S, align_corners=ac, return_masks=True, enable_depth_ssim=depth_terms,
                                enable_depth_consis=depth_terms)
    w = weights or MG.GEOM_WEIGHTS
    total = sum(w[k] * v.mean() for k, v in lp.items())
    total.backward()
    return lp, masks, total, (disps, pose, fb, ff)


def run_oracle(inp, ac, S, weights=None, depth_terms=False):
    m = O.GeomLossOracle(num_scales=S, align_corners=ac)
    disps, pose, fb, ff = MG.lists_to_t(inp, True)
    il, it, ir = [T(a) for a in inp.imgs]
    lp, masks = m.geom_losses(il, it, ir, disps[0], disps[1], disps[2], pose, fb, ff, T(inp.K), T(inp.K_inv),
                              enable_depth_ssim=depth_terms, enable_depth_consis=depth_terms)
    w = weights or MG.GEOM_WEIGHTS
    total = sum(w[k] * v.mean() for k, v in lp.items())
    total.backward()
    return lp, masks, total, (disps, pose, fb, ff)


MASKS = ("valid_bwd", "valid_fwd", "occ_bwd", "occ_fwd", "dyna_bwd", "dyna_fwd", "texture_bwd", "texture_fwd")


def check_masks(mk_h, mk_o, margins, S, strict, names=MASKS):
    """Every differing pixel must lie inside its mask's fp32 noise floor; returns (nflip, npx, within)."""
    within = M.within_counts({k: margins[k] for k in names})
    nflip, npx = 0, 0
    for k in names:
        tau = M.TAU[M.family(k)]
        for s in range(S):
            a, b = N(mk_h[k][s]), N(mk_o[k][s])
            assert a.shape == b.shape
            f = a != b
            nflip += int(f.sum())
            npx += a.size
            assert (margins[k][s][f] < tau).all(), "%s scale %d: %d differing pixel(s) outside the noise floor %g " \
                "(largest margin %g)" % (k, s, int(f.sum()), tau, float(margins[k][s][f].max()))
    print("masks: %d differing of %d decisions; %d pixel(s) inside the noise floor" % (nflip, npx, sum(within.values())))
    if strict:
        assert sum(within.values()) == 0, "seed is not margin-checked: %r" % (within,)
        assert nflip == 0
    return nflip, npx, within


def compare(inp, ac, S, weights=None, strict=False, depth_terms=False):
    lp_h, mk_h, tot_h, (dh, ph, fbh, ffh) = run_hip(inp, ac, S, weights, depth_terms)
    lp_o, mk_o, tot_o, (do, po, fbo, ffo) = run_oracle(inp, ac, S, weights, depth_terms)
    assert ("loss_depth_ssim" in lp_h) == depth_terms and ("loss_depth_consis" in lp_h) == depth_terms
    nflip, npx, _ = check_masks(mk_h, mk_o, M.geom_margins(inp, ac, S), S, strict)
    for k in lp_h:
        np.testing.assert_allclose(N(lp_h[k]), N(lp_o[k]), rtol=5e-6 + 4.0 * nflip / npx, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(N(tot_h), N(tot_o), rtol=5e-6 + 4.0 * nflip / npx)

    # element-wise gradients: 1e-4 of the gradient's scale; only a flipped pixel may move its neighbourhood
    def gcmp(a, b, name, rel=1e-4, per_flip=12, extra=0):
        a, b = N(a.grad), N(b.grad)
        scale = max(np.abs(b).max(), 1e-12)
        bad = int((np.abs(a - b) > rel * scale + 1e-9).sum())
        assert bad <= per_flip * nflip + extra, "%s: %d bad elements (nflip=%d) max diff %g scale %g" % (
            name, bad, nflip, np.abs(a - b).max(), scale)
    for f in range(3):
        for s in range(S):
            gcmp(dh[f][s], do[f][s], "gdisp_%d_%d" % (f, s))
    for s in range(S):
        gcmp(fbh[s], fbo[s], "gflow_b_%d" % s)
        gcmp(ffh[s], ffo[s], "gflow_f_%d" % s)
    gp_h, gp_o = N(ph.grad), N(po.grad)
    tol = 2e-5 if nflip == 0 else 2e-2
    assert np.abs(gp_h - gp_o).max() <= tol * max(np.abs(gp_o).max(), 1e-12) + 1e-7, (gp_h, gp_o, nflip)
    return nflip, npx


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("shape", [(2, 32, 96), (2, 128, 448), (1, 256, 832)])
def test_fused_stack_vs_oracle(shape, ac):
    """Margin-checked seeds: masks EQUAL, losses 5e-6, gradients 1e-4 / 2e-5 (pose)."""
    b, h, w = shape
    inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=STRICT[shape])
    compare(inp, ac, 3, strict=True)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("shape", [(2, 128, 448), (4, 256, 832)])
def test_fused_stack_with_depth_terms_vs_oracle(shape, ac):
    """dfe_geom_args.depth_terms: all ten loss vectors -- the eight active ones plus the depth
    SSIM and depth-consistency terms the reference keeps commented (model_geometry.py:889-891,897-899), weighted as
    the reference's config weights them -- and the gradients of their sum (incl. the source disparities, which only
    the consistency term reaches through the sampled projected depth) against the oracle, same tolerances."""
    b, h, w = shape
    strict = shape in STRICT
    inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=STRICT.get(shape, 1234))
    compare(inp, ac, 3, strict=strict, depth_terms=True)


@pytest.mark.parametrize("ac", [False, True])
def test_fused_stack_baseline_batch(ac):
    """BASELINE configs[2] size, B=4 256x832 S=3, against the oracle (1.1 M decisions per mask family: a few pixels
    always sit inside the occlusion noise floor, so the per-pixel rule applies instead of a margin-checked seed)."""
    compare(synthetic.make_loss_stack_inputs(4, 256, 832, 3, seed=1234), ac, 3)


def raw_pose_inputs(kind, shape, seed):
    """Loss-stack inputs whose poses are NOT conditioned: ``sigma`` = raw Gaussian 6-vectors (rotations sigma 0.05 rad,
    as golden G2 draws them), ``posecnn`` = the output of a seeded, randomly initialised PoseCNN on the frames."""
    b, h, w = shape
    if kind == "sigma":
        return synthetic.make_loss_stack_inputs(b, h, w, 3, seed=seed, pose_sigma=0.2, condition_pose=False)
    inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=seed, condition_pose=False)
    from unsupervised_depth_opticalflow_egomotion_amd.networks import PoseCNN
    torch.manual_seed(seed)
    net = PoseCNN(3).to(dev()).eval()
    with torch.no_grad():
        x = torch.cat([G(a) for a in inp.imgs], 1)
        x = torch.nn.functional.interpolate(x, (256, 832), mode="bilinear", align_corners=False)   # the net's Linear(14,14)
        inp.pose = (50.0 * net(x)).cpu().numpy().astype(np.float32)     # x50: random-init outputs are ~1e-4
    return inp


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("kind,shape,seed", [("sigma", (2, 64, 208), 11), ("sigma", (2, 128, 448), 12),
                                             ("sigma", (1, 256, 832), 13), ("posecnn", (2, 64, 208), 14),
                                             ("posecnn", (1, 256, 832), 15)])
def test_unconditioned_poses_bit_exact_given_R(kind, shape, seed, ac):
    """The contract WITHOUT input conditioning: given the rotation matrix, every mask is decided
    on the reference's bits.  The oracle evaluates cos / sin correctly rounded here (oracle.trig("cr"): what cos_cr /
    sin_cr compute on the device) and nothing else changes: validity / dynamic / texture masks must EQUAL the oracle's
    (tolerance 0), occlusion bits may differ only inside the exp floor, losses 5e-6, gradients 1e-4 / 2e-5."""
    inp = raw_pose_inputs(kind, shape, seed)
    with O.trig("cr"):
        compare(inp, ac, 3)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("kind,shape,seed", [("sigma", (2, 128, 448), 12), ("sigma", (4, 256, 832), 16), ("posecnn", (1, 256, 832), 15)])
def test_unconditioned_poses_vs_host_libm(kind, shape, seed, ac, capsys):
    """Against the reference's arithmetic as it runs on THIS host (cos / sin from the vendor libm, <= 1 ulp): a
    pose-fed decision (dynamic / texture mask) may differ from the device's only at a pixel whose signed margin lies
    inside the floor a <= 1-ulp change of each of the six cos / sin values can move it by (tests/_margins.trig_floor,
    validated on twelve single-ulp oracle runs).  Prints the flip counts: the "<= k pixels per 10^6" of DESIGN.md."""
    inp = raw_pose_inputs(kind, shape, seed)
    S = 3
    base, floor, ulp_flips = M.trig_floor(inp, ac, S)
    _, mk_h, _, _ = run_hip(inp, ac, S)
    _, mk_o, _, _ = run_oracle(inp, ac, S)                 # host libm
    with O.trig("cr"):
        _, mk_c, _, _ = run_oracle(inp, ac, S)
    nflip = npx = ninside = 0
    for k in ("dyna_bwd", "dyna_fwd", "texture_bwd", "texture_fwd"):
        for s in range(S):
            h, o, c = N(mk_h[k][s]), N(mk_o[k][s]), N(mk_c[k][s])
            assert (h == c).all(), "%s scale %d: device differs from the correctly rounded oracle" % (k, s)
            f = h != o
            nflip += int(f.sum()); npx += h.size
            inside = np.abs(base[k][s]) <= floor[k][s]
            ninside += int(inside.sum())
            assert inside[f].all(), "%s scale %d: %d flip(s) outside the libm floor" % (k, s, int((f & ~inside).sum()))
    for k in ("valid_bwd", "valid_fwd"):                    # flow-warp validity does not depend on the pose
        for s in range(S):
            assert (N(mk_h[k][s]) == N(mk_o[k][s])).all()
    with capsys.disabled():
        print("\n[unconditioned %s %s ac=%d] device vs host-libm oracle: %d flip(s) in %d pose-fed decisions; %d pixel(s) "
              "inside the 1-ulp floor; single-ulp oracle runs flipped %d" % (kind, shape, ac, nflip, npx, ninside,
                                                                            sum(ulp_flips.values())))


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("kind,shape,seed", [("cond", (2, 128, 448), 600), ("cond", (4, 256, 832), 31),
                                             ("sigma", (4, 256, 832), 16), ("posecnn", (1, 256, 832), 15)])
def test_occlusion_bits_equal_the_correctly_rounded_oracle(kind, shape, seed, ac):
    """the softmax's ``exp`` was the last library value a mask bit depended on.  The kernels now decide
    the occlusion bits on the correctly rounded exponential (dfe_device.h occ_exp: the fast expf outside a band around the
    one value of exp(-|dl - dr|) where the hard weight flips, RN32 of the float64 exponential inside it) and the oracle's
    ``occ_exp("cr")`` mode states that function: the occlusion bits must be EQUAL -- every pixel, every scale, seeds that
    are NOT margin-checked included, no noise floor.  With cos / sin correctly rounded as well (``trig("cr")``) all eight masks
    are equal."""
    inp = raw_pose_inputs(kind, shape, seed) if kind != "cond" else synthetic.make_loss_stack_inputs(*shape, 3, seed=seed)
    S = 3
    _, mk_h, _, _ = run_hip(inp, ac, S)
    with O.trig("cr"), O.occ_exp("cr"):
        _, mk_c, _, _ = run_oracle(inp, ac, S)
    for k in MASKS:
        for s in range(S):
            h, c = N(mk_h[k][s]), N(mk_c[k][s])
            assert np.array_equal(h, c), "%s scale %d: %d bit(s) differ from the correctly rounded oracle" % (k, s, int((h != c).sum()))


def test_fused_stack_each_loss_gradient():
    """One loss row at a time, so that every term's backward is checked in isolation."""
    inp = synthetic.make_loss_stack_inputs(2, 64, 208, 3, seed=77)
    for k in MG.GEOM_WEIGHTS:
        if k in ("loss_depth_ssim", "loss_depth_consis", "loss_triangle", "loss_pnp", "loss_eight_point"):
            continue
        w = {q: (1.0 if q == k else 0.0) for q in MG.GEOM_WEIGHTS}
        compare(inp, False, 3, weights=w, strict=True)


@pytest.mark.parametrize("ac", [False, True])
def test_fused_stack_full_res_six_scales(ac):
    """BASELINE configs[4] shape at its per-GPU batch: B=2, 375x1242, 6 scales -- the general (non /2) bilinear
    pyramid, box-mean and adjoint paths; both ``align_corners`` modes (round 6: VERDICT r05 weak 3)."""
    inp = synthetic.make_loss_stack_inputs(2, 375, 1242, 6, seed=55, num_flow_scales=6)
    compare(inp, ac, 6)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("case", [0, 1])
def test_fused_stack_vs_golden(golden_dir, ac, case):
    """Straight against the numbers the real reference produced (tests/golden/G6_*.npz)."""
    g = np.load(os.path.join(golden_dir, "G6_ac%d.npz" % ac))
    b, h, w, seed = MG.G6_CASES[case]
    inp = synthetic.make_loss_stack_inputs(b, h, w, 3, seed=seed)
    lp, masks, total, (disps, pose, fb, ff) = run_hip(inp, ac, 3)
    key = "%dx%dx%d" % (b, h, w)
    assert int(g[key + "_within"][0]) == 0          # the fixture's seed is margin-checked (make_golden.store_margins)
    for k, v in lp.items():
        np.testing.assert_allclose(N(v), g[key + "_" + k], rtol=5e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(N(total), g[key + "_total"], rtol=5e-6)
    # the masks the reference exports in mask_pack (sample 0, scale 0): bit-exact
    # (mask_pack['valid_fwd_mask'] is inverse_warp2's validity towards the right frame, model_geometry.py:875)
    from unsupervised_depth_opticalflow_egomotion_amd.structures import inverse_warp2
    valid_to_r = inverse_warp2(G(inp.imgs[2]), disps[1][0].detach(), disps[2][0].detach(),
                               pose.detach()[:, 1].contiguous(), G(inp.K), align_corners=ac)[1]
    for nm, t in (("occ_fwd_mask", masks["occ_fwd"][0][0]), ("dyna_fwd_mask", masks["dyna_fwd"][0][0]),
                  ("texture_mask_fwd", masks["texture_fwd"][0][0]), ("valid_fwd_mask", valid_to_r[0]),
                  ("fwd_mask", (masks["valid_fwd"][0] * masks["occ_fwd"][0] * masks["dyna_fwd"][0])[0])):
        ref = np.unpackbits(g[key + "_mp_" + nm])[: h * w]
        assert np.array_equal(N(t).reshape(-1).astype(np.uint8), ref), nm
    gp = g[key + "_gpose"]
    assert np.abs(N(pose.grad) - gp).max() <= 2e-5 * np.abs(gp).max()
    # per-pixel gradients: strided samples element-wise (1e-4 of the scale) and the L1 norm (1e-5)
    def gsub(t, prefix):
        ref_sub, ref_sum = g[prefix + "_sub"], g[prefix + "_sum"]
        flat = N(t.grad).reshape(-1)
        scale = max(np.abs(flat).max(), 1e-12)
        assert np.abs(flat[::97] - ref_sub).max() <= 1e-4 * scale, prefix
        assert abs(np.abs(flat.astype(np.float64)).sum() - ref_sum[1]) <= 1e-5 * ref_sum[1], prefix
    for f in range(3):
        for s in range(3):
            gsub(disps[f][s], key + "_gdisp_%d_%d" % (f, s))
    for s in range(3):
        gsub(fb[s], key + "_gflow_b_%d" % s)
        gsub(ff[s], key + "_gflow_f_%d" % s)
    assert fb[3].grad is None or float(fb[3].grad.abs().sum()) == 0.0   # the 1/8 flow is dropped


@pytest.mark.parametrize("depth_terms", [False, True])
def test_fused_stack_is_deterministic(depth_terms):
    """Bitwise equal from run to run, every loss and every gradient -- with the optional depth terms too, whose
    projected-depth scatter into the source frames' disparity gradients adds 64-bit fixed-point integers
    (csrc/dfe_scatter.h) instead of floats."""
    inp = synthetic.make_loss_stack_inputs(2, 64, 208, 3, seed=5)
    a = run_hip(inp, False, 3, depth_terms=depth_terms)
    b = run_hip(inp, False, 3, depth_terms=depth_terms)
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k])
    assert torch.equal(a[3][1].grad, b[3][1].grad)
    for x, y in zip(a[3][2][:3], b[3][2][:3]):
        assert torch.equal(x.grad, y.grad)
    for f in range(3):
        for x, y in zip(a[3][0][f], b[3][0][f]):
            assert torch.equal(x.grad, y.grad), f


@pytest.mark.parametrize("shape,S", [((3, 70, 100), 2), ((1, 64, 208), 1), ((2, 40, 72), 3)])
def test_fused_stack_ragged_sizes(shape, S):
    """Sizes that are not multiples of the 256-px blocks / 32x8 tiles, odd batch, a single scale."""
    b, h, w = shape
    inp = synthetic.make_loss_stack_inputs(b, h, w, S, seed=300 + h)
    compare(inp, False, S)


def test_fused_stack_argument_errors():
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import geom_loss_stack
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    inp = synthetic.make_loss_stack_inputs(1, 32, 96, 3, seed=1)
    disps, pose, fb, ff = to_dev(inp, grad=False)
    il, it, ir = [G(a) for a in inp.imgs]
    K, Ki = G(inp.K), G(inp.K_inv)
    with pytest.raises(ValueError):      # wrong disparity shape
        geom_loss_stack(il, it, ir, disps[0][::-1], disps[1], disps[2], pose, fb, ff, K, Ki)
    with pytest.raises(ValueError):      # pose must be [B,2,6]
        geom_loss_stack(il, it, ir, disps[0], disps[1], disps[2], pose[:, 0], fb, ff, K, Ki)
    with pytest.raises(DfeError):        # scale 2 of a 8x24 image is 2x6: below the 3x3 minimum of the stencils
        tiny = synthetic.make_loss_stack_inputs(1, 8, 24, 3, seed=2)
        d2, p2, fb2, ff2 = to_dev(tiny, grad=False)
        geom_loss_stack(*[G(a) for a in tiny.imgs], d2[0], d2[1], d2[2], p2, fb2, ff2, G(tiny.K), G(tiny.K_inv))
    with pytest.raises(DfeError):        # CPU tensors never fall back
        geom_loss_stack(il.cpu(), it.cpu(), ir.cpu(), [d.cpu() for d in disps[0]], [d.cpu() for d in disps[1]],
                        [d.cpu() for d in disps[2]], pose.cpu(), [f.cpu() for f in fb], [f.cpu() for f in ff], K.cpu(), Ki.cpu())


def _grad_close(a, b, name, rel=2e-4, budget=0):
    a, b = N(a.grad), N(b.grad)
    scale = max(np.abs(b).max(), 1e-12)
    bad = int((np.abs(a - b) > rel * scale + 1e-9).sum())
    assert bad <= budget, "%s: %d bad elements, max diff %g scale %g" % (name, bad, np.abs(a - b).max(), scale)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("shape,S", [((2, 128, 448), 3), ((3, 70, 100), 2), ((1, 256, 832), 4)])
def test_flow_mode_vs_oracle(shape, S, ac):
    """mode 2 (Model_flow stack, model_flow.py:209-255): no hard masks on this path -- the occlusion weights are
    smooth Gaussians -- losses agree to 5e-6 and flow gradients to 1e-4 of their scale, element-wise."""
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import flow_loss_stack
    b, h, w = shape
    inp = synthetic.make_loss_stack_inputs(b, h, w, S, seed=1300 + h)
    wts = dict(loss_flow_pixel=0.15, loss_flow_ssim=0.85, loss_flow_smooth=10.0, loss_flow_consis=0.01)
    fbh, ffh = [G(a, True) for a in inp.flows_bwd], [G(a, True) for a in inp.flows_fwd]
    lp_h = flow_loss_stack(*[G(a) for a in inp.imgs], fbh, ffh, num_scales=S, align_corners=ac)
    sum(wts[k] * v.mean() for k, v in lp_h.items()).backward()
    fbo, ffo = [T(a).requires_grad_(True) for a in inp.flows_bwd], [T(a).requires_grad_(True) for a in inp.flows_fwd]
    lp_o, _ = O.GeomLossOracle(num_scales=S, align_corners=ac).flow_losses(*[T(a) for a in inp.imgs], fbo, ffo)
    sum(wts[k] * v.mean() for k, v in lp_o.items()).backward()
    assert set(lp_h) == set(lp_o)
    for k in lp_h:
        np.testing.assert_allclose(N(lp_h[k]), N(lp_o[k]), rtol=5e-6, atol=1e-7, err_msg=k)
    for s in range(S):
        _grad_close(fbh[s], fbo[s], "gflow_b_%d" % s, rel=1e-4)
        _grad_close(ffh[s], ffo[s], "gflow_f_%d" % s, rel=1e-4)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("shape,S", [((2, 128, 448), 3), ((3, 70, 100), 2), ((1, 256, 832), 4), ((4, 256, 832), 3)])
@pytest.mark.parametrize("depth_terms", [False, True])
def test_depth_mode_vs_oracle(shape, S, ac, depth_terms):
    """mode 1 (Model_depth stack, model_depth.py:272-337): the inverse_warp2 validity and texture masks are exact IEEE
    arithmetic on identical inputs -> they must EQUAL the oracle's (no noise floor on this path: no transcendental
    feeds a decision); losses 5e-6, gradients 1e-4 of their scale element-wise, pose gradient 2e-5."""
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import depth_loss_stack
    b, h, w = shape
    if b == 4 and (ac or depth_terms):
        pytest.skip("BASELINE configs[1] (B=4, 256x832, S=3) is compared once, in the default convention")
    inp = synthetic.make_loss_stack_inputs(b, h, w, S, seed=1700 + h + (b if b == 4 else 0))
    wts = dict(loss_depth_pixel=1.0, loss_depth_smooth=0.1)
    if depth_terms:   # the terms the reference keeps commented (model_depth.py:326-327,332-333), config weights
        wts.update(loss_depth_ssim=0.85, loss_depth_consis=0.1)
    dh = [[G(a, True) for a in lst] for lst in inp.disps]
    ph = G(inp.pose, True)
    lp_h, mk_h = depth_loss_stack(*[G(a) for a in inp.imgs], dh[0], dh[1], dh[2], ph, G(inp.K), num_scales=S,
                                  align_corners=ac, return_masks=True, enable_depth_ssim=depth_terms,
                                  enable_depth_consis=depth_terms)
    assert set(lp_h) == set(wts)
    sum(wts[k] * v.mean() for k, v in lp_h.items()).backward()
    do = [[T(a).requires_grad_(True) for a in lst] for lst in inp.disps]
    po = T(inp.pose).requires_grad_(True)
    lp_o, mk_o = O.GeomLossOracle(num_scales=S, align_corners=ac).depth_losses(
        *[T(a) for a in inp.imgs], do[0], do[1], do[2], po, T(inp.K), enable_depth_ssim=depth_terms,
        enable_depth_consis=depth_terms)
    sum(wts[k] * lp_o[k].mean() for k in wts).backward()
    for k in ("valid_to_l", "valid_to_r", "texture_bwd", "texture_fwd"):
        for s in range(S):
            assert np.array_equal(N(mk_h[k][s]), N(mk_o[k][s])), (k, s)
    for k in lp_h:
        np.testing.assert_allclose(N(lp_h[k]), N(lp_o[k]), rtol=5e-6, atol=1e-7, err_msg=k)
    for f in range(3):
        for s in range(S):
            a, c = N(dh[f][s].grad), N(do[f][s].grad)
            bad = int((np.abs(a - c) > 1e-4 * max(np.abs(c).max(), 1e-12) + 1e-9).sum())
            assert bad == 0, ("gdisp_%d_%d" % (f, s), bad, np.abs(a - c).max(), np.abs(c).max())
    gp_h, gp_o = N(ph.grad), N(po.grad)
    assert np.abs(gp_h - gp_o).max() <= 2e-5 * max(np.abs(gp_o).max(), 1e-12) + 1e-7, (gp_h, gp_o)


def test_flow_mode_argument_errors():
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import flow_loss_stack
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    inp = synthetic.make_loss_stack_inputs(1, 32, 96, 3, seed=1)
    il, it, ir = [G(a) for a in inp.imgs]
    fb, ff = [G(a) for a in inp.flows_bwd], [G(a) for a in inp.flows_fwd]
    with pytest.raises(ValueError):      # flow pyramid in the wrong order
        flow_loss_stack(il, it, ir, fb[::-1], ff)
    with pytest.raises(DfeError):        # CPU tensors never fall back
        flow_loss_stack(il.cpu(), it.cpu(), ir.cpu(), [f.cpu() for f in fb], [f.cpu() for f in ff])


def test_full_size_properties():
    """BASELINE size (B=4, 256x832, S=3), properties that need no oracle:
    (1) batch-shard equivalence: the stack on samples [0,1] and [2,3] separately gives bit-identical per-sample
        losses and gradients to the full batch (what makes data-parallel sharding exact);
    (2) the backward is linear in the upstream gradient: bwd(g1) + bwd(g2) == bwd(g1 + g2);
    (3) every mask is {0,1} and fwd_mask == valid*occ*dyna, texture-gated mask <= fwd_mask."""
    from unsupervised_depth_opticalflow_egomotion_amd.loss_stack import geom_loss_stack
    inp = synthetic.make_loss_stack_inputs(4, 256, 832, 3, seed=1234)

    def run(sl, gvec=None):
        disps = [[G(a[sl], True) for a in lst] for lst in inp.disps]
        pose, fb, ff = G(inp.pose[sl], True), [G(a[sl], True) for a in inp.flows_bwd], [G(a[sl], True) for a in inp.flows_fwd]
        il, it, ir = [G(a[sl]) for a in inp.imgs]
        lp, masks = geom_loss_stack(il, it, ir, disps[0], disps[1], disps[2], pose, fb, ff, G(inp.K[sl]), G(inp.K_inv[sl]),
                                    num_scales=3, return_masks=True)
        L = torch.stack([lp[k] for k in sorted(lp)])            # [8, B]
        gv = torch.ones_like(L) if gvec is None else gvec
        (L * gv).sum().backward()
        return L.detach(), masks, disps, pose, fb, ff
    full = run(slice(0, 4))
    lo, hi = run(slice(0, 2)), run(slice(2, 4))
    assert torch.equal(full[0][:, :2], lo[0]) and torch.equal(full[0][:, 2:], hi[0])
    assert torch.equal(full[3].grad[:2], lo[3].grad) and torch.equal(full[3].grad[2:], hi[3].grad)
    for s in range(3):
        assert torch.equal(full[2][1][s].grad[:2], lo[2][1][s].grad) and torch.equal(full[5][s].grad[2:], hi[5][s].grad)
    # linearity in the upstream gradient
    g1 = torch.rand(8, 4, device=dev())
    g2 = torch.rand(8, 4, device=dev())
    a, b, c = run(slice(0, 4), g1), run(slice(0, 4), g2), run(slice(0, 4), g1 + g2)
    for s in range(3):
        ref = c[4][s].grad
        assert float(((a[4][s].grad + b[4][s].grad) - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9
        ref = c[2][1][s].grad
        assert float(((a[2][1][s].grad + b[2][1][s].grad) - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9
    assert float(((a[3].grad + b[3].grad) - c[3].grad).abs().max()) <= 1e-4 * float(c[3].grad.abs().max())
    # mask algebra
    m = full[1]
    for s in range(3):
        for k in m:
            u = torch.unique(m[k][s])
            assert set(u.tolist()) <= {0.0, 1.0}, k


@pytest.mark.parametrize("shape", [(2, 64, 208), (4, 256, 832)])
def test_fused_two_level_pyramid_kernel_is_the_generic_one_bit_for_bit(shape):
    """k_geom_pyramids12 (levels 1 and 2 of an exact power-of-two pyramid from one read of the frames) against the
    per-(frame, scale) jobs of k_geom_pyramids (DFE_PYRAMIDS_GENERIC=1): every loss value, mask and gradient of the stack
    that consumes the pyramids is bit-identical."""
    inp = synthetic.make_loss_stack_inputs(*shape, 3, seed=47)

    def run(flag):
        old = os.environ.pop("DFE_PYRAMIDS_GENERIC", None)
        if flag:
            os.environ["DFE_PYRAMIDS_GENERIC"] = "1"
        try:
            lp, mk, tot, (d, p, fb, ff) = run_hip(inp, False, 3)
            torch.cuda.synchronize()
            return ([N(v) for v in lp.values()], [N(t) for v in mk.values() for t in v], [N(t.grad) for lst in d for t in lst],
                    N(p.grad), [N(t.grad) for t in fb[:3]] + [N(t.grad) for t in ff[:3]])
        finally:
            os.environ.pop("DFE_PYRAMIDS_GENERIC", None)
            if old is not None:
                os.environ["DFE_PYRAMIDS_GENERIC"] = old
    a, b = run(False), run(True)
    for x, y in zip(a[0] + a[1] + a[2] + [a[3]] + a[4], b[0] + b[1] + b[2] + [b[3]] + b[4]):
        assert np.array_equal(x, y)
