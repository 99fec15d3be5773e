"""GPU: the strided convolutions of csrc/ops_sconv.hip (forward with the fused bias + activation, data gradient, weight gradient
of k x k / stride 2 / padding k // 2 layers, fp32 MFMA straight from NCHW) through the C ABI against float64 ``aten::convolution``
/ ``convolution_backward``: on the networks' layer shapes (depth_model.py:60-95, feature_pyramid.py:7-36, pose_cnn.py:14-36) and on
ragged ones (odd sizes, channel counts that are no multiple of 4 / 16 / 64, one-pixel outputs).  Tolerance 2e-5 of the result's
scale (plain fp32 sums in another order than aten's; measured <= 1e-6), bit-reproducible (fixed-order split sums, no atomics)."""
import pytest
import torch
import torch.nn.functional as F

from unsupervised_depth_opticalflow_egomotion_amd import _lib, convs, ops

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, W, k)
SHAPES = [(2, 3, 64, 64, 208, 7), (2, 64, 128, 32, 104, 3), (1, 128, 256, 32, 104, 3), (2, 256, 512, 16, 52, 3),      # ResNet18 encoder
          (2, 3, 16, 64, 208, 3), (2, 16, 32, 64, 208, 3), (1, 32, 64, 64, 208, 3), (2, 64, 96, 32, 104, 3), (2, 96, 128, 16, 52, 3),
          (3, 128, 196, 8, 26, 3),                                                                                     # FeaturePyramid
          (1, 9, 16, 64, 208, 7), (2, 16, 32, 32, 104, 5), (2, 256, 256, 8, 26, 3), (4, 256, 256, 4, 13, 3),              # PoseCNN
          (1, 3, 5, 6, 6, 3), (2, 33, 17, 7, 9, 3), (3, 40, 70, 10, 13, 3), (1, 17, 20, 5, 130, 3), (2, 70, 33, 13, 7, 3),
          (1, 5, 64, 9, 11, 7), (2, 9, 10, 17, 23, 7), (1, 16, 32, 21, 27, 5), (1, 7, 20, 3, 3, 5), (1, 4, 4, 1, 1, 3), (1, 20, 24, 2, 70, 3)]


def dev():
    return torch.device("cuda:0")


def make(shape):
    B, Ci, Co, H, W, k = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, k, k, device=dev()) / (k * Ci ** 0.5)
    P = k // 2
    gy = torch.randn(B, Co, (H + 2 * P - k) // 2 + 1, (W + 2 * P - k) // 2 + 1, device=dev())
    return x, w, gy, P


def close(a, ref, tol=2e-5):
    err, scale = float((a.double() - ref).abs().max()), float(ref.abs().max())
    assert a.shape == ref.shape and err <= tol * scale + 1e-12, (err, scale)


@pytest.mark.parametrize("shape", SHAPES)
def test_sconv_matches_float64(shape):
    B, Ci, Co, H, W, k = shape
    x, w, gy, P = make(shape)
    ref = F.conv2d(x.double(), w.double(), None, 2, P)
    rgx, rgw, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [2, 2], [P, P], [1, 1], False, [0, 0], 1,
                                                       [True, True, False])
    assert ops.sconv_fwd_supported(x.shape, Co, k)
    y = ops.sconv_fwd(x, w)
    close(y, ref)
    assert torch.equal(y, ops.sconv_fwd(x, w))
    if k in (3, 5):
        assert ops.sconv_dgrad_supported(x.shape, Co, k)
        gx = ops.sconv_dgrad(gy, w, x.shape)
        close(gx, rgx)
        assert torch.equal(gx, ops.sconv_dgrad(gy, w, x.shape))
    if ops.sconv_wgrad_supported(x.shape, Co, k, 2, P):
        gw = ops.sconv_wgrad(x, gy, k, 2, P)
        close(gw, rgw)
        assert torch.equal(gw, ops.sconv_wgrad(x, gy, k, 2, P))
    else:
        assert k != 3 or Ci < 16      # every 3x3 layer with >= 16 input channels, the stems and 5x5 x 16 are in the family


@pytest.mark.parametrize("shape,slope", [((2, 16, 32, 32, 104, 3), 0.1), ((1, 9, 16, 64, 208, 7), 0.0), ((2, 16, 32, 32, 104, 5), 0.0),
                                         ((4, 256, 256, 4, 13, 3), 0.0), ((2, 33, 17, 7, 9, 3), 0.1)])
def test_sconv_fused_epilogue_is_the_separate_pass_bit_for_bit(shape, slope):
    """bias + LeakyReLU / ReLU inside the kernel's epilogue (or inside the split sum of a thin layer) = the plain convolution
    followed by the same two fp32 operations."""
    x, w, _, P = make(shape)
    bias = torch.randn(shape[2], device=dev())
    y0 = ops.sconv_fwd(x, w)
    z = y0 + bias.view(1, -1, 1, 1)
    ref = torch.where(z > 0, z, z * slope)
    assert torch.equal(ops.sconv_fwd(x, w, bias, slope), ref)


def test_sconv_reads_and_writes_batch_strided_views_and_rejects_bad_arguments():
    shape = (3, 16, 32, 18, 30, 3)
    x, w, gy, P = make(shape)
    wide_x = torch.randn(3, 40, 18, 30, device=dev())
    wide_x[:, 8:24] = x
    wide_g = torch.randn(3, 50, 9, 15, device=dev())
    wide_g[:, 10:42] = gy
    xv, gv = wide_x[:, 8:24], wide_g[:, 10:42]
    assert torch.equal(ops.sconv_fwd(xv, w), ops.sconv_fwd(x, w))
    assert torch.equal(ops.sconv_dgrad(gv, w, x.shape), ops.sconv_dgrad(gy, w, x.shape))
    assert torch.equal(ops.sconv_wgrad(xv, gv, 3, 2, 1), ops.sconv_wgrad(x, gy, 3, 2, 1))
    out = torch.zeros(3, 50, 9, 15, device=dev())
    ops.sconv_fwd(x, w, out=out[:, 10:42])
    assert torch.equal(out[:, 10:42], ops.sconv_fwd(x, w)) and float(out[:, :10].abs().max()) == 0.0 and float(out[:, 42:].abs().max()) == 0.0
    lib = _lib.get_lib()
    assert lib.dfe_sconv_fwd_floats(1, 8, 8, 16, 16, 4) == 0 and lib.dfe_sconv_fwd_floats(1, 8, 8, 16, 16, 9) == 0
    assert lib.dfe_sconv_dgrad_floats(1, 8, 8, 16, 16, 7) == 0
    assert lib.dfe_sconv_wgrad_floats(1, 8, 8, 16, 16, 3, 3, 1) == 0          # stride 3
    assert lib.dfe_sconv_wgrad_floats(1, 64, 64, 16, 16, 5, 2, 2) == 0        # 5x5 with 64 input channels: outside the family
    with pytest.raises(_lib.DfeError):
        ops.sconv_fwd(x, torch.randn(32, 16, 4, 4, device=dev()))
    with pytest.raises(_lib.DfeError):
        ops.sconv_wgrad(x, gy[:, :, :-1], 3, 2, 1)
    with pytest.raises(_lib.DfeError):
        ops.sconv_dgrad(gy, w, (3, 16, 20, 30))
    ws = torch.empty(1 << 16, device=dev())
    null = None
    assert lib.dfe_sconv_fwd(null, 0, _lib.ptr(w), null, 1.0, _lib.ptr(out), 0, _lib.ptr(ws), 3, 16, 32, 18, 30, 3, null) == -1
    assert lib.dfe_sconv_fwd(_lib.ptr(x), 1, _lib.ptr(w), null, 1.0, _lib.ptr(out), out.stride(0), _lib.ptr(ws), 3, 16, 32, 18, 30, 3, null) == -2


def test_sconv_nonfinite_inputs_stay_where_they_belong():
    """A NaN in x reaches exactly the outputs whose window holds it; zero padding multiplies nothing (no 0 * inf)."""
    shape = (1, 16, 32, 12, 20, 3)
    x, w, _, P = make(shape)
    x[0, 3, 5, 7] = float("nan")
    y = ops.sconv_fwd(x, w)
    bad = torch.isnan(y).any(dim=1)[0]
    want = torch.zeros_like(bad)
    for oy in range(y.shape[2]):
        for ox in range(y.shape[3]):
            want[oy, ox] = (2 * oy - 1 <= 5 <= 2 * oy + 1) and (2 * ox - 1 <= 7 <= 2 * ox + 1)
    assert torch.equal(bad, want)


@pytest.mark.parametrize("k,ci,co,bias", [(3, 32, 64, True), (7, 3, 64, False), (5, 16, 32, True), (7, 9, 16, True)])
def test_strided_conv2d_module_takes_the_kernels_and_matches_aten(k, ci, co, bias, monkeypatch):
    """convs.Conv2d(stride 2) and ops.conv_bias_act route forward, data and weight gradients to csrc/ops_sconv.hip; values and
    gradients match the float64 reference graph, and DFE_SCONV=0 (module constant) restores MIOpen."""
    torch.manual_seed(k + ci)
    conv = convs.Conv2d(ci, co, k, 2, k // 2, bias=bias).to(dev())
    x = torch.randn(2, ci, 32, 52, device=dev(), requires_grad=True)
    calls = []
    for name in ("sconv_fwd", "sconv_dgrad", "sconv_wgrad"):
        orig = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda o, n: lambda *a, **kw: (calls.append(n), o(*a, **kw))[1])(orig, name))
    if bias:
        assert ops.conv_bias_act_eligible(x, conv)
        y = ops.conv_bias_act(x, conv, 0.1)
    else:
        y = F.leaky_relu(conv(x), 0.1)
    g = torch.randn_like(y)
    y.backward(g)
    want = {"sconv_fwd", "sconv_wgrad"} | ({"sconv_dgrad"} if k in (3, 5) else set())
    assert set(calls) == want, calls
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True) if bias else None
    yd = F.leaky_relu(F.conv2d(xd, wd, bd, 2, k // 2), 0.1)
    yd.backward(g.double())
    close(y.detach(), yd.detach())
    close(x.grad, xd.grad, 5e-5)
    close(conv.weight.grad, wd.grad, 5e-5)
    if bias:
        close(conv.bias.grad, bd.grad, 5e-5)
    monkeypatch.setattr(convs, "SCONV", False)
    calls.clear()
    conv(x.detach())
    assert not calls and not ops.conv_bias_act_eligible(x, conv)
