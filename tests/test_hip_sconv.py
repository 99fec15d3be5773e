"""GPU: the strided weight gradient of csrc/ops_sconv.hip (k x k / stride 2 / padding k // 2 layers, fp32 MFMA straight from NCHW)
through the C ABI against float64 ``aten::convolution_backward``: on the networks' layer shapes (depth_model.py:60-95,
feature_pyramid.py:7-36, pose_cnn.py:14-36) and on ragged ones (odd sizes, channel counts that are no multiple of 4 / 16 / 64,
one-pixel outputs, valid padding, stride 1).  Tolerance 2e-5 of the result's scale (plain fp32 sums in another order than aten's;
measured <= 1e-6), bit-reproducible (fixed-order split sums, no atomics)."""
import pytest
import torch
import torch.nn.functional as F

from unsupervised_depth_opticalflow_egomotion_amd import _lib, convs, ops

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, W, k, stride, P)
SHAPES = [(2, 3, 64, 64, 208, 7, 2, 3), (2, 3, 16, 64, 208, 3, 2, 1), (2, 16, 32, 64, 208, 3, 2, 1), (1, 9, 16, 64, 208, 7, 2, 3),
          (2, 16, 32, 32, 104, 5, 2, 2),                                                                  # the routed layers (thin inputs)
          (2, 64, 128, 32, 104, 3, 2, 1), (2, 256, 512, 16, 52, 3, 2, 1), (3, 128, 196, 8, 26, 3, 2, 1), (4, 256, 256, 4, 13, 3, 2, 1),
          (1, 3, 5, 6, 6, 3, 2, 1), (2, 33, 17, 7, 9, 3, 2, 1), (3, 40, 70, 10, 13, 3, 2, 1), (1, 17, 20, 5, 130, 3, 2, 1), (2, 70, 33, 13, 7, 3, 2, 1),
          (1, 5, 16, 9, 11, 7, 2, 3), (2, 9, 10, 17, 23, 7, 2, 3), (1, 16, 32, 21, 27, 5, 2, 2), (1, 7, 16, 3, 3, 5, 2, 2), (1, 4, 4, 1, 1, 3, 2, 1),
          (1, 20, 24, 2, 70, 3, 2, 1), (2, 16, 16, 4, 4, 3, 2, 0), (1, 20, 40, 9, 9, 3, 1, 1), (12, 3, 64, 256, 832, 7, 2, 3)]


def dev():
    return torch.device("cuda:0")


def make(shape):
    B, Ci, Co, H, W, k, s, P = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, k, k, device=dev())
    gy = torch.randn(B, Co, (H + 2 * P - k) // s + 1, (W + 2 * P - k) // s + 1, device=dev())
    return x, w, gy


def aten_wgrad(x, w, gy, s, P):
    return torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [P, P], [1, 1], False, [0, 0], 1, [False, True, False])[1]


@pytest.mark.parametrize("shape", SHAPES)
def test_sconv_wgrad_matches_float64(shape):
    B, Ci, Co, H, W, k, s, P = shape
    x, w, gy = make(shape)
    assert ops.sconv_wgrad_supported(x.shape, Co, k, s, P)
    gw = ops.sconv_wgrad(x, gy, k, s, P)
    ref = aten_wgrad(x.double(), w.double(), gy.double(), s, P)
    err, scale = float((gw.double() - ref).abs().max()), float(ref.abs().max())
    assert gw.shape == ref.shape and err <= 2e-5 * scale + 1e-12, (err, scale)
    assert torch.equal(gw, ops.sconv_wgrad(x, gy, k, s, P))


def test_sconv_wgrad_grid_targets_change_the_association_only():
    """dfe_sconv_tune: another number of pixel splits = another (fixed) order of the same sums."""
    shape = (4, 9, 16, 64, 208, 7, 2, 3)
    x, w, gy = make(shape)
    lib = _lib.get_lib()
    ref = aten_wgrad(x.double(), w.double(), gy.double(), 2, 3)
    try:
        outs = []
        for blocks, rows in ((0, 0), (96, 0), (2048, 1)):
            assert lib.dfe_sconv_tune(blocks, rows) == 0
            g = ops.sconv_wgrad(x, gy, 7, 2, 3)
            assert float((g.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
            assert torch.equal(g, ops.sconv_wgrad(x, gy, 7, 2, 3))
            outs.append(g)
        assert not torch.equal(outs[0], outs[1]) or not torch.equal(outs[0], outs[2])
    finally:
        lib.dfe_sconv_tune(0, 0)


def test_sconv_wgrad_reads_batch_strided_views_and_rejects_bad_arguments():
    shape = (3, 16, 32, 18, 30, 3, 2, 1)
    x, w, gy = make(shape)
    wide_x = torch.randn(3, 40, 18, 30, device=dev())
    wide_x[:, 8:24] = x
    wide_g = torch.randn(3, 50, 9, 15, device=dev())
    wide_g[:, 10:42] = gy
    assert torch.equal(ops.sconv_wgrad(wide_x[:, 8:24], wide_g[:, 10:42], 3, 2, 1), ops.sconv_wgrad(x, gy, 3, 2, 1))
    lib = _lib.get_lib()
    assert lib.dfe_sconv_wgrad_floats(1, 8, 8, 16, 16, 3, 3, 1) == 0          # stride 3
    assert lib.dfe_sconv_wgrad_floats(1, 8, 8, 16, 16, 9, 2, 4) == 0          # 9x9
    assert lib.dfe_sconv_wgrad_floats(1, 64, 64, 16, 16, 5, 2, 2) == 0        # 5x5 with 64 input channels: outside the family
    assert lib.dfe_sconv_wgrad_floats(1, 8, 8, 2, 2, 7, 2, 0) == 0            # no output pixel
    with pytest.raises(_lib.DfeError):
        ops.sconv_wgrad(x, gy[:, :, :-1], 3, 2, 1)
    with pytest.raises(_lib.DfeError):
        ops.sconv_wgrad(torch.randn(1, 64, 16, 16, device=dev()), torch.randn(1, 64, 8, 8, device=dev()), 5, 2, 2)
    gw, ws = torch.empty(32, 16, 3, 3, device=dev()), torch.empty(1 << 20, device=dev())
    P = _lib.ptr
    assert lib.dfe_sconv_wgrad(None, 0, P(gy), gy.stride(0), P(gw), P(ws), 3, 16, 32, 18, 30, 3, 2, 1, None) == -1
    assert lib.dfe_sconv_wgrad(P(x), 1, P(gy), gy.stride(0), P(gw), P(ws), 3, 16, 32, 18, 30, 3, 2, 1, None) == -2
    assert lib.dfe_sconv_wgrad(P(x), x.stride(0), P(gy), gy.stride(0), P(gw), P(ws), 3, 16, 32, 18, 30, 8, 2, 4, None) == -4


def test_sconv_wgrad_nonfinite_inputs_stay_in_their_channel():
    """A NaN in one input channel reaches that channel's filter taps only (the channels share staging slots and MFMA steps)."""
    shape = (2, 9, 16, 20, 36, 7, 2, 3)
    x, w, gy = make(shape)
    x[1, 4, 7, 9] = float("nan")
    gw = ops.sconv_wgrad(x, gy, 7, 2, 3)
    assert torch.isnan(gw[:, 4]).any() and torch.isfinite(gw[:, :4]).all() and torch.isfinite(gw[:, 5:]).all()
    # tap (ky, kx) meets input pixel (7, 9) only when 7 + 3 - ky and 9 + 3 - kx are even
    nan_taps = torch.isnan(gw[0, 4])
    want = torch.tensor([[(10 - ky) % 2 == 0 and (12 - kx) % 2 == 0 for kx in range(7)] for ky in range(7)], device=dev())
    assert torch.equal(nan_taps, want)


@pytest.mark.parametrize("k,ci,co,bias,routed", [(7, 3, 64, False, True), (5, 16, 32, True, True), (7, 9, 16, True, True), (3, 16, 32, True, True),
                                                 (3, 32, 64, True, False)])
def test_strided_conv2d_module_routes_thin_layers_weight_gradients(k, ci, co, bias, routed, monkeypatch):
    """convs.Conv2d(stride 2): the weight gradient of a layer with <= 16 input channels goes to dfe_sconv_wgrad, everything else
    stays on MIOpen; values and gradients match the float64 reference graph; DFE_SCONV=0 (module constant) restores MIOpen."""
    torch.manual_seed(k + ci)
    conv = convs.Conv2d(ci, co, k, 2, k // 2, bias=bias).to(dev())
    x = torch.randn(2, ci, 32, 52, device=dev(), requires_grad=True)
    calls = []
    orig = ops.sconv_wgrad
    monkeypatch.setattr(ops, "sconv_wgrad", lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1])
    y = F.leaky_relu(conv(x), 0.1)
    g = torch.randn_like(y)
    y.backward(g)
    assert len(calls) == (1 if routed else 0)
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True) if bias else None
    yd = F.leaky_relu(F.conv2d(xd, wd, bd, 2, k // 2), 0.1)
    yd.backward(g.double())
    for a, ref in ((y.detach(), yd.detach()), (x.grad, xd.grad), (conv.weight.grad, wd.grad)) + (((conv.bias.grad, bd.grad),) if bias else ()):
        assert float((a.double() - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    monkeypatch.setattr(convs, "SCONV", False)
    calls.clear()
    conv.zero_grad()
    F.leaky_relu(conv(x.detach()), 0.1).backward(g)
    assert not calls
