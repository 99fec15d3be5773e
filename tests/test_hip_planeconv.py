"""GPU: the small-plane 3x3 convolutions (csrc/ops_planeconv.hip: forward + bias + LeakyReLU into two concatenated buffers,
data gradient, weight gradient; reference pwc_tf.py:28-47 / 113-135) through the C ABI against float64
``aten::convolution`` / ``convolution_backward`` of the same tensors.  Tolerance: 1e-5 of the result's scale (fp32 fma
chains in a different order than MIOpen's; the float64 reference is the arbiter), results reproducible bit for bit."""
import pytest
import torch
import torch.nn.functional as F

from unsupervised_depth_opticalflow_egomotion_amd import ops
from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, W): every layer of PWC levels 6 and 5 at 256x832 (B = 8 pairs), ragged / tiny / one-row / wide cases
SHAPES = [(8, 81, 128, 4, 13), (8, 128, 128, 4, 13), (8, 256, 96, 4, 13), (8, 224, 64, 4, 13), (8, 160, 32, 4, 13),
          (8, 211, 128, 8, 26), (8, 128, 128, 8, 26), (8, 256, 96, 8, 26), (8, 224, 64, 8, 26), (8, 160, 32, 8, 26),
          (1, 1, 1, 1, 1), (2, 3, 5, 3, 5), (3, 17, 33, 1, 70), (2, 20, 70, 9, 7), (1, 36, 48, 16, 52), (2, 5, 16, 64, 2),
          (8, 128, 128, 16, 52)]      # the last one takes the 64-channel block tile (NSUB = 4) in both directions


def dev():
    return torch.device("cuda:0")


def close(a, ref, what):
    scale = float(ref.abs().max())
    err = float((a.double() - ref).abs().max())
    assert err <= 1e-5 * scale + 1e-12, (what, err, scale)


@pytest.mark.parametrize("shape", SHAPES)
def test_planeconv_matches_float64(shape):
    B, Ci, Co, H, W = shape
    assert get_lib().dfe_planeconv_supported(B, Ci, Co, H, W) == 1
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, 3, 3, device=dev()) / (3.0 * Ci ** 0.5)
    bias = torch.randn(Co, device=dev())
    gy = torch.randn(B, Co, H, W, device=dev())
    # forward + bias + LeakyReLU(0.1) into channel slices of two wider buffers
    d1 = torch.full((B, Co + 3, H, W), 7.0, device=dev())
    d2 = torch.full((B, Co + 5, H, W), 9.0, device=dev())
    ops.planeconv_fwd_into(x, w, bias, 0.1, d1, 3, d2, 0)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), 1, 1), 0.1)
    close(d1[:, 3:], ref, "fwd d1")
    assert torch.equal(d1[:, 3:], d2[:, :Co]) and bool((d1[:, :3] == 7.0).all()) and bool((d2[:, Co:] == 9.0).all())
    y = ops.planeconv_forward(x, w)                       # no bias, no activation
    close(y, F.conv2d(x.double(), w.double(), None, 1, 1), "fwd raw")
    assert torch.equal(y, ops.planeconv_forward(x, w))    # fixed-order sums
    gx, gw = ops.planeconv_backward(gy, x, w)
    rgx, rgw, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False,
                                                      [0, 0], 1, [True, True, False])
    close(gx, rgx, "dgrad")
    close(gw, rgw, "wgrad")
    gx2, gw2 = ops.planeconv_backward(gy, x, w)
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2)


def test_planeconv_rejects_large_planes_and_null():
    lib = get_lib()
    assert lib.dfe_planeconv_supported(8, 64, 64, 64, 208) == 0 and lib.dfe_planeconv_ws_floats(8, 64, 64, 64, 208) == 0
    x = torch.zeros(1, 4, 64, 208, device=dev())
    w = torch.zeros(4, 4, 3, 3, device=dev())
    assert not ops.planeconv_eligible(x, w)
    assert lib.dfe_planeconv_fwd(None, None, None, 1.0, None, 0, None, 0, None, 1, 1, 1, 1, 1, None) != 0


def test_planeconv_nonfinite_inputs_propagate():
    x = torch.randn(1, 8, 4, 13, device=dev())
    w = torch.randn(16, 8, 3, 3, device=dev())
    x[0, 3, 2, 5] = float("nan")
    y = ops.planeconv_forward(x, w)
    ref = F.conv2d(x, w, None, 1, 1)
    assert torch.equal(torch.isnan(y), torch.isnan(ref))


def test_planeconv_act_autograd_matches_aten():
    """ops.planeconv_act (PoseCNN's Conv2d(12, 12, 3, 1, 1) + ReLU on 2x7 planes) against the ATen composition: output and the
    three gradients."""
    torch.manual_seed(4)
    x = torch.randn(4, 12, 2, 7, device=dev())
    w = torch.randn(12, 12, 3, 3, device=dev()) * 0.2
    b = torch.randn(12, device=dev()) * 0.1
    gy = torch.randn(4, 12, 2, 7, device=dev())
    outs = []
    for fused in (True, False):
        xi, wi, bi = [t.clone().requires_grad_(True) for t in (x, w, b)]
        y = ops.planeconv_act(xi, wi, bi, 0.0) if fused else F.relu(F.conv2d(xi, wi, bi, 1, 1))
        (y * gy).sum().backward()
        outs.append([y.detach(), xi.grad, wi.grad, bi.grad])
    for u, v in zip(*outs):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-9


@pytest.mark.parametrize("shape,slope", [((4, 256, 12, 2, 7), 1.0), ((4, 24, 12, 2, 7), 0.0), ((4, 12, 12, 2, 7), 0.0), ((3, 5, 7, 4, 13), 0.1)])
def test_conv1x1_small_autograd_matches_aten(shape, slope):
    """ops.Conv1x1SmallFn (PoseCNN's 1x1 convolutions on 2x7 planes) against the ATen composition: output and the gradients."""
    import torch.nn as nn
    B, Ci, Co, H, W = shape
    torch.manual_seed(B + Ci)
    conv = nn.Conv2d(Ci, Co, 1).to(dev())
    x = torch.randn(B, Ci, H, W, device=dev())
    gy = torch.randn(B, Co, H, W, device=dev())
    assert ops.conv1x1_small_eligible(x, conv)
    outs = []
    for fused in (True, False):
        conv.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = ops.conv1x1_small(xi, conv, slope) if fused else F.leaky_relu(conv(xi), slope)
        (y * gy).sum().backward()
        outs.append([y.detach(), xi.grad, conv.weight.grad.clone(), conv.bias.grad.clone()])
    for u, v in zip(*outs):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-9
    big = torch.zeros(1, 4, 64, 208, device=dev())
    assert not ops.conv1x1_small_eligible(big, nn.Conv2d(4, 4, 1).to(dev()))
