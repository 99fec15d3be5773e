"""GPU: the fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (csrc/ops_wino.hip) through the C ABI against
float64 ``aten::convolution`` / ``convolution_backward``: forward (padding 1 and valid) and the data-gradient form, on the
networks' layer shapes and on ragged ones (odd sizes, channel counts that are no multiple of 2 / 16 / 32).  Tolerance 2e-5 of
the result's scale (fp32 Winograd: the filter and data transforms add a few roundings to the direct sums), reproducible."""
import pytest
import torch
import torch.nn.functional as F

from unsupervised_depth_opticalflow_egomotion_amd import ops

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, W, P)
SHAPES = [(2, 64, 64, 64, 208, 1), (2, 128, 128, 32, 104, 1), (1, 115, 128, 64, 208, 1), (2, 96, 32, 34, 50, 0),
          (1, 256, 96, 16, 52, 1), (3, 17, 33, 7, 9, 1), (1, 1, 1, 3, 3, 1), (2, 5, 70, 11, 6, 0), (1, 18, 40, 2, 2, 1),
          (1, 34, 32, 5, 64, 1), (2, 40, 48, 18, 54, 0), (1, 33, 20, 6, 10, 2), (2, 32, 32, 7, 9, 2), (1, 8, 8, 4, 34, 0),
          (2, 24, 16, 12, 20, 1), (1, 16, 8, 6, 10, 0), (1, 33, 16, 8, 8, 2), (1, 20, 3, 9, 11, 1),      # Co <= 16: the half-tile kernel
          (12, 512, 512, 8, 26, 1), (3, 200, 96, 6, 10, 1), (2, 130, 40, 8, 8, 0), (4, 256, 64, 4, 14, 2)]   # few tiles, many channels: channel splits


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", SHAPES)
def test_wino_conv_matches_float64(shape):
    B, Ci, Co, H, W, P = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, 3, 3, device=dev()) / (3.0 * Ci ** 0.5)
    y = ops.wino_conv3x3(x, w, P)
    ref = F.conv2d(x.double(), w.double(), None, 1, P)
    assert y.shape == ref.shape
    err, scale = float((y.double() - ref).abs().max()), float(ref.abs().max())
    assert err <= 2e-5 * scale + 1e-12, (err, scale)
    assert torch.equal(y, ops.wino_conv3x3(x, w, P))
    if P in (0, 1):      # data gradient of the same convolution: gy [B,Co,Ho,Wo] -> gx [B,Ci,H,W] (full correlation when P = 0)
        gy = torch.randn(*ref.shape, device=dev())
        gx = ops.wino_conv3x3(gy, w, 1 if P == 1 else 2, transposed=True)
        rgx = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [P, P], [1, 1], False, [0, 0], 1,
                                                  [True, False, False])[0]
        assert gx.shape == x.shape
        err, scale = float((gx.double() - rgx).abs().max()), float(rgx.abs().max())
        assert err <= 2e-5 * scale + 1e-12, (err, scale)


def test_wino_conv_nonfinite_and_errors():
    from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib
    x = torch.randn(1, 4, 6, 8, device=dev())
    w = torch.randn(8, 4, 3, 3, device=dev())
    x[0, 1, 2, 3] = float("nan")
    y = ops.wino_conv3x3(x, w, 1)
    ref = F.conv2d(x, w, None, 1, 1)
    assert bool(torch.isnan(y[0, :, 1:4, 2:5]).all()) and bool(torch.isnan(ref[0, :, 1:4, 2:5]).all())
    assert get_lib().dfe_wino_conv3x3(None, None, None, 0, None, 0, 1, 1, 1, 4, 4, 1, 0, None) == -1
    assert get_lib().dfe_wino_weight_floats(5, 33) == 64 * 5 * 16


@pytest.mark.parametrize("shape", [(2, 32, 32, 16, 32, 2), (1, 40, 36, 32, 64, 4), (1, 16, 24, 32, 48, 8), (2, 33, 17, 8, 12, 2), (1, 8, 8, 64, 64, 16)])
def test_wino_dilated_conv_matches_float64(shape):
    """dilated 3x3 convolutions (padding = dilation; PWC's context network): forward and data gradient on the phase images."""
    B, Ci, Co, H, W, d = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, 3, 3, device=dev()) / (3.0 * Ci ** 0.5)
    gy = torch.randn(B, Co, H, W, device=dev())
    y = ops.wino_conv3x3(x, w, dilation=d)
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    gx = ops.wino_conv3x3(gy, w, transposed=True, dilation=d)
    rgx = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
                                              [True, False, False])[0]
    assert float((gx.double() - rgx).abs().max()) <= 2e-5 * float(rgx.abs().max())


WGRAD_SHAPES = [(2, 64, 64, 16, 24, 1, 1), (3, 40, 70, 10, 12, 1, 1), (2, 33, 17, 8, 6, 0, 1), (1, 3, 5, 4, 4, 1, 1), (2, 96, 32, 34, 50, 0, 1),
                (4, 128, 128, 32, 104, 1, 1), (1, 16, 16, 64, 208, 1, 1), (1, 70, 35, 7, 9, 1, 1), (2, 5, 3, 3, 3, 1, 1), (1, 65, 65, 5, 31, 0, 1),
                (2, 130, 40, 9, 13, 1, 1), (1, 64, 64, 13, 27, 1, 1), (1, 64, 64, 4, 106, 0, 1), (2, 224, 64, 12, 50, 1, 1),
                (1, 32, 32, 16, 32, 2, 2), (1, 40, 36, 32, 64, 4, 4), (2, 33, 17, 8, 12, 2, 2), (1, 24, 16, 32, 48, 8, 8), (1, 8, 8, 64, 64, 16, 16)]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_wino_wgrad_matches_float64(shape):
    """the Winograd-domain weight gradient (csrc/ops_wino_wgrad.hip) against float64 aten: padding 1 and valid, dilated layers (on
    their phase images), sizes off the 2x2 tiles and the 12-tile chunks, channel counts off the 32 / 64 tiles -- with every
    wave tile and chunk size the kernel is built for.  3e-5 of scale (measured <= 1e-6); bit-reproducible."""
    from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib
    B, Ci, Co, H, W, P, d = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, 3, 3, device=dev())
    gy = torch.randn(B, Co, H if d > 1 else H + 2 * P - 2, W if d > 1 else W + 2 * P - 2, device=dev())
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [P, P], [d, d], False, [0, 0], 1,
                                              [False, True, False])[1]
    scale = float(ref.abs().max())
    lib = get_lib()
    try:
        for tile, chunk in ((0, 12), (11, 12), (11, 8), (21, 12), (21, 8), (12, 8)):
            assert lib.dfe_wino_wgrad_tune(tile, 0, 0, chunk) == 0
            gw = ops.wino_wgrad3x3(x, gy, 1 if d > 1 else P, d)
            err = float((gw.double() - ref).abs().max())
            assert err <= 3e-5 * scale, (tile, chunk, err, scale)
            assert torch.equal(gw, ops.wino_wgrad3x3(x, gy, 1 if d > 1 else P, d)), (tile, chunk)
    finally:
        lib.dfe_wino_wgrad_tune(0, 0, 0, 12)


def test_wino_wgrad_reads_batch_strided_views_and_rejects_bad_arguments():
    """x / gy as channel slices of wider buffers (the PWC decoder's concatenated tensors) are read in place; bad arguments
    come back as error codes."""
    from unsupervised_depth_opticalflow_egomotion_amd._lib import get_lib
    torch.manual_seed(3)
    xb = torch.randn(2, 80, 12, 20, device=dev())
    gb = torch.randn(2, 50, 12, 20, device=dev())
    x, gy = xb[:, 16:], gb[:, :40]
    gw = ops.wino_wgrad3x3(x, gy, 1)
    assert torch.equal(gw, ops.wino_wgrad3x3(x.contiguous(), gy.contiguous(), 1))
    lib = get_lib()
    assert lib.dfe_wino_wgrad3x3(None, 0, None, 0, None, None, 1, 1, 1, 4, 4, 1, None) == -1
    assert lib.dfe_wino_wgrad_floats(1, 8, 8, 4, 4, 2) == 0 and lib.dfe_wino_wgrad_tune(22, 0, 0, 0) != 0


@pytest.mark.parametrize("shape", [(2, 64, 64, 64, 208, 1, 1), (1, 115, 128, 32, 104, 1, 1), (2, 96, 32, 34, 50, 0, 1), (3, 17, 33, 7, 9, 1, 1),
                                   (2, 24, 16, 12, 20, 1, 1), (12, 512, 512, 8, 26, 1, 1), (2, 130, 40, 8, 8, 0, 1), (2, 32, 32, 16, 32, 1, 4),
                                   (1, 40, 20, 16, 48, 1, 8)])
def test_wino_fused_epilogue_is_the_separate_pass_bit_for_bit(shape):
    """dfe_wino_conv3x3_u_act (round 5): bias + LeakyReLU inside the output transform, written to one or two destination
    buffers at channel offsets -- equal, bit for bit, to dfe_bias_act_fwd applied to the plain kernel's output (plain,
    half-tile, channel-split, valid and dilated forms; cached and per-call filters)."""
    B, Ci, Co, H, W, P, d = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, device=dev())
    w = torch.randn(Co, Ci, 3, 3, device=dev()) / (3.0 * Ci ** 0.5)
    bias = torch.randn(Co, device=dev())
    plain = ops.wino_conv3x3(x, w, P, dilation=d)
    for slope in (0.1, 0.0, 1.0):
        ref = ops.bias_act(plain.clone(), bias, slope)
        y = ops.wino_conv3x3(x, w, P, dilation=d, bias=bias, slope=slope)
        assert torch.equal(y, ref), slope
    ref = ops.bias_act(plain.clone(), bias, 0.1)
    d1 = torch.full((B, Co + 5, *plain.shape[2:]), 7.0, device=dev())
    d2 = torch.full((B, Co + 3, *plain.shape[2:]), 9.0, device=dev())
    r = ops.wino_conv3x3(x, w, P, dilation=d, bias=bias, slope=0.1, out=d1, out_off=2, out2=d2, out2_off=3)
    assert r is d1 and torch.equal(d1[:, 2:2 + Co], ref) and torch.equal(d2[:, 3:], ref)
    assert bool((d1[:, :2] == 7.0).all()) and bool((d1[:, 2 + Co:] == 7.0).all()) and bool((d2[:, :3] == 9.0).all())
    assert torch.equal(ops.wino_conv3x3(x, w, P, dilation=d, slope=0.1), ops.bias_act(plain.clone(), None, 0.1))     # no bias


def test_conv_bias_act_function_matches_the_two_node_graph():
    """ops.ConvBiasActFn (net_utils.conv on the Winograd kernel: one launch forward, one autograd node) against
    bias_act(conv2d(...)): outputs and the input / bias gradients equal bit for bit (same kernels underneath)."""
    from unsupervised_depth_opticalflow_egomotion_amd import convs
    torch.manual_seed(21)
    for (B, Ci, Co, H, W, d) in [(2, 32, 48, 32, 104, 1), (1, 64, 64, 32, 64, 2)]:
        conv = torch.nn.Conv2d(Ci, Co, 3, 1, d, d).to(dev())
        x = torch.randn(B, Ci, H, W, device=dev(), requires_grad=True)
        assert ops.conv_bias_act_eligible(x, conv)
        y = ops.conv_bias_act(x, conv, 0.1)
        g = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad(y, [x, conv.weight, conv.bias], g)
        y2 = ops.bias_act(convs.conv2d(x, conv.weight, None, 1, d, d), conv.bias, 0.1)
        gx2, gw2, gb2 = torch.autograd.grad(y2, [x, conv.weight, conv.bias], g)
        assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gb, gb2)
        # (small layers' weight gradients are MIOpen's: split-K float atomics, not reproducible bit for bit)
        assert float((gw - gw2).abs().max()) <= 1e-5 * float(gw2.abs().max())


# ---- transformed filters kept across steps (ops.WinoWeightCache, dfe_wino_transform_weights_multi / dfe_wino_conv3x3_u)
CACHE_SHAPES = [(2, 64, 64, 64, 208, 1, 1), (1, 115, 128, 32, 104, 1, 1), (2, 96, 32, 34, 50, 0, 1), (3, 17, 33, 7, 9, 1, 1),
                (2, 24, 16, 12, 20, 1, 1), (12, 512, 512, 8, 26, 1, 1), (2, 130, 40, 8, 8, 0, 1), (2, 32, 32, 16, 32, 1, 4),
                (1, 40, 20, 16, 48, 1, 8)]


def _both(x, gy, w, P, d):
    y = ops.wino_conv3x3(x, w, P, dilation=d)
    gx = ops.wino_conv3x3(gy, w, 1 if P == 1 else 2, transposed=True, dilation=d)
    return y, gx


def test_cached_filters_are_bit_identical_to_the_per_call_transform():
    """Every parameter the kernel has seen is transformed by ONE launch at refresh(); calls then read the cached U: outputs
    (forward and data gradient; plain, half-tile, channel-split, valid and dilated forms) equal the per-call path bit for bit."""
    cache = ops.WinoWeightCache()
    old, ops.wino_weights = ops.wino_weights, cache
    try:
        assert cache.enabled
        cases = []
        for shape in CACHE_SHAPES:
            B, Ci, Co, H, W, P, d = shape
            torch.manual_seed(sum(shape))
            x = torch.randn(B, Ci, H, W, device=dev())
            w = torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, device=dev()) / (3.0 * Ci ** 0.5))
            Ho, Wo = (H, W) if d > 1 else (H + 2 * P - 2, W + 2 * P - 2)
            gy = torch.randn(B, Co, Ho, Wo, device=dev())
            with torch.no_grad():
                cases.append((x, gy, w, P, d, _both(x, gy, w, P, d)))       # misses: registered, transformed per call
        assert cache.hits == 0 and cache.misses == 2 * len(cases) and len(cache.entries) == len(cases)
        cache.refresh()
        assert cache.blockmap.numel() == sum(int(ops.get_lib().dfe_wino_transform_blocks(c[2].shape[1], c[2].shape[0])) +
                                             int(ops.get_lib().dfe_wino_transform_blocks(c[2].shape[0], c[2].shape[1])) for c in cases)
        for x, gy, w, P, d, (y0, gx0) in cases:
            with torch.no_grad():
                y1, gx1 = _both(x, gy, w, P, d)
            assert torch.equal(y0, y1) and torch.equal(gx0, gx1), (tuple(w.shape), P, d)
        assert cache.hits == 2 * len(cases)
    finally:
        ops.wino_weights = old


def test_cached_filters_miss_when_the_parameter_changes():
    """In-place torch updates bump the version counter: the stale entry is not used (the call transforms for itself) until
    the next refresh; a parameter that died or moved to other storage is dropped; plain tensors are never registered."""
    cache = ops.WinoWeightCache()
    old, ops.wino_weights = ops.wino_weights, cache
    try:
        torch.manual_seed(5)
        x = torch.randn(2, 48, 20, 36, device=dev())
        w = torch.nn.Parameter(torch.randn(40, 48, 3, 3, device=dev()) * 0.05)
        with torch.no_grad():
            ops.wino_conv3x3(x, w, 1)
            cache.refresh()
            y_old = ops.wino_conv3x3(x, w, 1)
            assert cache.hits == 1
            w.mul_(-2.0)                                    # version bump
            y_new = ops.wino_conv3x3(x, w, 1)
            assert cache.hits == 1 and torch.equal(y_new, ops.wino_conv3x3(x, w.detach().clone(), 1))
            assert float((y_new + 2.0 * y_old).abs().max()) <= 1e-4 * float(y_old.abs().max())
            cache.refresh()
            assert torch.equal(ops.wino_conv3x3(x, w, 1), y_new) and cache.hits == 2
            cache.invalidate()
            assert torch.equal(ops.wino_conv3x3(x, w, 1), y_new) and cache.hits == 2
            n = len(cache.entries)
            ops.wino_conv3x3(x, w.detach().clone(), 1)      # not a Parameter: not registered
            assert len(cache.entries) == n
            w.data = w.data.clone()                          # other storage: the old entry goes at the next refresh
            ops.wino_conv3x3(x, w, 1)
            cache.refresh()
            assert len(cache.entries) == 1 and torch.equal(ops.wino_conv3x3(x, w, 1), y_new) and cache.hits == 3
            del w
            cache.refresh()
            assert len(cache.entries) == 0
    finally:
        ops.wino_weights = old


def test_cache_survives_models_rebuilt_at_the_same_addresses():
    """a model built again after the old one was freed gets the same blocks from the caching allocator,
    so its Parameters re-register under the SAME keys.  The device table of the previous membership must not be reused (its U
    pointers belong to freed entries): build, train and free the same layer three times -- hits every time, outputs equal
    to the per-call transform, and the optimiser's state untouched by stray writes."""
    from unsupervised_depth_opticalflow_egomotion_amd import optim
    cache = ops.WinoWeightCache()
    old, ops.wino_weights = ops.wino_weights, cache
    try:
        torch.manual_seed(11)
        x = torch.randn(2, 64, 24, 40, device=dev())
        ptrs = []
        for rebuild in range(3):
            w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev()) * 0.05)
            ptrs.append(w.data_ptr())
            opt = optim.FusedAdam([w], lr=1e-3)
            hits0 = cache.hits
            for step in range(3):
                y = ops.wino_conv3x3(x, w, 1)
                with torch.no_grad():
                    cache_off = ops.WinoWeightCache.transform_now(w, False)
                U = cache.lookup(w, False)
                if step > 0:
                    assert U is not None and torch.equal(U, cache_off), (rebuild, step)
                w.grad = torch.randn_like(w) * 0.01
                opt.step()                                   # refreshes the cache
                assert torch.isfinite(opt.state[w]["exp_avg"]).all() and torch.isfinite(opt.state[w]["exp_avg_sq"]).all()
                ref = ops.WinoWeightCache.transform_now(w, False)
                assert torch.equal(cache.lookup(w, False), ref), (rebuild, step)
                m = opt.state[w]["exp_avg"].clone()
                torch.cuda.synchronize()
                assert torch.equal(m, opt.state[w]["exp_avg"])
            assert cache.hits > hits0
            del w, opt, y, U
            torch.cuda.synchronize()
        # (the caching allocator normally hands the same block back: the scenario of the finding; not asserted, it is its business)
        assert len(cache.entries) <= 1
    finally:
        ops.wino_weights = old


def test_cache_verify_mode_catches_a_raw_data_write():
    """A ``p.data.mul_()`` between optimiser steps does not bump the version counter: the cached filters are stale and the
    convolution would silently use them.  DFE_WINO_CACHE_VERIFY=1 turns that into an error; ``invalidate()`` is the remedy."""
    from unsupervised_depth_opticalflow_egomotion_amd._lib import DfeError
    cache = ops.WinoWeightCache()
    cache.verify = True
    old, ops.wino_weights = ops.wino_weights, cache
    try:
        torch.manual_seed(12)
        x = torch.randn(1, 32, 16, 24, device=dev())
        w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=dev()) * 0.05)
        with torch.no_grad():
            ops.wino_conv3x3(x, w, 1)
            cache.refresh()
            y0 = ops.wino_conv3x3(x, w, 1)                   # verified hit
            assert cache.hits == 1
            w.data.mul_(2.0)                                 # raw write: no version bump
            with pytest.raises(DfeError):
                ops.wino_conv3x3(x, w, 1)
            cache.invalidate()
            y1 = ops.wino_conv3x3(x, w, 1)                   # miss: per-call transform of the new weights
            assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * float(y1.abs().max())
            cache.refresh()
            assert torch.equal(ops.wino_conv3x3(x, w, 1), y1)
    finally:
        ops.wino_weights = old


def test_fused_adam_refreshes_the_cached_filters():
    """optim.FusedAdam writes parameters through raw pointers (no version bump) and therefore rebuilds the cache itself: in a
    convolution layer trained for three steps, every step's output and input gradient (cached filters from step 2 on) equal
    the per-call transform of the weights of that moment, bit for bit."""
    from unsupervised_depth_opticalflow_egomotion_amd import convs, optim

    cache = ops.WinoWeightCache()
    old, ops.wino_weights = ops.wino_weights, cache
    try:
        torch.manual_seed(11)
        conv = torch.nn.Conv2d(64, 64, 3, 1, 1).to(dev())
        x = torch.randn(4, 64, 64, 104, device=dev())
        opt = optim.FusedAdam(conv.parameters(), lr=1e-2)
        w_before = conv.weight.detach().clone()
        for step in range(3):
            opt.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = convs.conv2d(xi, conv.weight, None, 1, 1)
            gy = torch.randn_like(y)
            y.backward(gy)
            w_now = conv.weight.detach().clone()               # a plain tensor: never cached
            assert torch.equal(y.detach(), ops.wino_conv3x3(x, w_now, 1))
            assert torch.equal(xi.grad, ops.wino_conv3x3(gy, w_now, 1, transposed=True))
            assert (cache.hits, cache.misses) == (2 * step, 2 + 2 * (step + 1))     # step 0 transforms per call, then the cache
            opt.step()
        assert not torch.equal(conv.weight.detach(), w_before)
    finally:
        ops.wino_weights = old
