"""The one door to the networks' convolutions: MIOpen's, or this build's fp32 matrix-core kernels where they win -- the
fused Winograd F(2x2, 3x3) kernel for the forward pass and the data gradient of the 3x3 stride-1 layers (csrc/ops_wino.hip,
round 4), its weight gradient in the same domain (csrc/ops_wino_wgrad.hip, round 5).  Everything is fp32.

``raw_forward`` / ``raw_backward`` are the same decisions for code that already lives inside an autograd Function
(ops.DenseDecodeFn, ops.ThinConv3x3Fn, ops.ConvBiasActFn).

Removed in round 5 (measured, never the headline, slower than fp32 since the Winograd kernel: 24.05 against 21.63 ms in round
4): the opt-in bf16 mode (``--amp bf16``: casts at every convolution's door, MIOpen's bf16 kernels).  What it would take to
pay -- activations kept bf16 channels-last between the convolutions, ~25 glue kernels in a second dtype and layout -- is
priced in profiles/r04_nhwc_probe.txt; git history holds the code (convs.py, csrc/ops_cast.hip at round 4's HEAD).

Measured and not adopted (round 3, profiles/r03_conv_swaps.txt): computing a stride-1 "same" convolution's forward pass
with MIOpen's backward-data kernels on the flipped, channel-transposed weights (and vice versa): 458 -> 432 us at best."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import miopen_tuning

miopen_tuning.activate()      # before the first convolution: MIOpen reads MIOPEN_USER_DB_PATH when it first opens its databases


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


_cb = torch.ops.aten.convolution_backward

# Large 3x3 stride-1 layers in fp32: this build's fused Winograd F(2x2, 3x3) kernel on the matrix cores (csrc/ops_wino.hip,
# ops.wino_conv3x3) for the forward pass and the data gradient -- MIOpen runs the same algorithm on the vector ALU and is
# 1.2-1.7x slower where there are enough 2x2 output tiles to fill the chip (tools/wino_bench.py); the weight gradient stays
# MIOpen's.  DFE_WINO_MIN_TILES: smallest B * ceil(Ho/2) * ceil(Wo/2) that takes the kernel (0 = never).
WINO_MIN_TILES = int(os.environ.get("DFE_WINO_MIN_TILES", "500"))
WINO_MIN_CHANNELS = int(os.environ.get("DFE_WINO_MIN_CHANNELS", "16"))
WINO_DILATED = os.environ.get("DFE_WINO_DILATED", "1") == "1"     # dilated layers too (instead of MIOpen / the phase-image path)


def _wino_eligible(x, w_shape, cin, stride, padding, dilation, groups=1):
    """x [B,cin,H,W] convolved 3x3 / stride 1 with ``padding`` in {0, 1} (or dilated with padding = dilation dividing H and
    W): enough tiles and reduction channels?"""
    if (WINO_MIN_TILES <= 0 or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4
            or groups != 1 or tuple(w_shape[2:]) != (3, 3) or stride != (1, 1) or dilation[0] != dilation[1]):
        return False
    B, C, H, W = x.shape
    d = dilation[0]
    if d > 1:
        if padding != (d, d) or H % d or W % d or H // d < 2 or W // d < 2 or not WINO_DILATED:
            return False
        Ho, Wo = H // d, W // d
        B = B * d * d
    else:
        if padding not in ((0, 0), (1, 1)):
            return False
        P = padding[0]
        Ho, Wo = H + 2 * P - 2, W + 2 * P - 2
    if C != cin or Ho < 1 or Wo < 1:
        return False
    return (B * ((Ho + 1) // 2) * ((Wo + 1) // 2) >= WINO_MIN_TILES and C >= WINO_MIN_CHANNELS and x.numel() < (1 << 30))


# Weight gradient in the Winograd domain (dfe_wino_wgrad3x3, csrc/ops_wino_wgrad.hip; round 5): raw NCHW rows staged in LDS, every
# wave transforms the tiles of its (channel, tile) lanes in registers and feeds the fp32 matrix cores -- no layout transposes, no
# zero fill, no atomics (MIOpen: NHWC implicit GEMM + three batched transposes + a fill per call, split-K float atomics).
# 1.2-3.0x MIOpen on every 3x3 stride-1 layer of the step with >= 16 channels on both sides (tools/wgrad_bench.py,
# profiles/r05_wgrad_bench.md).  DFE_WINO_WGRAD=0: MIOpen's weight gradients everywhere.
WINO_WGRAD = os.environ.get("DFE_WINO_WGRAD", "1") != "0"
WINO_WGRAD_MIN_MACS = float(os.environ.get("DFE_WINO_WGRAD_MIN_GMAC", "0.8")) * 1e9      # direct multiply-adds of the layer
# dilated layers: the same kernel on the d x d phase images (two strided copies, ops.phase_images) up to this dilation -- 247 / 270 us
# against MIOpen's 359 / 342 at dilation 2 / 4 (8 x 128 -> 128 @ 64x208); at 8 / 16 the phase images are 8x26 / 4x13 pixels, too
# short for the kernel's 12-tile chunks (307 / 211 against 297 / 144): those two stay on MIOpen
WINO_WGRAD_MAX_DILATION = int(os.environ.get("DFE_WINO_WGRAD_MAX_DILATION", "4"))


def _wino_wgrad_eligible(x, gy_shape, w_shape, padding, d):
    """x [B,Ci,H,W] (fp32, HIP), the output gradient's shape, the filter's shape, padding pair, dilation."""
    if not WINO_WGRAD or x.dtype != torch.float32 or not x.is_cuda or x.dim() != 4:
        return False
    if d > 1 and not (d <= WINO_WGRAD_MAX_DILATION and padding == (d, d) and x.shape[2] % d == 0 and x.shape[3] % d == 0):
        return False
    if d == 1 and padding not in ((0, 0), (1, 1)):
        return False
    Co, Ci = int(w_shape[0]), int(w_shape[1])
    if tuple(w_shape[2:]) != (3, 3) or Ci != x.shape[1] or min(Co, Ci) < 16:      # (16 -> 16 at 128x416 x 12: 83 against MIOpen's 129 us)
        return False
    n_out = gy_shape[0] * Co * gy_shape[2] * gy_shape[3]
    return float(n_out) * Ci * 9 >= WINO_WGRAD_MIN_MACS and x.numel() < (1 << 30) and n_out < (1 << 30)


def _wgrad_route(x, w_shape, stride, padding, dilation, groups):
    """A layer whose forward pass stays on MIOpen but whose weight gradient takes dfe_wino_wgrad3x3 (few tiles, many channels)."""
    if groups != 1 or stride != (1, 1) or dilation[0] != dilation[1] or x.dim() != 4:
        return False
    d = dilation[0]
    ho = x.shape[2] if d > 1 else x.shape[2] + 2 * padding[0] - 2
    wo = x.shape[3] if d > 1 else x.shape[3] + 2 * padding[1] - 2
    return ho >= 1 and wo >= 1 and _wino_wgrad_eligible(x, (x.shape[0], w_shape[0], ho, wo), w_shape, padding, d)


# Weight gradients of the stride-2 layers with thin inputs (the ResNet18 encoder's 7x7 stem depth_model.py:60-95, FeaturePyramid's
# 3 -> 16 and 16 -> 32 layers feature_pyramid.py:7-36, PoseCNN's 7x7 x 9 and 5x5 x 16 layers pose_cnn.py:14-36): csrc/ops_sconv.hip
# reads NCHW directly -- one launch plus the split sum, where MIOpen wrapped an NHWC implicit GEMM in three layout transposes and a
# zero fill and ran at 6 - 40 TFLOP/s: 164 against 328 us (12 x 3 -> 64 @ 256x832, 7x7), 63 against 138 us (4 x 9 -> 16, 7x7);
# tools/sconv_bench.py, profiles/r05_sconv_bench_all.md.  Step: 19.00 -> 18.90 ms.  Wider layers (MIOpen at 64 - 70 TFLOP/s, this
# kernel at 50 - 57) and the forward / data-gradient kernels of the same family lost in the step and are not routed
# (EXPERIMENT_LOG.md "strided convolutions").  DFE_SCONV=0: back to MIOpen.
SCONV = os.environ.get("DFE_SCONV", "1") != "0"
SCONV_WGRAD_MAX_CI = int(os.environ.get("DFE_SCONV_WGRAD_MAX_CI", "16"))


def _sconv_route(x, w_shape, stride, padding, dilation, groups=1):
    """A stride-2 k x k layer (k in {3, 5, 7}, padding k // 2) of fp32 NCHW tensors on the GPU whose weight gradient
    dfe_sconv_wgrad takes."""
    k = int(w_shape[2])
    if not (SCONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and groups == 1 and tuple(stride) == (2, 2)
            and tuple(dilation) == (1, 1) and int(w_shape[3]) == k and k in (3, 5, 7) and tuple(padding) == (k // 2, k // 2)
            and int(w_shape[1]) <= SCONV_WGRAD_MAX_CI and x.numel() < (1 << 30)):
        return False
    from . import ops
    return ops.sconv_wgrad_supported(x.shape, w_shape[0], k, 2, k // 2)


def raw_forward(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """y = conv(x, w) without bias, fp32 in / fp32 out, no autograd of its own."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if _wino_eligible(x, w.shape, w.shape[1], stride, padding, dilation):
        from . import ops
        return ops.wino_conv3x3(x, w, padding[0], dilation=dilation[0])
    return F.conv2d(x, w, None, stride, padding, dilation)


def raw_backward(gy, x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1), want_x=True, want_w=True, want_b=False):
    """(gx, gw, gb) of y = conv(x, w) + b; fp32 in / fp32 out."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    bias_sizes = [int(w.shape[0])] if want_b else None
    d = dilation[0]
    own_w = want_w and stride == (1, 1) and tuple(w.shape[2:]) == (3, 3) and dilation == (d, d) and \
        _wino_wgrad_eligible(x, gy.shape, w.shape, padding, d)
    gx = gw = gb = None
    if want_w and _sconv_route(x, w.shape, stride, padding, dilation):
        from . import ops
        gw = ops.sconv_wgrad(x, gy, int(w.shape[2]), 2, int(w.shape[2]) // 2)
        want_w = False
    if want_x and (padding in ((1, 1), (0, 0)) or (d > 1 and padding == (d, d))) and \
            _wino_eligible(gy, w.shape, w.shape[0], stride, (d, d) if d > 1 else (1, 1), dilation):
        from . import ops
        # the data gradient = the same kernel on the transposed filter (full correlation, padding 2, for a valid convolution)
        gx = ops.wino_conv3x3(gy, w, 1 if padding != (0, 0) else 2, transposed=True, dilation=d)
        want_x = False
    if own_w:
        from . import ops
        gw = ops.wino_wgrad3x3(x, gy, padding[0] if d == 1 else 1, d)
        want_w = False
    if want_x or want_w or want_b:
        r = _cb(gy, x, w, bias_sizes, list(stride), list(padding), list(dilation), False, [0, 0], 1, [want_x, want_w, want_b])
        gx = r[0] if want_x else gx
        gw = r[1] if want_w else gw
        gb = r[2] if want_b else gb
    return gx, gw, gb


# Dilated 3x3 "same" convolutions (PWC's context network, pwc_tf.py:31-36: dilation 2, 4, 8, 16 at 64x208) as a dense 3x3
# convolution on the d*d phase images: pixel (y, x) of a dilation-d convolution only ever meets pixels (y + i d, x + j d),
# i.e. the d*d sub-images x[:, :, p::d, q::d] are convolved independently with the same 3x3 weights, padding 1 (= d
# full-resolution pixels).  Same products, same sums; what changes is the kernel MIOpen can use: the dilated problem
# goes to its implicit-GEMM kernels (51-94 TFLOP/s, dilation 8 forward 458 us), the phase images to Winograd (204 us).
# Measured per layer (B = 8, tools/dilated_conv_bench.py): forward 379 / 345 / 458 / 153 -> 290 / 285 / 204 / 127 us, data
# gradient 365 / 342 / 277 / 252 -> 308 / 297 / 220 / 137 us, weight gradient unchanged; the two re-layout copies each
# way cost 21-52 us.  In the step (profiles/r03_dilated_phase_conv.txt): 26.81 ms without, 26.60 with every dilated layer on
# the phase path, 26.62 with dilation >= 8 only -- the default: the two layers that gain most, half the copies.
PHASE_MIN_DILATION = int(os.environ.get("DFE_PHASE_CONV", "8"))      # smallest dilation that takes the phase path; 0 = off


def _phase_eligible(x, w, stride, padding, dilation, groups):
    d = dilation[0]
    return (PHASE_MIN_DILATION > 0 and x.is_cuda and groups == 1 and d >= PHASE_MIN_DILATION and dilation == (d, d)
            and stride == (1, 1) and padding == (d, d) and tuple(w.shape[2:]) == (3, 3) and x.dim() == 4
            and x.shape[2] % d == 0 and x.shape[3] % d == 0)


def _phase_conv(x, w, d):
    B, C, H, W = x.shape
    xs = x.view(B, C, H // d, d, W // d, d).permute(0, 3, 5, 1, 2, 4).reshape(B * d * d, C, H // d, W // d)
    ys = conv2d(xs, w, None, 1, 1, 1)
    return ys.view(B, d, d, w.shape[0], H // d, W // d).permute(0, 3, 4, 1, 5, 2).reshape(B, w.shape[0], H, W)


class _RawConvFn(torch.autograd.Function):
    """conv2d without bias through raw_forward / raw_backward (fp32 layers that take this build's Winograd kernel)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation):
        ctx.cfg = (stride, padding, dilation)
        ctx.save_for_backward(x, w)
        return raw_forward(x, w, stride, padding, dilation)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation = ctx.cfg
        gx, gw, _ = raw_backward(gy.contiguous(), x, w, stride, padding, dilation, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """``F.conv2d`` with this build's kernels where they are eligible (fp32 HIP tensors, groups == 1)."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if groups == 1 and dilation[0] > 1 and _wino_eligible(x, w.shape, w.shape[1], stride, padding, dilation):
        y = _RawConvFn.apply(x, w, stride, padding, dilation)
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    if dilation[0] > 1 and _phase_eligible(x, w, stride, padding, dilation, groups):
        y = _phase_conv(x, w, dilation[0])
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    if groups == 1 and (_wino_eligible(x, w.shape, w.shape[1], stride, padding, dilation) or
                        _wgrad_route(x, w.shape, stride, padding, dilation, groups) or
                        _sconv_route(x, w.shape, stride, padding, dilation, groups)):
        y = _RawConvFn.apply(x, w, stride, padding, dilation)
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    return F.conv2d(x, w, bias, stride, padding, dilation, groups)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters and state-dict keys) whose forward goes through ``conv2d`` above."""

    def forward(self, x):
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            return super().forward(x)
        if (self.out_channels == 2 and x.is_cuda and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1):
            from . import ops                      # PWC's flow heads (pwc_tf.py:39-40): rolling-window HIP kernels
            if ops.flow_head_eligible(x, self.weight, self.bias):
                return ops.FlowHeadFn.apply(x, self.weight, self.bias)
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
