"""The one door to MIOpen's convolutions (the networks' FLOPs; everything else on the step is this build's HIP code).

Two things are decided here, both per call and both invisible to the modules' state dicts:

* **compute dtype** (``compute_dtype``): ``None`` = fp32, the default and the only mode the parity contract and the
  headline bench line use.  ``torch.bfloat16`` is the opt-in mixed-precision mode of SURVEY.md 8(f) rank 1
  (``train.py --amp bf16`` / ``bench.py --amp bf16``): activations and weights are cast to bf16 at the convolution's door,
  MIOpen runs on the bf16 matrix cores with fp32 accumulation, and the result is cast back -- every glue kernel, every
  normalisation, the whole loss stack and the optimiser stay fp32 (master weights are the fp32 parameters themselves).
* **which MIOpen kernel family computes a pass** (``ALGO_SWAPS``): a stride-1 "same" convolution's forward pass is also
  the backward-data pass of the channel-transposed, spatially flipped weights, and vice versa.  MIOpen's fp32 solvers for
  the PWC context network's dilated layers are lopsided (dilation 8 forward 51 TFLOP/s, its backward-data 87; dilation 16
  backward-data 47, its forward 79: profiles/r03_conv_census.txt), so those passes are routed through the faster
  family.  Same arithmetic up to fp32 summation order.

``raw_forward`` / ``raw_backward`` are the same two decisions for code that already lives inside an autograd Function
(ops.DenseDecodeFn, ops.ThinConv3x3Fn)."""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

_STATE = {"dtype": None}

# (kernel, dilation) -> (forward through backward-data, backward-data through forward); stride 1, padding = dilation only.
# Measured on MI355X, fp32, 8 x C x 64 x 208 (scratch/swap_bench.py -> profiles/r03_conv_swaps.txt).
ALGO_SWAPS = {}


def set_compute_dtype(dtype):
    if dtype not in (None, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be None (fp32), torch.bfloat16 or torch.float16")
    _STATE["dtype"] = dtype


def get_compute_dtype():
    return _STATE["dtype"]


@contextlib.contextmanager
def compute_dtype(dtype):
    old = _STATE["dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _STATE["dtype"] = old


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _swaps(w, stride, padding, dilation):
    if _STATE["dtype"] is not None or not ALGO_SWAPS:
        return False, False
    kh, kw = int(w.shape[2]), int(w.shape[3])
    if kh != kw or stride != (1, 1) or dilation[0] != dilation[1] or padding != (dilation[0] * (kh - 1) // 2,) * 2:
        return False, False
    return ALGO_SWAPS.get((kh, dilation[0]), (False, False))


def _flip_t(w):
    """[Co,Ci,k,k] -> [Ci,Co,k,k], taps reversed: the weights whose backward-data pass is ``w``'s forward pass."""
    return w.flip(2, 3).transpose(0, 1).contiguous()


_cb = torch.ops.aten.convolution_backward


def raw_forward(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """y = conv(x, w) without bias, fp32 in / fp32 out, no autograd of its own."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    dt = _STATE["dtype"]
    if dt is not None:
        return F.conv2d(x.to(dt), w.to(dt), None, stride, padding, dilation).float()
    if _swaps(w, stride, padding, dilation)[0]:
        y_like = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], device=x.device, dtype=x.dtype)
        return _cb(x.contiguous(), y_like, _flip_t(w), None, list(stride), list(padding), list(dilation), False, [0, 0], 1,
                   [True, False, False])[0]
    return F.conv2d(x, w, None, stride, padding, dilation)


def raw_backward(gy, x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1), want_x=True, want_w=True, want_b=False):
    """(gx, gw, gb) of y = conv(x, w) + b; fp32 in / fp32 out."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    dt = _STATE["dtype"]
    bias_sizes = [int(w.shape[0])] if want_b else None
    if dt is not None:
        gx, gw, gb = _cb(gy.to(dt), x.to(dt), w.to(dt), bias_sizes, list(stride), list(padding), list(dilation), False, [0, 0], 1,
                         [want_x, want_w, want_b])
        return (gx.float() if gx is not None else None, gw.float() if gw is not None else None,
                gb.float() if gb is not None else None)
    if want_x and _swaps(w, stride, padding, dilation)[1]:
        gx = F.conv2d(gy.contiguous(), _flip_t(w), None, stride, padding, dilation)
        _, gw, gb = _cb(gy, x, w, bias_sizes, list(stride), list(padding), list(dilation), False, [0, 0], 1, [False, want_w, want_b]) \
            if (want_w or want_b) else (None, None, None)
        return gx, gw, gb
    return _cb(gy, x, w, bias_sizes, list(stride), list(padding), list(dilation), False, [0, 0], 1, [want_x, want_w, want_b])


class _ConvFn(torch.autograd.Function):
    """conv2d without bias through raw_forward / raw_backward (used only when one of the two decisions above applies)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation):
        dt = _STATE["dtype"]
        ctx.cfg = (stride, padding, dilation, dt)
        xs = x if dt is None else x.to(dt)     # the activation is kept in the compute dtype (half the footprint in bf16)
        ctx.save_for_backward(xs, w)
        if dt is not None:
            return F.conv2d(xs, w.to(dt), None, stride, padding, dilation).float()
        return raw_forward(x, w, stride, padding, dilation)

    @staticmethod
    def backward(ctx, gy):
        xs, w = ctx.saved_tensors
        stride, padding, dilation, dt = ctx.cfg
        with compute_dtype(dt):
            gx, gw, _ = raw_backward(gy, xs, w, stride, padding, dilation, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """``F.conv2d`` with the two decisions of this module applied (HIP tensors, groups == 1); plain ``F.conv2d`` otherwise."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    special = x.is_cuda and groups == 1 and (_STATE["dtype"] is not None or any(_swaps(w, stride, padding, dilation)))
    if not special:
        return F.conv2d(x, w, bias, stride, padding, dilation, groups)
    y = _ConvFn.apply(x, w, stride, padding, dilation)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters and state-dict keys) whose forward goes through ``conv2d`` above."""

    def forward(self, x):
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            return super().forward(x)
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
