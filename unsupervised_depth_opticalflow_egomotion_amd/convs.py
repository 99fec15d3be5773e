"""The one door to MIOpen's convolutions (the networks' FLOPs; everything else on the step is this build's HIP code).

What is decided here, per call and invisible to the modules' state dicts, is the **compute dtype** (``compute_dtype``):
``None`` = fp32, the default and the only mode the parity contract and the headline bench line use.  ``torch.bfloat16`` is
the opt-in mixed-precision mode of SURVEY.md 8(f) rank 1 (``train.py --amp bf16`` / ``bench.py --amp bf16``): activations
and weights are cast to bf16 at the convolution's door, MIOpen runs on the bf16 matrix cores with fp32 accumulation, and
the result is cast back -- every glue kernel, every normalisation, the whole loss stack and the optimiser stay fp32 (master
weights are the fp32 parameters themselves).

``raw_forward`` / ``raw_backward`` are the same decision for code that already lives inside an autograd Function
(ops.DenseDecodeFn, ops.ThinConv3x3Fn).

Measured and not adopted (round 3, profiles/r03_conv_swaps.txt): computing a stride-1 "same" convolution's forward pass
with MIOpen's backward-data kernels on the flipped, channel-transposed weights (and vice versa).  The census of isolated
calls (profiles/r03_conv_census.txt) shows lopsided pairs (dilation 8: forward 51 TFLOP/s, backward-data 87), but the
transposed problem lands on the same kernels: 458 -> 432 us at best, within noise everywhere else."""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

_STATE = {"dtype": None}


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


_cb = torch.ops.aten.convolution_backward


def raw_forward(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """y = conv(x, w) without bias, fp32 in / fp32 out, no autograd of its own."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    dt = _STATE["dtype"]
    if dt is not None:
        return F.conv2d(x.to(dt), w.to(dt), None, stride, padding, dilation).float()
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
    return _cb(gy, x, w, bias_sizes, list(stride), list(padding), list(dilation), False, [0, 0], 1, [want_x, want_w, want_b])


class _ConvFn(torch.autograd.Function):
    """conv2d without bias in the reduced compute dtype (used only when one is set)."""

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


# ------------------------------------------------------------------------------------------------ weight-gradient streams
# A convolution's weight gradient is needed by nobody until the optimiser step, yet autograd's ConvolutionBackward
# computes it in line with the data gradient, in the middle of the backward pass's critical path.  With
# ``weight_grad_streams(True)`` every convolution that goes through this module is recorded as TWO autograd nodes: the
# convolution itself (its backward returns the data gradient and parks (gy, x)) and an identity on the weight that was
# applied under a side stream -- autograd replays a node's backward on the stream its forward ran on, so that node's
# backward, which computes the parked weight gradient, is enqueued on the side stream, ordered after the data-gradient
# node by the engine's own events, and the gradient accumulation on the leaf waits for it.  No manual synchronisation.
_WG = {"on": False, "streams": {}, "pending": {}, "seq": 0}


@contextlib.contextmanager
def weight_grad_streams(on=True):
    old = _WG["on"]
    _WG["on"] = bool(on)
    try:
        yield
    finally:
        _WG["on"] = old


def _wg_stream(dev):
    """The weight-gradient stream that belongs to the CURRENT compute stream (one per network chain)."""
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    st = _WG["streams"].get(key)
    if st is None:
        st = _WG["streams"][key] = torch.cuda.Stream(dev)
    return st


class _WeightOnStream(torch.autograd.Function):
    """Identity on a weight tensor, applied under the weight-gradient stream; its backward computes the weight gradient
    parked under ``key`` by the consumer's backward (``park_weight_grad``)."""

    @staticmethod
    def forward(ctx, w, key):
        ctx.key = key
        return w.view_as(w)

    @staticmethod
    def backward(ctx, _placeholder):
        job = _WG["pending"].pop(ctx.key, None)
        if job is None:               # the consumer produced no gradient for this weight
            return None, None
        fn, tensors = job
        st = torch.cuda.current_stream()
        for t in tensors:
            t.record_stream(st)       # produced on the network's stream, read here
        return fn(), None


def weight_on_stream(w):
    """(alias of w whose gradient is computed on the weight-gradient stream, key) -- or (w, None) when the mode is off."""
    if not (_WG["on"] and w.is_cuda and w.requires_grad and torch.is_grad_enabled()):
        return w, None
    _WG["seq"] += 1
    key = _WG["seq"]
    with torch.cuda.stream(_wg_stream(w.device)):
        w2 = _WeightOnStream.apply(w, key)
    return w2, key


def park_weight_grad(key, w, fn, tensors):
    """Called from a consumer's backward: defer ``fn() -> gw`` (reading ``tensors``) to ``w``'s weight-gradient node and
    return the placeholder the consumer hands to autograd as w's gradient (right shape, no memory)."""
    _WG["pending"][key] = (fn, tensors)
    return torch.zeros((), device=w.device, dtype=w.dtype).expand(w.shape)


class _ConvSplitFn(torch.autograd.Function):
    """conv2d without bias whose backward returns the data gradient and parks the weight gradient (see above)."""

    @staticmethod
    def forward(ctx, x, w2, key, stride, padding, dilation):
        dt = _STATE["dtype"]
        ctx.cfg = (key, stride, padding, dilation, dt)
        xs = x if dt is None else x.to(dt)     # cast once: the forward, the data gradient and the weight gradient share it
        y = raw_forward(xs, w2, stride, padding, dilation)
        ctx.save_for_backward(xs, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w2 = ctx.saved_tensors
        key, stride, padding, dilation, dt = ctx.cfg
        gy = gy.contiguous() if dt is None else gy.to(dt)
        gx = None
        with compute_dtype(dt):
            if ctx.needs_input_grad[0]:
                gx = raw_backward(gy, x, w2, stride, padding, dilation, True, False)[0]
        gw = None
        if ctx.needs_input_grad[1]:
            def run():
                with compute_dtype(dt):
                    return raw_backward(gy, x, w2, stride, padding, dilation, False, True)[1]
            gw = park_weight_grad(key, w2, run, (gy, x))
        return gx, gw, None, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """``F.conv2d``; in the reduced compute dtype when one is set, with the weight gradient on its own stream when
    ``weight_grad_streams`` is on (HIP tensors, groups == 1)."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if not (x.is_cuda and groups == 1):
        return F.conv2d(x, w, bias, stride, padding, dilation, groups)
    w2, key = weight_on_stream(w)
    if key is not None:
        y = _ConvSplitFn.apply(x, w2, key, stride, padding, dilation)
    elif _STATE["dtype"] is not None:
        y = _ConvFn.apply(x, w, stride, padding, dilation)
    else:
        return F.conv2d(x, w, bias, stride, padding, dilation, groups)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters and state-dict keys) whose forward goes through ``conv2d`` above."""

    def forward(self, x):
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            return super().forward(x)
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
