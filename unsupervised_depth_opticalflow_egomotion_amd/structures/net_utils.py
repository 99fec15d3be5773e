"""Reference-signature shims for core/networks/structures/net_utils.py (warp_flow :16-54,
conv :7-11, deconv :13-14).  warp_flow runs the HIP kernel k_warp_flow_fwd/bwd."""
import torch.nn as nn
import torch.nn.functional as F

from .. import convs, ops


class ConvAct(nn.Sequential):
    """Conv2d(bias=True) + LeakyReLU(slope) with the reference's child names (``.0`` = the Conv2d).  On the GPU the
    convolution runs on MIOpen without its bias and the bias + activation epilogue is one in-place HIP pass
    (ops.bias_act; the backward pass yields the bias gradient from the same read)."""

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        c = self[0]
        if (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1
                and ops.planeconv_eligible(x, c.weight)):   # FeaturePyramid's top levels (8x26, 4x13): small-plane MFMA kernels
            return ops.planeconv_act(x, c.weight, c.bias, self[1].negative_slope)
        if ops.conv_bias_act_eligible(x, c):        # round 5: bias + LeakyReLU inside the Winograd kernel's output transform
            return ops.conv_bias_act(x, c, self[1].negative_slope)
        z = convs.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
        return ops.bias_act(z, c.bias, self[1].negative_slope)


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    """Conv2d + LeakyReLU(0.1) block (the ``.0`` in the state-dict keys is the Conv2d)."""
    return ConvAct(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                  dilation=dilation, bias=True),
        nn.LeakyReLU(0.1))


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding, bias=True)


def warp_flow(x, flow, use_mask=False, align_corners=None):
    """Warp ``x`` [B,C,H,W] (im2) back to im1 along ``flow`` [B,2,H,W]; with ``use_mask`` every
    sample whose in-bounds bilinear weight is < 0.9999 is zeroed.  Differentiable wrt flow and x."""
    return ops.warp_flow(x, flow, use_mask=use_mask, align_corners=align_corners)
