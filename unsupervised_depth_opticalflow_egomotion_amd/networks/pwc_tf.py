"""PWC-style coarse-to-fine flow decoder (reference pwc_tf.py:16-179).  The per-level feature warp, the 81-tap
cost volume and the concatenation behind them run as one HIP operator per level (k_warp_flow_*, k_corr_*,
k_pwc_cat_tail); the convolutions run on MIOpen.
Needs H and W divisible by 64."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..convs import Conv2d
from ..structures.net_utils import ConvAct, conv, warp_flow


class PWC_tf(nn.Module):
    def __init__(self, md=4):
        super().__init__()
        self.md = md
        self.corr = self.corr_naive
        self.leakyRELU = nn.LeakyReLU(0.1)
        nd = (2 * md + 1) ** 2
        dd = [128, 128, 96, 64, 32]
        feat = {6: 0, 5: 128, 4: 96, 3: 64, 2: 32}
        for lvl in (6, 5, 4, 3, 2):
            od = nd + (feat[lvl] + 2 if lvl < 6 else 0)
            setattr(self, "conv%d_0" % lvl, conv(od, 128, kernel_size=3, stride=1))
            setattr(self, "conv%d_1" % lvl, conv(dd[0], 128, kernel_size=3, stride=1))
            setattr(self, "conv%d_2" % lvl, conv(dd[0] + dd[1], 96, kernel_size=3, stride=1))
            setattr(self, "conv%d_3" % lvl, conv(dd[1] + dd[2], 64, kernel_size=3, stride=1))
            setattr(self, "conv%d_4" % lvl, conv(dd[2] + dd[3], 32, kernel_size=3, stride=1))
            setattr(self, "predict_flow%d" % lvl, self.predict_flow(dd[3] + dd[4]))
        self.dc_conv1 = conv(dd[4] + 2, 128, kernel_size=3, stride=1, padding=1, dilation=1)
        self.dc_conv2 = conv(128, 128, kernel_size=3, stride=1, padding=2, dilation=2)
        self.dc_conv3 = conv(128, 128, kernel_size=3, stride=1, padding=4, dilation=4)
        self.dc_conv4 = conv(128, 96, kernel_size=3, stride=1, padding=8, dilation=8)
        self.dc_conv5 = conv(96, 64, kernel_size=3, stride=1, padding=16, dilation=16)
        self.dc_conv6 = conv(64, 32, kernel_size=3, stride=1, padding=1, dilation=1)
        self.dc_conv7 = self.predict_flow(32)

    def predict_flow(self, in_planes):
        return Conv2d(in_planes, 2, kernel_size=3, stride=1, padding=1, bias=True)

    def warp(self, x, flow):
        return warp_flow(x, flow, use_mask=False)

    def corr_naive(self, input1, input2, d=4):
        """81-displacement channel-mean correlation [B,(2d+1)^2,H,W] (HIP)."""
        return ops.corr81(input1, input2, d)

    @staticmethod
    def resize(x, out_hw, mult, pre):
        """``F.interpolate(x * mult, out_hw, mode='bilinear')`` (pre) / ``F.interpolate(x, out_hw, ...) * mult`` as written
        in pwc_tf.py:118-119 and :175-178.  On the GPU one HIP kernel each way (ATen-CPU association forward, gather
        adjoint backward: the flow gradients do not depend on atomics order); ratios above 4 keep the ATen operators."""
        if x.is_cuda and out_hw[0] <= 4 * x.shape[2] and out_hw[1] <= 4 * x.shape[3]:
            return ops.resize_bilinear(x, out_hw, mult, pre)
        if pre:
            return F.interpolate(x * mult, list(out_hw), mode="bilinear", align_corners=False)
        return F.interpolate(x, list(out_hw), mode="bilinear", align_corners=False) * mult

    def level_input(self, c1, c2, up_flow):
        """``torch.cat((self.corr(c1, self.warp(c2, up_flow)), c1, up_flow), 1)`` (pwc_tf.py:119-121, repeated per level).
        On the GPU the triple is one operator (``dfe_pwc_level_fwd/bwd``): the cost volume is written into its slice
        of the concatenated tensor and the backward pass reads the slices of its gradient in place.  A subclass that
        replaces ``warp`` / ``corr_naive`` (the CPU baseline does) gets the plain composition of its own methods."""
        own = type(self).warp is PWC_tf.warp and getattr(self.corr, "__func__", None) is PWC_tf.corr_naive
        if own and c1.is_cuda:
            return ops.pwc_level_input(c1, c2, up_flow)
        return torch.cat((self.corr(c1, self.warp(c2, up_flow)), c1, up_flow), 1)

    def _dense_block_is_standard(self, lvl):
        for k in range(5):
            m = getattr(self, "conv%d_%d" % (lvl, k))
            c = m[0]
            if type(m) is not ConvAct or not isinstance(m[1], nn.LeakyReLU) or \
                    (c.kernel_size, c.stride, c.padding, c.dilation, c.groups) != ((3, 3), (1, 1), (1, 1), (1, 1), 1) or c.bias is None:
                return False
        p = getattr(self, "predict_flow%d" % lvl)
        return (p.kernel_size, p.stride, p.padding, p.dilation, p.groups) == ((3, 3), (1, 1), (1, 1), (1, 1), 1) and p.bias is not None

    def _decode(self, lvl, x):
        """pwc_tf.py:113-118 (level 6) and the same six lines of every other level -> (flow, x4).  On the GPU the block is
        one operator around its six MIOpen convolutions (ops.dense_decode): no torch.cat copies, no slice copies or
        gradient-accumulation adds in the backward pass."""
        if x.is_cuda and self._dense_block_is_standard(lvl):
            convs = [getattr(self, "conv%d_%d" % (lvl, k))[0] for k in range(5)]
            return ops.dense_decode(x, convs, getattr(self, "predict_flow%d" % lvl),
                                    getattr(self, "conv%d_0" % lvl)[1].negative_slope)
        x0 = getattr(self, "conv%d_0" % lvl)(x)
        x1 = getattr(self, "conv%d_1" % lvl)(x0)
        x2 = getattr(self, "conv%d_2" % lvl)(torch.cat((x0, x1), 1))
        x3 = getattr(self, "conv%d_3" % lvl)(torch.cat((x1, x2), 1))
        x4 = getattr(self, "conv%d_4" % lvl)(torch.cat((x2, x3), 1))
        return getattr(self, "predict_flow%d" % lvl)(torch.cat((x3, x4), 1)), x4

    def forward(self, feature_list_1, feature_list_2, img_hw):
        c1 = dict(zip((1, 2, 3, 4, 5, 6), feature_list_1))
        c2 = dict(zip((1, 2, 3, 4, 5, 6), feature_list_2))
        flow, _ = self._decode(6, self.corr(c1[6], c2[6]))
        flows = {6: flow}
        x4 = None
        for lvl in (5, 4, 3, 2):
            up = self.resize(flows[lvl + 1], [2 * flows[lvl + 1].shape[2], 2 * flows[lvl + 1].shape[3]], 2.0, pre=False)
            delta, x4 = self._decode(lvl, self.level_input(c1[lvl], c2[lvl], up))
            flows[lvl] = delta + up
        x = self.dc_conv4(self.dc_conv3(self.dc_conv2(self.dc_conv1(torch.cat([flows[2], x4], 1)))))
        flows[2] = flows[2] + self.dc_conv7(self.dc_conv6(self.dc_conv5(x)))
        h, w = img_hw[0], img_hw[1]
        return [self.resize(flows[2], [h, w], 4.0, pre=True), self.resize(flows[3], [h // 2, w // 2], 4.0, pre=True),
                self.resize(flows[4], [h // 4, w // 4], 4.0, pre=True), self.resize(flows[5], [h // 8, w // 8], 4.0, pre=True)]
