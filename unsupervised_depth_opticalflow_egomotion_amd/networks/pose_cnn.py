"""PoseCNN with the attention refinement head (reference pose_cnn.py:14-93).  Output [B, F-1, 6];
index 0 = target->left ("bwd"), 1 = target->right ("fwd").  The three Linear(14,14) layers act over
H/128 * W/128 positions, i.e. the net only accepts 256x832-like inputs (pose_cnn.py:37-39)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import convs, ops
from ..convs import Conv2d


class PoseCNN(nn.Module):
    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        n = 6 * (num_input_frames - 1)
        chans = [(3 * num_input_frames, 16, 7), (16, 32, 5), (32, 64, 3), (64, 128, 3), (128, 256, 3),
                 (256, 256, 3), (256, 256, 3)]
        net = [Conv2d(ci, co, k, 2, k // 2) for ci, co, k in chans]
        # registration order follows the reference (pose_conv before net): parameter order is what
        # optimizer_state_dict indexes, so resumed checkpoints stay compatible
        self.pose_conv = Conv2d(256, n, 1)
        self.relu = nn.ReLU(True)
        self.net = nn.ModuleList(net)
        self.query_fc = nn.Linear(14, 14)
        self.key_fc = nn.Linear(14, 14)
        self.value_fc = nn.Linear(14, 14)
        self.refine_net = nn.ModuleList([Conv2d(2 * n, n, 1, 1, 0), Conv2d(n, n, 3, 1, 1),
                                         Conv2d(n, n, 3, 1, 1), Conv2d(n, n, 3, 1, 1)])
        self.refine_pose_conv = Conv2d(n, n, 1)

    def conv_relu(self, conv, x):
        """``self.relu(conv(x))`` (pose_cnn.py:68-69, 84-85).  On the GPU the convolution runs without its bias and the bias +
        ReLU epilogue is one in-place HIP pass whose backward also yields the bias gradient (ops.bias_act, slope 0)."""
        if x.is_cuda:
            if (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                    and conv.groups == 1 and ops.planeconv_eligible(x, conv.weight)):
                return ops.planeconv_act(x, conv.weight, conv.bias, 0.0)     # the 2x7 refinement planes: this build's MFMA kernels
            if ops.conv1x1_small_eligible(x, conv):
                return ops.conv1x1_small(x, conv, 0.0)
            return ops.bias_act(convs.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups),
                                conv.bias, 0.0)
        return self.relu(conv(x))

    @staticmethod
    def _conv1x1(conv, x):
        """A 1x1 convolution without activation (pose_conv, refine_pose_conv: pose_cnn.py:32,48,88,72) on the tiny-plane kernel."""
        if x.is_cuda and ops.conv1x1_small_eligible(x, conv):
            return ops.conv1x1_small(x, conv, 1.0)
        return conv(x)

    def atten_refine(self, x):
        B, C, H, W = x.size()
        flat = x.view(B, C, H * W)
        q, k, v = self.query_fc(flat), self.key_fc(flat), self.value_fc(flat)
        attn = torch.softmax(torch.bmm(q, k.permute(0, 2, 1)), 1)
        out = torch.cat([flat, torch.bmm(attn, v)], 1).view(B, 2 * C, H, W)
        for conv in self.refine_net:
            out = self.conv_relu(conv, out)
        out = self._conv1x1(self.refine_pose_conv, out).mean(3).mean(2)
        return 0.01 * out.view(-1, self.num_input_frames - 1, 6)

    def forward(self, x):
        for conv in self.net:
            x = self.conv_relu(conv, x)
        x = self._conv1x1(self.pose_conv, x)
        delta = self.atten_refine(x)
        x = x.mean(3).mean(2)
        return 0.01 * x.view(-1, self.num_input_frames - 1, 6) + delta
