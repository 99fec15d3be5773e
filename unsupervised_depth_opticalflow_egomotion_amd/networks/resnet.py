"""ResNet-18/34 encoder with torchvision's module names (torchvision is not installed here).

Only what the reference's ``ResnetEncoder`` touches (depth_model.py:60-95): conv1, bn1, relu, maxpool,
layer1-4 and the never-used ``fc`` (kept so that the state-dict keys -- 513 k parameters that never
receive a gradient -- match checkpoints)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..convs import Conv2d


class FrameBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state-dict keys) whose batch may hold ``groups`` independently
    normalised groups of consecutive samples -- the frames of a triplet, which the reference pushes through the depth
    net one call at a time (model_geometry.py:786-788).  Statistics are per (group, channel) and the running statistics
    receive the ``groups`` momentum updates in group order, i.e. the result of ``groups`` sequential calls.  An optional
    residual add and ReLU are applied to the output.  On a HIP device in training mode this is one fused op
    (ops.grouped_batch_norm); on the host, or in eval mode, the ATen graph."""
    groups = 1
    count_deferred = False     # Depth_Model.forward_frames bumps every layer's num_batches_tracked in one foreach call

    def forward(self, x, residual=None, relu=False):
        if x.is_cuda and self.training:
            if self.momentum is None or not self.track_running_stats or not self.affine:
                raise ValueError("FrameBatchNorm2d: only affine, momentum-tracked batch norm is implemented in HIP")
            if not self.count_deferred:
                self.num_batches_tracked.add_(self.groups)
            return ops.grouped_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.groups,
                                          self.eps, self.momentum, residual=residual, relu=relu)
        if self.training and self.groups > 1:
            y = torch.cat([super(FrameBatchNorm2d, self).forward(c) for c in x.chunk(self.groups, 0)], 0)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class StemMaxPool(nn.MaxPool2d):
    """nn.MaxPool2d(3, 2, 1); on a HIP device the 1-byte-index kernels of ops.maxpool3x3s2 (same values and gradients
    bit for bit), on the host the ATen operator."""

    def forward(self, x):
        if x.is_cuda and (self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode) == (3, 2, 1, 1, False) \
                and x.dim() == 4 and x.shape[0] * x.shape[1] <= 65535:
            return ops.maxpool3x3s2(x)
        return super().forward(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = FrameBatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = FrameBatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(out), residual=idt, relu=True)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = FrameBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = StemMaxPool(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 FrameBatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def resnet18(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("no network access: pretrained ImageNet weights are unavailable")
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("no network access: pretrained ImageNet weights are unavailable")
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)
