"""Six-level conv feature pyramid of the PWC flow net (reference feature_pyramid.py:7-36):
channels 16/32/64/96/128/196 at strides 2..64."""
import torch.nn as nn

from ..structures.net_utils import conv


class FeaturePyramid(nn.Module):
    def __init__(self):
        super().__init__()
        widths = [3, 16, 32, 64, 96, 128, 196]
        for lvl in range(6):
            setattr(self, "conv%d" % (2 * lvl + 1), conv(widths[lvl], widths[lvl + 1], kernel_size=3, stride=2))
            setattr(self, "conv%d" % (2 * lvl + 2), conv(widths[lvl + 1], widths[lvl + 1], kernel_size=3, stride=1))

    def forward(self, img):
        feats, x = [], img
        for lvl in range(6):
            x = getattr(self, "conv%d" % (2 * lvl + 2))(getattr(self, "conv%d" % (2 * lvl + 1))(x))
            feats.append(x)
        return tuple(feats)
