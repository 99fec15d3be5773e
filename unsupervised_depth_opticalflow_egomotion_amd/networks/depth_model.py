"""DepthNet: monodepth2-style ResNet-18 encoder + skip decoder (reference depth_model.py:60-211).
Sigmoid disparities at ``depth_scale`` scales, finest first."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import resnet


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers=18, pretrained=False):
        super().__init__()
        if num_layers not in (18, 34):
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = {18: resnet.resnet18, 34: resnet.resnet34}[num_layers](pretrained)

    def forward(self, image):
        e = self.encoder
        x = (image - 0.45) / 0.225
        f0 = e.bn1(e.conv1(x), relu=True)
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


class Conv3x3(nn.Module):
    """Reflection-pad + 3x3 conv (key: ``.conv``)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1)
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """Conv3x3 + ELU (key: ``.conv.conv``)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv3x3(cin, cout)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4)):
        super().__init__()
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.upconvs = nn.ModuleList()
        for i in range(4, -1, -1):
            cin = num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            pair = nn.ModuleList([ConvBlock(cin, self.num_ch_dec[i])])
            cin2 = self.num_ch_dec[i] + (num_ch_enc[i - 1] if i > 0 else 0)
            pair.append(ConvBlock(cin2, self.num_ch_dec[i]))
            self.upconvs.append(pair)
        self.dispconvs = nn.ModuleList([Conv3x3(self.num_ch_dec[s], 1) for s in self.scales])
        self.sigmoid = nn.Sigmoid()

    def forward(self, feats):
        return self.forward_fused(feats) if feats[-1].is_cuda else self.forward_aten(feats)

    def forward_aten(self, feats):
        """The module graph as written (host tensors: the CPU baseline and the CPU-side tests)."""
        out = {}
        x = feats[-1]
        for scale in range(4, -1, -1):
            blk = self.upconvs[4 - scale]
            x = blk[0](x)
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            if scale > 0:
                x = torch.cat([x, feats[scale - 1]], 1)
            x = blk[1](x)
            if scale in self.scales:
                out[scale] = self.sigmoid(self.dispconvs[self.scales.index(scale)](x))
        return out

    def forward_fused(self, feats):
        """Same graph on the GPU: the bias / ELU / bilinear x2 / cat / reflection-pad glue between the MIOpen
        convolutions runs as two fused HIP passes (ops.elu_pad, ops.elu_up2_cat_pad); the convolutions are applied
        bias-free to the padded tensors.  ``a`` / ``c`` hold convolution outputs *before* bias and ELU."""
        from .. import ops
        out = {}
        p = ops.elu_pad(feats[-1], None, apply_elu=False)        # encoder output: already activated
        for scale in range(4, -1, -1):
            c0, c1 = self.upconvs[4 - scale][0].conv.conv, self.upconvs[4 - scale][1].conv.conv
            a = ops.conv3x3_valid(p, c0.weight)
            q = ops.elu_up2_cat_pad(a, c0.bias, feats[scale - 1] if scale > 0 else None)
            c = ops.conv3x3_valid(q, c1.weight)
            if scale in self.scales or scale > 0:
                p = ops.elu_pad(c, c1.bias, apply_elu=True)      # shared by the disparity head and the next stage
            if scale in self.scales:
                head = self.dispconvs[self.scales.index(scale)].conv
                out[scale] = ops.disp_head(p, head.weight, head.bias)      # Conv3x3(C -> 1) + bias + sigmoid
        return out


class Depth_Model(nn.Module):
    def __init__(self, depth_scale, num_layers=18):
        super().__init__()
        self.depth_scale = depth_scale
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=False)
        self.decoder = DepthDecoder(self.encoder.num_ch_enc, scales=range(depth_scale))

    def forward(self, img):
        out = self.decoder(self.encoder(img))
        return [out[i] for i in range(self.depth_scale)]

    def forward_frames(self, frames, batched=None):
        """The frames of a triplet in ONE pass over a batch of len(frames)*B: equivalent to calling the net once per
        frame, in order (model_geometry.py:786-788) -- every BatchNorm normalises each frame's B samples separately and
        updates its running statistics frame by frame (resnet.FrameBatchNorm2d) -- with a third of the launches and
        larger convolutions.  ``batched``: the frames already stacked along the batch ([n*B,3,H,W]), when the caller has
        them in that layout.  Returns one disparity list per frame."""
        n, B = len(frames), frames[0].shape[0]
        bns = self.__dict__.get("_frame_bns")      # (walking ~140 modules per step cost 0.2 ms of host time: cached; the set of
        if bns is None:                            # BatchNorm layers of a built network does not change)
            bns = self.__dict__["_frame_bns"] = [m for m in self.modules() if isinstance(m, resnet.FrameBatchNorm2d)]
        defer = self.training and frames[0].is_cuda
        for m in bns:
            m.groups = n
            m.count_deferred = defer
        try:
            if defer:     # one multi-tensor kernel instead of one tiny add per layer
                torch._foreach_add_([m.num_batches_tracked for m in bns], n)
            out = self.forward(batched if batched is not None else torch.cat(list(frames), 0))   # batched: the frames already stacked along the batch
        finally:
            for m in bns:
                m.groups = 1
                m.count_deferred = False
        parts = [o.split(B) for o in out]          # one concatenation in the backward pass instead of n zero-filled slices
        return [[q[i] for q in parts] for i in range(n)]
