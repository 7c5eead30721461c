"""WiderResNet-A2 (pre-activation identity-mapping blocks) behind the reference's `network.wider_resnet` surface
(reference network/wider_resnet.py:43-48 bnrelu, 64-181 IdentityResidualBlock, 267-376 WiderResNetA2, 379-395 the
`wider_resnet{16,20,38}_a2` factories), executing on the HIP kernels of mrfp_amd/csrc.

Same module tree and state_dict keys as the reference (`mod1.conv1`, `modK.blockJ.bn1.0`, `.convs.conv1`,
`.convs.bn2.0`, ..., `.proj_conv`, `bn_out.0`), so reference checkpoints load unchanged.  Every `bnrelu` pair runs as
the fused statistics + apply(+ReLU) passes, the convolutions as the MFMA implicit GEMM, Dropout2d (mod6 p = 0.3,
mod7 p = 0.5; reference wider_resnet.py:333-338 with `nn.Dropout = nn.Dropout2d`, line 302) as a per-(image, channel)
scale inside one apply pass.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import mynn
from .. import ops


def bnrelu(channels):
    """reference wider_resnet.py:43-48."""
    return nn.Sequential(mynn.Norm2d(channels), nn.ReLU(inplace=True))


def _bnrelu(seq, x):
    return seq[0].fused(x, relu=True)


class _DropMasks:
    """Source of the Dropout2d keep-masks [B,C] (already divided by 1-p).  Tests inject fixed masks."""

    def __init__(self):
        self.injected = None          # dict name -> Tensor[B,C] or None

    def __call__(self, name, B, C, p, device):
        if self.injected is not None:
            return self.injected[name].to(device=device, dtype=torch.float32).reshape(B, C)
        keep = torch.bernoulli(torch.full((B, C), 1.0 - p, device=device))
        return keep / (1.0 - p)


DROP_MASKS = _DropMasks()


class HipDropout2d(nn.Dropout2d):
    """nn.Dropout2d: whole channels of an image are zeroed with probability p, the rest scaled by 1/(1-p)."""

    mask_name = ""

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        B, C = x.shape[0], x.shape[1]
        return ops.channel_scale(x, DROP_MASKS(self.mask_name, B, C, self.p, x.device))


class GlobalAvgPool2d(nn.Module):
    """reference wider_resnet.py:50-61."""

    def forward(self, inputs):
        return ops.global_avg_pool(inputs).float().flatten(1)


class IdentityResidualBlock(nn.Module):
    """reference wider_resnet.py:64-181: bn1 -> (proj_conv | identity) shortcut; convs(bn1) + shortcut."""

    def __init__(self, in_channels, channels, stride=1, dilation=1, groups=1, norm_act=bnrelu, dropout=None,
                 dist_bn=False):
        super().__init__()
        self.dist_bn = dist_bn
        if len(channels) != 2 and len(channels) != 3:
            raise ValueError("channels must contain either two or three values")
        if len(channels) == 2 and groups != 1:
            raise ValueError("groups > 1 are only valid if len(channels) == 3")
        if groups != 1:
            raise ValueError("grouped convolutions are not on the MRFP hot path (groups=%d)" % groups)
        is_bottleneck = len(channels) == 3
        need_proj_conv = stride != 1 or in_channels != channels[-1]
        self.bn1 = norm_act(in_channels)
        C2 = mynn.HipConv2d
        if not is_bottleneck:
            layers = [("conv1", C2(in_channels, channels[0], 3, stride=stride, padding=dilation, bias=False, dilation=dilation)),
                      ("bn2", norm_act(channels[0])),
                      ("conv2", C2(channels[0], channels[1], 3, stride=1, padding=dilation, bias=False, dilation=dilation))]
            if dropout is not None:
                layers = layers[0:2] + [("dropout", dropout())] + layers[2:]
        else:
            layers = [("conv1", C2(in_channels, channels[0], 1, stride=stride, padding=0, bias=False)),
                      ("bn2", norm_act(channels[0])),
                      ("conv2", C2(channels[0], channels[1], 3, stride=1, padding=dilation, bias=False, dilation=dilation)),
                      ("bn3", norm_act(channels[1])),
                      ("conv3", C2(channels[1], channels[2], 1, stride=1, padding=0, bias=False))]
            if dropout is not None:
                layers = layers[0:4] + [("dropout", dropout())] + layers[4:]
        self.convs = nn.Sequential(OrderedDict(layers))
        if need_proj_conv:
            self.proj_conv = C2(in_channels, channels[-1], 1, stride=stride, padding=0, bias=False)

    def forward(self, x):
        b1 = _bnrelu(self.bn1, x)
        if hasattr(self, "proj_conv"):
            # bn1 feeds conv1 and the projection: chained through conv1's dgrad epilogue (HipConv2d.forward_skip)
            out, b1 = self.convs.conv1.forward_skip(b1)
            shortcut = self.proj_conv(b1)
        else:
            out = self.convs.conv1(b1)
            shortcut = x
        for name, m in list(self.convs.named_children())[1:]:
            if name.startswith("bn"):
                out = _bnrelu(m, out)
            else:
                out = m(out)
        return ops.add(out, shortcut)


class WiderResNetA2(nn.Module):
    """reference wider_resnet.py:267-376 (structure = blocks per module mod2..mod7)."""

    def __init__(self, structure, norm_act=bnrelu, classes=0, dilation=False, dist_bn=False):
        super().__init__()
        self.dist_bn = dist_bn
        norm_act = bnrelu
        self.structure = structure
        self.dilation = dilation
        if len(structure) != 6:
            raise ValueError("Expected a structure with six values")
        self.mod1 = nn.Sequential(OrderedDict([("conv1", mynn.HipConv2d(3, 64, 3, stride=1, padding=1, bias=False))]))
        in_channels = 64
        channels = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
        for mod_id, num in enumerate(structure):
            blocks = []
            for block_id in range(num):
                if not dilation:
                    dil = 1
                    stride = 2 if block_id == 0 and 2 <= mod_id <= 4 else 1
                else:
                    dil = 2 if mod_id == 3 else 4 if mod_id > 3 else 1
                    stride = 2 if block_id == 0 and mod_id == 2 else 1
                drop = None
                if mod_id in (4, 5):
                    drop = partial(self._dropout, "mod%d.block%d" % (mod_id + 2, block_id + 1), 0.3 if mod_id == 4 else 0.5)
                blocks.append(("block%d" % (block_id + 1),
                               IdentityResidualBlock(in_channels, channels[mod_id], norm_act=norm_act, stride=stride,
                                                     dilation=dil, dropout=drop, dist_bn=self.dist_bn)))
                in_channels = channels[mod_id][-1]
            if mod_id < 2:
                self.add_module("pool%d" % (mod_id + 2), nn.MaxPool2d(3, stride=2, padding=1))
            self.add_module("mod%d" % (mod_id + 2), nn.Sequential(OrderedDict(blocks)))
        self.bn_out = norm_act(in_channels)
        if classes != 0:
            self.classifier = nn.Sequential(OrderedDict([("avg_pool", GlobalAvgPool2d()), ("fc", nn.Linear(in_channels, classes))]))

    @staticmethod
    def _dropout(name, p):
        d = HipDropout2d(p=p)
        d.mask_name = name
        return d

    def stem(self, img):
        """mod1 -> pool2 -> mod2 -> pool3: 128 channels at 1/4 resolution (what the MRFP+ composition calls xp)."""
        out = self.mod1(ops.as_activation(img))
        out = self.mod2(ops.max_pool_3x3_s2(out))
        return ops.max_pool_3x3_s2(out)

    def forward(self, img):
        out = self.stem(img)
        out = self.mod7(self.mod6(self.mod5(self.mod4(self.mod3(out)))))
        out = _bnrelu(self.bn_out, out)
        if hasattr(self, "classifier"):
            return self.classifier(out)
        return out


_NETS = {"16": {"structure": [1, 1, 1, 1, 1, 1]}, "20": {"structure": [1, 1, 1, 3, 1, 1]}, "38": {"structure": [3, 3, 6, 3, 1, 1]}}
__all__ = []
for _name, _params in _NETS.items():
    _net = "wider_resnet" + _name + "_a2"
    setattr(sys.modules[__name__], _net, partial(WiderResNetA2, **_params))
    __all__.append(_net)
