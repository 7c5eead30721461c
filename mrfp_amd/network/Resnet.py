"""ResNet trunks with the `wt_layer` whitening / instance-norm taps, same module tree and
state_dict keys as the reference (reference network/Resnet.py:73-227, 338-722), executing on the
HIP kernels: every conv -> HipConv2d, every norm -> fused HIP statistics/apply passes, with ReLU
and the residual add folded into the apply pass of the last norm of each block.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from ..config import cfg
from . import mynn
from .instance_whitening import InstanceWhitening

__all__ = ["ResNet", "ResNet3X3", "BasicBlock", "Bottleneck", "resnet18", "resnet34", "resnet50", "resnet101",
           "resnet152"]

_CKPT_NAMES = {"resnet18": "resnet18-5c106cde.pth", "resnet34": "resnet34-333f7ec4.pth",
               "resnet50": "resnet50-19c8e357.pth", "resnet101": "resnet101-5d3b4d8f.pth",
               "resnet152": "resnet152-b121ed2d.pth"}


def _iw_module(iw: int, channels: int):
    """The module a `wt_layer` code selects (reference Resnet.py:88-112, 166-190, 525-549)."""
    if iw in (1, 2):
        return InstanceWhitening(channels)
    if iw == 3:
        return mynn.HipInstanceNorm2d(channels, affine=False)
    if iw == 4:
        return mynn.HipInstanceNorm2d(channels, affine=True)
    if iw == 5:
        from .sync_switchwhiten import SyncSwitchWhiten2d
        return SyncSwitchWhiten2d(channels, num_pergroup=16, sw_type=2, T=5, tie_weight=False, eps=1e-5,
                                  momentum=0.99, affine=True)
    return None


def _norm_relu(layer, iw, x, w_arr):
    """norm selected by `iw` followed by ReLU; IN / BN fold the ReLU into their apply pass."""
    if iw in (1, 2):
        x, w = layer(x)
        w_arr.append(w)
        return ops.relu(x)
    if isinstance(layer, (mynn.HipBatchNorm2d, mynn.HipInstanceNorm2d)):
        return layer.fused(x, relu=True)
    return ops.relu(layer(x))


def _norm_relu_pool(layer, iw, x, w_arr):
    """_norm_relu followed by the stem's MaxPool2d(3, 2, 1) (reference Resnet.py:549-551); an InstanceNorm there runs as one
    operator with the pool."""
    if iw not in (1, 2) and isinstance(layer, mynn.HipInstanceNorm2d):
        return layer.fused_relu_pool(x)
    return ops.max_pool_3x3_s2(_norm_relu(layer, iw, x, w_arr))


class _Block(nn.Module):
    """Common tail of BasicBlock / Bottleneck: residual add, optional iw layer, ReLU."""

    def _set_iw(self, iw, channels):
        self.iw = iw
        layer = _iw_module(iw, channels)
        if layer is not None:
            self.instance_norm_layer = layer
        self.relu = nn.ReLU(inplace=iw not in (1, 2))

    def _unpack(self, x_tuple):
        if len(x_tuple) != 2:
            print("error!!!")            # reference Resnet.py:196-198
            return None, None
        return x_tuple[0], x_tuple[1]

    def _first_conv(self, x):
        """conv1(x) and the tensor the skip connection should read: an alias of x whose gradient the conv1 dgrad
        kernel accumulates."""
        # (blocks with a downsample branch feed the alias to the downsample conv: its dgrad output becomes the addend of
        #  conv1's dgrad launch instead of a separate accumulation pass over the block input)
        return self.conv1.forward_skip(x)

    def _tail(self, last_bn, out, x, w_arr):
        residual = x if self.downsample is None else self.downsample[1].fused(self.downsample[0](x))
        if self.iw >= 1:
            # (iw 3 / 4: an InstanceNorm reads this output next -- the apply pass hands it the plane sums)
            out = last_bn.fused(out, res=residual, emit_stats=self.iw in (3, 4) and isinstance(last_bn, mynn.HipBatchNorm2d))
            out = _norm_relu(self.instance_norm_layer, self.iw, out, w_arr)
        else:
            out = last_bn.fused(out, res=residual, relu=True)
        return [out, w_arr]


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, iw=0):
        super().__init__()
        self.conv1 = mynn.HipConv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = mynn.Norm2d(planes)
        self.conv2 = mynn.HipConv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = mynn.Norm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self._set_iw(iw, planes * self.expansion)

    def forward(self, x_tuple):
        x, w_arr = self._unpack(x_tuple)
        if x is None:
            return None
        out, skip = self._first_conv(x)
        out = self.bn1.fused(out, relu=True)
        return self._tail(self.bn2, self.conv2(out), skip, w_arr)


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, iw=0):
        super().__init__()
        self.conv1 = mynn.HipConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = mynn.Norm2d(planes)
        self.conv2 = mynn.HipConv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = mynn.Norm2d(planes)
        self.conv3 = mynn.HipConv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = mynn.Norm2d(planes * self.expansion)
        self.downsample = downsample
        self.stride = stride
        self._set_iw(iw, planes * self.expansion)

    def forward(self, x_tuple):
        x, w_arr = self._unpack(x_tuple)
        if x is None:
            return None
        out, skip = self._first_conv(x)
        out = self.bn1.fused(out, relu=True)
        out = self.bn2.fused(self.conv2(out), relu=True)
        return self._tail(self.bn3, self.conv3(out), skip, w_arr)


class _Trunk(nn.Module):
    def _make_layer(self, block, planes, blocks, stride=1, wt_layer=0):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                mynn.HipConv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                mynn.Norm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, iw=0)]
        self.inplanes = planes * block.expansion
        for index in range(1, blocks):   # the iw module sits on the LAST block only (Resnet.py:579-585)
            layers.append(block(self.inplanes, planes, iw=0 if (wt_layer > 0 and index < blocks - 1) else wt_layer))
        return nn.Sequential(*layers)

    def _finish(self, block, layers, wt_layer, num_classes):
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], wt_layer=wt_layer[3])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, wt_layer=wt_layer[4])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, wt_layer=wt_layer[5])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, wt_layer=wt_layer[6])
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self.wt_layer = wt_layer
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _stages(self, x, w_arr):
        from ..conv import wgrad_boundary      # stage boundaries: the queued weight gradients of a stage are issued as grouped launches
        x_tuple = self.layer1([wgrad_boundary(x), w_arr])
        for layer in (self.layer2, self.layer3, self.layer4):
            x_tuple[0] = wgrad_boundary(x_tuple[0])
            x_tuple = layer(x_tuple)
        return x_tuple[0]


def _stem_norm(code, channels):
    layer = _iw_module(code, channels)
    return layer if layer is not None else mynn.Norm2d(channels)


class ResNet(_Trunk):
    """7x7-stem ResNet (reference Resnet.py:514-615); `wt_layer[2]` selects the stem norm."""

    def __init__(self, block, layers, wt_layer=None, num_classes=1000):
        self.inplanes = 64
        super().__init__()
        wt_layer = wt_layer if wt_layer is not None else [0] * 7
        self.conv1 = mynn.HipConv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = _stem_norm(wt_layer[2], 64)
        self.relu = nn.ReLU(inplace=wt_layer[2] not in (1, 2))
        self._finish(block, layers, wt_layer, num_classes)

    def forward(self, x):
        w_arr = []
        x = _norm_relu_pool(self.bn1, self.wt_layer[2], self.conv1(ops.as_activation(x)), w_arr)
        return self._stages(x, w_arr)


class ResNet3X3(_Trunk):
    """Deep-stem (three 3x3 convs) ResNet used for resnet-101 (reference Resnet.py:338-512)."""

    def __init__(self, block, layers, wt_layer=None, num_classes=1000):
        self.inplanes = 128
        super().__init__()
        wt_layer = wt_layer if wt_layer is not None else [0] * 7
        self.conv1 = mynn.HipConv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = _stem_norm(wt_layer[0], 64)
        self.relu1 = nn.ReLU(inplace=wt_layer[0] not in (1, 2))
        self.conv2 = mynn.HipConv2d(64, 64, 3, 1, 1, bias=False)
        self.bn2 = _stem_norm(wt_layer[1], 64)
        self.relu2 = nn.ReLU(inplace=wt_layer[1] not in (1, 2))
        self.conv3 = mynn.HipConv2d(64, 128, 3, 1, 1, bias=False)
        self.bn3 = _stem_norm(wt_layer[2], 128)
        self.relu3 = nn.ReLU(inplace=wt_layer[2] not in (1, 2))
        self._finish(block, layers, wt_layer, num_classes)

    def stem(self, x, w_arr):
        """conv-norm-ReLU x 3 and the max pool."""
        x = _norm_relu(self.bn1, self.wt_layer[0], self.conv1(ops.as_activation(x)), w_arr)
        x = _norm_relu(self.bn2, self.wt_layer[1], self.conv2(x), w_arr)
        return _norm_relu_pool(self.bn3, self.wt_layer[2], self.conv3(x), w_arr)

    def forward(self, x):
        w_arr = []
        return self._stages(self.stem(x, w_arr), w_arr)


def _maybe_pretrained(model, name, pretrained):
    """The reference downloads ImageNet weights here (Resnet.py:647-660).  There is no network in this
    build: a local checkpoint under cfg.MODEL.PRETRAINED_DIR is used when present, otherwise the
    initialiser's weights stay."""
    if not pretrained:
        return model
    print("########### pretrained ##############")
    d = cfg.MODEL.PRETRAINED_DIR or os.environ.get("MRFP_PRETRAINED_DIR")
    path = os.path.join(d, _CKPT_NAMES[name]) if d else None
    if path and os.path.exists(path):
        mynn.forgiving_state_restore(model, torch.load(path, map_location="cpu"))
    else:
        print("[mrfp_amd] no local %s checkpoint (set cfg.MODEL.PRETRAINED_DIR); keeping initialiser weights"
              % _CKPT_NAMES[name])
    return model


def resnet18(pretrained=True, wt_layer=None, **kwargs):
    return _maybe_pretrained(ResNet(BasicBlock, [2, 2, 2, 2], wt_layer=wt_layer or [0] * 7, **kwargs), "resnet18", pretrained)


def resnet34(pretrained=True, wt_layer=None, **kwargs):
    return _maybe_pretrained(ResNet(BasicBlock, [3, 4, 6, 3], wt_layer=wt_layer or [0] * 7, **kwargs), "resnet34", pretrained)


def resnet50(pretrained=True, wt_layer=None, **kwargs):
    return _maybe_pretrained(ResNet(Bottleneck, [3, 4, 6, 3], wt_layer=wt_layer or [0] * 7, **kwargs), "resnet50", pretrained)


def resnet101(pretrained=True, wt_layer=None, **kwargs):
    return _maybe_pretrained(ResNet3X3(Bottleneck, [3, 4, 23, 3], wt_layer=wt_layer or [0] * 7, **kwargs), "resnet101", pretrained)


def resnet152(pretrained=True, wt_layer=None, **kwargs):
    return _maybe_pretrained(ResNet(Bottleneck, [3, 8, 36, 3], wt_layer=wt_layer or [0] * 7, **kwargs), "resnet152", pretrained)
