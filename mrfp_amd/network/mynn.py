"""Norm / conv / resize layers and initialisers behind the reference's `network.mynn` surface
(reference network/mynn.py:19-25, 38-74, 114-138), executing on the HIP kernels of mrfp_amd/csrc.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..config import cfg


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers / state_dict keys) whose forward is the HIP
    statistics + apply pair.  `fused(...)` lets the owning block fold the ReLU, the residual add and
    a preceding nearest resize into the same two passes."""

    # num_batches_tracked is counted on the host and flushed into the buffer when the state dict is read:
    # a 1-element device kernel per BatchNorm per step (122 launches) is pure launch overhead.
    _nbt_pending = 0

    def _flush_nbt(self):
        if self._nbt_pending and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._nbt_pending)
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_nbt()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)

    def fused(self, x, *, relu=False, res=None, plan=None, emit_stats=False):
        training = self.training or (self.running_mean is None)
        if training and self.num_batches_tracked is not None:
            self._nbt_pending += 1
        return ops.batch_norm_act(x, self.weight, self.bias,
                                  self.running_mean if self.track_running_stats else None,
                                  self.running_var if self.track_running_stats else None,
                                  training=training, momentum=self.momentum, eps=self.eps,
                                  relu=relu, res=res, plan=plan, emit_stats=emit_stats)

    def forward(self, x):
        return self.fused(x)


class HipInstanceNorm2d(nn.InstanceNorm2d):
    """nn.InstanceNorm2d (affine or not, no running stats) on the HIP kernels."""

    _emit_plane_stats = False      # set by a caller that runs NP+ on this layer's output next (deepv3.MRFPPlus.forward)

    def fused(self, x, *, relu=False):
        return ops.instance_norm_act(x, self.weight, self.bias, eps=self.eps, relu=relu, emit_stats=self._emit_plane_stats)

    def fused_relu_pool(self, x):
        """ReLU and the 3x3 / stride 2 max pool behind this layer, in one operator (the stem: ops.instance_norm_relu_pool)."""
        return ops.instance_norm_relu_pool(x, self.weight, self.bias, eps=self.eps)

    def forward(self, x):
        return self.fused(x)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (OIHW fp32 master weight = checkpoint ABI) on the HIP implicit-GEMM kernels."""

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)

    def forward_skip(self, x):
        """(conv(x), x_skip): x_skip aliases x; its gradient is folded into this conv's dgrad launch."""
        return ops.conv2d_skip(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


def Norm2d(in_channels):
    """reference mynn.py:19-25: the BN class comes from cfg.MODEL.BNFUNC."""
    layer = cfg.MODEL.BNFUNC or HipBatchNorm2d
    return layer(in_channels)


def Upsample(x, size):
    """reference mynn.py:114-119: bilinear, align_corners=True."""
    return ops.upsample_bilinear(x, size)


def _init(models, bn_weight_std):
    for model in models:
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d, nn.GroupNorm, nn.SyncBatchNorm)):
                if bn_weight_std is None:
                    m.weight.data.fill_(1)
                else:
                    nn.init.normal_(m.weight, mean=0.0, std=bn_weight_std)
                m.bias.data.zero_()


def initialize_weights(*models):
    """reference mynn.py:38-55: Kaiming(fan_in) convs, unit BN."""
    _init(models, None)


def initialize_weights_kaimingnormal_forOC(*models):
    """reference mynn.py:57-74: the HRFP re-initialiser -- Kaiming(fan_in) convs, zero bias,
    BN weight ~ N(0, 0.5), BN bias 0."""
    _init(models, 0.5)


def freeze_weights(*models):
    for model in models:
        for p in model.parameters():
            p.requires_grad = False


def unfreeze_weights(*models):
    for model in models:
        for p in model.parameters():
            p.requires_grad = True


def forgiving_state_restore(net, loaded_dict):
    """reference mynn.py:121-138: load the tensors whose name and size match, keep the rest."""
    own = net.state_dict()
    own.update({k: v for k, v in loaded_dict.items() if k in own and own[k].size() == v.size()})
    net.load_state_dict(own)
    return net
