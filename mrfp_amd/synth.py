"""Deterministic synthetic weights and synthetic batches.

The reference fetches ImageNet weights at construction (reference Resnet.py:647-660), which
is impossible offline.  Every tensor of a state dict is instead drawn from a CPU generator
seeded by a hash of its *key*, so that the build container, the GPU box, the oracle and the
HIP model all see bit-identical fp32 weights without shipping a checkpoint.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    g = _gen(key, seed)
    shape = tuple(shape)
    hrfp = key.startswith("OC")
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("running_var"):
        return torch.rand(shape, generator=g) + 0.5
    if key.endswith(".weight") and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    if key.endswith(".weight"):
        if hrfp:                                     # reference mynn.py:57-74: N(0, 0.5)
            return torch.randn(shape, generator=g) * 0.5
        return torch.rand(shape, generator=g) + 0.5
    if key.endswith(".bias"):
        if hrfp:
            return torch.zeros(shape)
        return torch.randn(shape, generator=g) * 0.1
    return torch.randn(shape, generator=g) * 0.1


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
                     residual_gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """spec: iterable of (key, shape) in state_dict order.

    residual_gain scales the weight of the LAST BatchNorm of every bottleneck (`*.bn3.weight`).  With the default 1.0
    every residual branch is as strong as its skip path and the trunk amplifies fp32 rounding noise 3-4x per stage
    (the reference's own fp32 logits are then 0.8e-3 .. 1.5e-3 away from an fp64 evaluation, at any batch size:
    tests/golden/make_golden_wc.py prints it); 0.3 is the regime of a trained / zero-init-residual network (noise
    5e-5 .. 8e-5) and is what the well-conditioned fixtures (mrfp_wc.npz, r101.npz) use."""
    sd = OrderedDict((k, synth_tensor(k, tuple(s), seed)) for k, s in spec)
    if residual_gain != 1.0:
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = sd[k] * residual_gain
    return sd


def spec_of(state_dict: Dict[str, torch.Tensor]):
    return [(k, tuple(v.shape)) for k, v in state_dict.items()]


def synth_batch(batch: int, height: int, width: int, seed: int = 1, num_classes: int = 19,
                ignore_frac: float = 0.03):
    """x = rand*255 fp32 (range of the reference's ToTensor without /255, reference
    dataloaders.py:128-133); labels uniform over the classes with ~3% set to 255 (ignore)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.rand(batch, 3, height, width, generator=g) * 255.0
    y = torch.randint(0, num_classes, (batch, height, width), generator=g)
    ign = torch.rand(batch, height, width, generator=g) < ignore_frac
    y[ign] = 255
    return x, y


def synth_noise(batch: int, seed: int = 2, channels=(64, 256)):
    """The two NP+ normal-draw pairs of one forward: alpha ~ N(1, .75), beta_noise ~ N(0, .75)
    (reference deepv3.py:274-275) for the 64- and the 256-channel call sites (128 / 256 with a deep-stem trunk)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = {}
    for name, c in (("np1", channels[0]), ("np2", channels[1])):
        out[name + "_alpha"] = 1.0 + 0.75 * torch.randn(batch, c, 1, 1, generator=g)
        out[name + "_beta"] = 0.75 * torch.randn(batch, c, 1, 1, generator=g)
    return out
