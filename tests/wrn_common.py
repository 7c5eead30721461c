"""Shared inputs of the WiderResNet-38 tests (seeds as in tests/golden/make_golden_wrn.py)."""
import json
import os

import numpy as np
import torch

from mrfp_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GW = np.load(os.path.join(HERE, "golden", "wrn38.npz"))
SPEC = json.load(open(os.path.join(HERE, "golden", "state_dict_spec.json")))


def drop_masks(B, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, C, p in (("mod6.block1", 1024, 0.3), ("mod7.block1", 2048, 0.5)):
        keep = (torch.rand(B, C, 1, 1, generator=g) >= p).float()
        out[name] = keep / (1.0 - p)
    return out


def trunk_case():
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["WiderResNetA2_38"]], seed=0)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 64, 64, generator=g) * 255.0
    gy = torch.randn(2, 4096, 8, 8, generator=g)
    return sd, x, gy, drop_masks(2, 12)


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).sum().sqrt().item()])
