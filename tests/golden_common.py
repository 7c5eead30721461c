"""Shared inputs of the tests that use tests/golden/mrfp_wc.npz (reference MRFPPlus, ResNet-50, well-conditioned weights,
4x192x192; tests/golden/make_golden_wc.py) and tests/golden/r101.npz (reference ResNet3X3-101 trunk + the MRFP+
composition run by the reference's own forward; tests/golden/make_golden_r101.py).  Seeds as in the generators."""
import json
import os

import numpy as np
import torch

from mrfp_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GWC = np.load(os.path.join(HERE, "golden", "mrfp_wc.npz"))
GR = np.load(os.path.join(HERE, "golden", "r101.npz"))
SPEC = json.load(open(os.path.join(HERE, "golden", "state_dict_spec.json")))
TAGS = {"ttt": (True, True, True), "fff": (False, False, False), "tft": (True, False, True), "ftf": (False, True, False)}
CROP = (slice(None), slice(None), slice(80, 88), slice(40, 48))
GAIN = 0.3


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).sum().sqrt().item()])


def relerr(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def wc_case():
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["MRFPPlus"]], seed=0, residual_gain=GAIN)
    x, y = synth.synth_batch(int(GWC["B"]), int(GWC["S"]), int(GWC["S"]), seed=41)
    return sd, x, y, synth.synth_noise(int(GWC["B"]), seed=42)


def r101_trunk_case():
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["ResNet3X3_101"]], seed=0, residual_gain=GAIN)
    g = torch.Generator().manual_seed(51)
    x = torch.rand(2, 3, 128, 128, generator=g) * 255.0
    gy = torch.randn(2, 2048, 4, 4, generator=g)
    return sd, x, gy


def r101_comp_case():
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["MRFPPlus_r101"]], seed=0, residual_gain=GAIN)
    B, S = int(GR["comp_B"]), int(GR["comp_S"])
    x, y = synth.synth_batch(B, S, S, seed=61)
    return sd, x, y, synth.synth_noise(B, seed=62, channels=(128, 256))


# reference ResNet3X3 attribute names <-> the `layer0.N` keys MRFPPlus gives the deep stem
STEM_ATTR = {"layer0.0": "conv1", "layer0.1": "bn1", "layer0.3": "conv2", "layer0.4": "bn2", "layer0.6": "conv3",
             "layer0.7": "bn3"}


def trunk_key(k):
    """build key (`layer0.N...`) -> key of a stand-alone ResNet3X3 module (`convN` / `bnN`)."""
    for a, b in STEM_ATTR.items():
        if k.startswith(a + "."):
            return b + k[len(a):]
    return k
