#!/usr/bin/env python3
"""Golden vectors for the WiderResNet-38-A2 trunk -- runs ONLY in the build container (needs /root/reference).

The reference's `network/wider_resnet.py` imports as-is.  This script builds
`WiderResNetA2([3,3,6,3,1,1], dilation=True)` from it, loads the build's per-key synthetic weights, replaces the two
Dropout2d draws (mod6 p=0.3, mod7 p=0.5) by injected keep-masks, runs forward + backward in train mode and in eval
mode on seeded inputs, asserts that the CPU restatement `oracle.mrfp_oracle.wider_resnet_a2` gives the same numbers,
and writes `wrn38.npz` (outputs only -- inputs and weights are re-derivable from seeds) plus the key/shape spec.
The MRFP+ composition on this trunk is build-defined (no reference behaviour): it is checked against the live oracle.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from mrfp_amd import synth  # noqa: E402
from oracle import mrfp_oracle as orc  # noqa: E402


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).sum().sqrt().item()])


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def drop_masks(B, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, C, p in (("mod6.block1", 1024, 0.3), ("mod7.block1", 2048, 0.5)):
        keep = (torch.rand(B, C, 1, 1, generator=g) >= p).float()
        out[name] = keep / (1.0 - p)
    return out


def main():
    sys.path.insert(0, REF)
    import network.wider_resnet as ref_wrn
    torch.manual_seed(0)
    model = ref_wrn.WiderResNetA2([3, 3, 6, 3, 1, 1], dilation=True)
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = synth.synth_state_dict(spec, seed=0)
    model.load_state_dict(sd)

    B, S = 2, 64
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, S, S, generator=g) * 255.0
    gy = torch.randn(B, 4096, S // 8, S // 8, generator=g)
    masks = drop_masks(B, 12)
    queue = []

    def injected_dropout2d(inp, p=0.5, training=True, inplace=False):
        if not training:
            return inp
        return inp * queue.pop(0)

    orig = F.dropout2d
    F.dropout2d = injected_dropout2d
    torch.nn.functional.dropout2d = injected_dropout2d
    try:
        model.train()
        queue[:] = [masks["mod6.block1"], masks["mod7.block1"]]
        out = model(x)
        loss = (out * gy).sum()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        running = {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k}
        model.zero_grad()
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            out_eval = model(x)
    finally:
        F.dropout2d = orig
        torch.nn.functional.dropout2d = orig

    # the build's restatement on the same numbers
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    new_stats, taps = {}, {}
    out_o = orc.wider_resnet_a2(work, x, True, new_stats=new_stats, drop_masks=masks, taps=taps)
    (out_o * gy).sum().backward()
    print("[train] out rel %.2e" % rel(out_o, out))
    assert rel(out_o, out) < 2e-5
    worst = 0.0
    for k, gref in grads.items():
        n = gref.double().pow(2).sum().sqrt().item()
        if n < 1e-12:
            continue
        e = (leaf[k].grad.double() - gref.double()).pow(2).sum().sqrt().item() / n
        worst = max(worst, e)
    print("[train] worst grad rel-L2 %.2e" % worst)
    assert worst < 5e-4
    for k, v in running.items():
        assert rel(new_stats[k], v) < 1e-5, k
    with torch.no_grad():
        out_eo = orc.wider_resnet_a2({k: v.clone() for k, v in sd.items()}, x, False)
    print("[eval] out rel %.2e" % rel(out_eo, out_eval))
    assert rel(out_eo, out_eval) < 2e-5

    fx = {"out_stats": stats(out), "out_crop": out[:, 100:108, 2:6, 2:6].detach().numpy(),
          "eval_out_stats": stats(out_eval), "eval_out_crop": out_eval[:, 100:108, 2:6, 2:6].numpy(),
          "mod3_stats": stats(taps["mod3"]), "mod5_stats": stats(taps["mod5"])}
    for k in ("mod1.conv1.weight", "mod2.block1.convs.conv1.weight", "mod4.block1.proj_conv.weight",
              "mod5.block2.convs.conv2.weight", "mod6.block1.convs.conv2.weight", "mod7.block1.convs.conv3.weight",
              "bn_out.0.weight", "mod3.block2.bn1.0.bias"):
        fx["grad_l2/" + k] = grads[k].double().pow(2).sum().sqrt().item()
        fx["grad_head/" + k] = grads[k].flatten()[:8].numpy()
    for k in ("mod2.block1.bn1.0.running_mean", "mod7.block1.convs.bn3.0.running_var", "bn_out.0.running_mean"):
        fx["running/" + k] = running[k][:8].numpy()
    np.savez_compressed(os.path.join(HERE, "wrn38.npz"), **fx)

    # key / shape specs (the GPU box rebuilds the weights from these with the per-key seeded synthesiser)
    spec_path = os.path.join(HERE, "state_dict_spec.json")
    specs = json.load(open(spec_path))
    specs["WiderResNetA2_38"] = [[k, list(s)] for k, s in spec]
    import contextlib
    import io
    from mrfp_amd import deepv3
    with contextlib.redirect_stdout(io.StringIO()):
        m = deepv3.MRFPPlus(19, trunk="wider_resnet38_a2")
    specs["MRFPPlus_wrn38"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    json.dump(specs, open(spec_path, "w"))
    print("wrote wrn38.npz and the specs")


if __name__ == "__main__":
    main()
