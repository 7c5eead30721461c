#!/usr/bin/env python3
"""Well-conditioned golden vectors for the reference's MRFPPlus (ResNet-50) -- runs ONLY in the build container
(needs /root/reference, imported exactly as tests/golden/make_golden.py does).

Why a second fixture: with the plain synthetic weights of mrfp_c1.npz every residual branch is as strong as its skip
path, and the reference's own fp32 logits are 0.8e-3 .. 1.5e-3 (max-norm) away from an fp64 evaluation of the same
graph -- at batch 2, 4 or 6 alike (printed below) -- so no independent fp32 implementation can be held to the
north_star's 1e-3 there with NP+ on.  Here the last BatchNorm weight of every bottleneck is scaled by 0.3
(`synth.synth_state_dict(residual_gain=0.3)`, the regime of a trained network): the reference's fp32-vs-fp64 noise
drops to < 1e-4, and tests/test_model_gpu.py asserts a PLAIN `< 1e-3` on loss and logits for all four toggle sets
(NP+ on in two of them), plus every per-stage statistic, so a drift can be attributed to a stage.

Writes tests/golden/mrfp_wc.npz (numbers only; inputs / weights / noise are re-derivable from seeds).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (reference import + injection helpers)
from mrfp_amd import synth  # noqa: E402
from oracle import mrfp_oracle as orc  # noqa: E402

B, S, GAIN = 4, 192, 0.3
SEED_X, SEED_N = 41, 42
CROP = (slice(None), slice(None), slice(80, 88), slice(40, 48))
TAGS = (("ttt", (True, True, True)), ("fff", (False, False, False)), ("tft", (True, False, True)),
        ("ftf", (False, True, False)))


def l2(t):
    return t.detach().double().pow(2).sum().sqrt().item()


def main():
    torch.set_num_threads(8)
    ref = mg.import_reference()
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    model = ref.MRFPPlus(num_classes=19, criterion=crit)
    spec = synth.spec_of(model.state_dict())
    sd0 = synth.synth_state_dict(spec, seed=0, residual_gain=GAIN)
    x, y = synth.synth_batch(B, S, S, seed=SEED_X)
    noise = synth.synth_noise(B, seed=SEED_N)
    out = {"B": np.int64(B), "S": np.int64(S), "gain": np.float64(GAIN)}
    keys = orc.trainable_keys(sd0)

    for tag, tg in TAGS:
        model.load_state_dict(sd0)
        model.train()
        model.zero_grad()
        cap = {}
        model.criterion = mg.CaptureCE(cap)
        with mg.Injector(ref, tg, noise):
            loss_ref = model(x, y, training=True)
        loss_ref.backward()
        gref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        ref_after = {k: v.clone() for k, v in model.state_dict().items()}

        leaf = {k: sd0[k].clone().requires_grad_(True) for k in keys}
        work = {k: v.clone() for k, v in sd0.items()}
        work.update(leaf)
        taps, new_stats = {}, {}
        loss_o = orc.mrfp_forward(work, x, y, training=True, toggles=tg, noise=noise, new_stats=new_stats, taps=taps)
        grads = torch.autograd.grad(loss_o, [leaf[k] for k in keys])
        go = dict(zip(keys, grads))
        r_loss = abs(loss_o.item() - loss_ref.item()) / abs(loss_ref.item())
        r_log = mg.rel(taps["logits"], cap["logits"])
        worst = max(((go[k].double() - gref[k].double()).norm() / gref[k].double().norm().clamp_min(1e-30)).item()
                    for k in keys if gref[k].double().norm() > 1e-9)
        assert r_loss < 2e-6 and r_log < 2e-5 and worst < 1e-3, (tag, r_loss, r_log, worst)
        for k, v in new_stats.items():
            assert mg.rel(v, ref_after[k]) < 1e-5, k

        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        leaf64 = {k: sd64[k].clone().requires_grad_(True) for k in keys}
        sd64.update(leaf64)
        taps64 = {}
        loss64 = orc.mrfp_forward(sd64, x.double(), y, training=True, toggles=tg,
                                  noise={k: v.double() for k, v in noise.items()}, taps=taps64)
        g64 = dict(zip(keys, torch.autograd.grad(loss64, [leaf64[k] for k in keys])))
        lnoise = mg.rel(cap["logits"], taps64["logits"])
        gnoise = max(((gref[k].double() - g64[k]).norm() / g64[k].norm()).item() for k in keys if g64[k].norm() > 1e-9)
        print(f"[{tag}] loss {loss_ref.item():.6f} oracle rel {r_loss:.1e}; logits rel {r_log:.1e}; worst grad {worst:.1e}; "
              f"reference fp32-vs-fp64: logits {lnoise:.2e}, worst gradient rel-L2 {gnoise:.2e}")
        assert lnoise < 4e-4, lnoise          # the point of this fixture

        out[f"{tag}_loss"] = np.float64(loss_ref.item())
        out[f"{tag}_loss64"] = np.float64(loss64.item())
        out[f"{tag}_logits_noise"] = np.float64(lnoise)
        out[f"{tag}_grad_noise"] = np.float64(gnoise)
        out[f"{tag}_logits_stats"] = mg.stats(cap["logits"])
        out[f"{tag}_logits_crop"] = cap["logits"].detach()[CROP].numpy()
        for name, t in taps.items():
            if name != "logits":
                out[f"{tag}_tap/{name}"] = mg.stats(t)
        # L2 norm of EVERY trainable gradient as the reference computed it, and of the fp64 evaluation
        out[f"{tag}_grad_l2"] = np.array([l2(gref[k]) for k in keys])
        out[f"{tag}_grad_l2_64"] = np.array([l2(g64[k]) for k in keys])
        out[f"{tag}_grad_self_noise"] = np.array([((gref[k].double() - g64[k]).norm() / g64[k].norm().clamp_min(1e-30)).item()
                                                  for k in keys])
        for k in mg.GRAD_KEYS:
            out[f"{tag}_grad_head/{k}"] = gref[k].flatten()[:8].numpy()
    out["grad_keys"] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, "mrfp_wc.npz"), **out)
    print("wrote mrfp_wc.npz")


if __name__ == "__main__":
    main()
