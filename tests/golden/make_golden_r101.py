#!/usr/bin/env python3
"""Golden vectors for the ResNet-101 (deep-stem `ResNet3X3`) trunk and for the MRFP+ network built on it -- runs ONLY
in the build container (needs /root/reference).  This is the network bench.py times (BASELINE.json configs[2]/[3]).

Part 1 -- the trunk, pinned by the reference class itself:
  `network.Resnet.resnet101(pretrained=False, wt_layer=[0,0,4,4,4,0,0])` (reference Resnet.py:338-512, 678-693:
  `ResNet3X3(Bottleneck, [3,4,23,3])`, stem norms BN / BN / InstanceNorm(128, affine), InstanceNorm taps on the last
  block of layer1 / layer2).  `ResNet3X3.forward` is run in train mode (forward + backward) and in eval mode and
  compared with `oracle.mrfp_oracle.resnet_trunk` on the same numbers.

Part 2 -- the MRFP+ composition, executed by the REFERENCE'S OWN `MRFPPlus.forward` (reference deepv3.py:280-367):
  the reference's constructor refuses any trunk but 'resnet-50' (deepv3.py:174-178), so the composition itself is
  build-defined -- but nothing in `forward` depends on the trunk except through `layer0[0..3]`, `layer1..4` and the
  channel count of the stem output.  A reference `MRFPPlus` is therefore built the normal way and re-fitted with
  reference parts only: layer0 = Sequential(Sequential(conv1,bn1,relu1,conv2,bn2,relu2,conv3), bn3, relu3, maxpool) of
  the reference's resnet101 (so `layer0[0..3]` walk the deep stem), layer1..4 of the same (with the D16 surgery of
  deepv3.py:184-189), and the two HRFP ends widened to the 128-channel stem (`OClayer1` 128->64, `OCdeclayer4` 64->128,
  `OC4_decbn` 128; plain nn.Conv2d / nn.BatchNorm2d as deepv3.py:221-237 builds them).  The reference forward then
  runs unmodified with injected randomness.  What stays "build-defined" is only that wiring choice; every arithmetic
  step of the fixture is the reference's code.  The oracle (keys as mrfp_amd.deepv3.MRFPPlus(trunk='resnet-101')
  names them) must reproduce it.

Weights: per-key synthetic, residual_gain 0.3 (see mrfp_amd/synth.py: with 33 bottlenecks the plain synthetic weights
amplify fp32 rounding noise beyond 1e-3 in the reference itself).  Writes tests/golden/r101.npz + the key/shape specs.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from mrfp_amd import synth  # noqa: E402
from oracle import mrfp_oracle as orc  # noqa: E402

GAIN = 0.3
WT = [0, 0, 4, 4, 4, 0, 0]
# reference ResNet3X3 attribute -> MRFPPlus `layer0.N` key (mrfp_amd/deepv3.py builds layer0 = Sequential(conv1, bn1, relu1,
# conv2, bn2, relu2, conv3, bn3, relu3, maxpool))
STEM_MAP = {"conv1": "layer0.0", "bn1": "layer0.1", "conv2": "layer0.3", "bn2": "layer0.4", "conv3": "layer0.6",
            "bn3": "layer0.7"}
# composition: reference-side key prefix (layer0 = Sequential(Sequential(...7), bn3, relu3, maxpool)) -> build key prefix
COMP_MAP = {"layer0.0.0.": "layer0.0.", "layer0.0.1.": "layer0.1.", "layer0.0.3.": "layer0.3.", "layer0.0.4.": "layer0.4.",
            "layer0.0.6.": "layer0.6.", "layer0.1.": "layer0.7."}
TAGS = (("ttt", (True, True, True)), ("fff", (False, False, False)), ("tft", (True, False, True)),
        ("ftf", (False, True, False)))
GRAD_KEYS = ["layer0.0.weight", "layer0.3.weight", "layer0.6.weight", "layer0.7.weight", "layer1.0.conv1.weight",
             "layer1.2.instance_norm_layer.bias", "layer2.3.conv2.weight", "layer3.0.downsample.0.weight",
             "layer3.11.conv2.weight", "layer3.22.bn3.weight", "layer4.2.conv2.weight", "aspp.features.2.0.weight",
             "aspp.img_conv.0.weight", "bot_fine.0.weight", "bot_aspp.1.bias", "final1.0.weight", "final1.4.weight",
             "final2.0.weight", "final2.0.bias"]


def l2(t):
    return t.detach().double().pow(2).sum().sqrt().item()


def to_build_key(k):
    for a, b in COMP_MAP.items():
        if k.startswith(a):
            return b + k[len(a):]
    return k


def trunk_part(ref_resnet, out):
    m = ref_resnet.resnet101(pretrained=False, wt_layer=WT)
    ref_sd = m.state_dict()

    def bk(k):                      # reference trunk key -> build key
        head = k.split(".")[0]
        return STEM_MAP[head] + k[len(head):] if head in STEM_MAP else k
    spec = [(bk(k), tuple(v.shape)) for k, v in ref_sd.items() if not k.startswith("fc.")]
    sd = synth.synth_state_dict(spec, seed=0, residual_gain=GAIN)
    load = {k: (sd[bk(k)] if not k.startswith("fc.") else v) for k, v in ref_sd.items()}
    m.load_state_dict(load)
    B, S = 2, 128
    g = torch.Generator().manual_seed(51)
    x = torch.rand(B, 3, S, S, generator=g) * 255.0
    gy = torch.randn(B, 2048, S // 32, S // 32, generator=g)
    m.train()
    y = m(x)
    (y * gy).sum().backward()
    gref = {bk(n): p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    after = {bk(k): v.clone() for k, v in m.state_dict().items() if "running" in k}
    m.load_state_dict(load)
    m.eval()
    with torch.no_grad():
        y_eval = m(x)

    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    ns, taps = {}, {}
    yo = orc.resnet_trunk(work, x, True, new_stats=ns, taps=taps)
    (yo * gy).sum().backward()
    worst = max(((leaf[k].grad.double() - gr.double()).norm() / gr.double().norm()).item() for k, gr in gref.items()
                if gr.double().norm() > 1e-12)
    print("[trunk] train out rel %.1e  worst grad rel-L2 %.1e" % (mg.rel(yo, y), worst))
    assert mg.rel(yo, y) < 2e-6 and worst < 1e-4
    for k, v in after.items():
        assert mg.rel(ns[k], v) < 1e-6, k
    with torch.no_grad():
        yeo = orc.resnet_trunk({k: v.clone() for k, v in sd.items()}, x, False)
    print("[trunk] eval out rel %.1e" % mg.rel(yeo, y_eval))
    assert mg.rel(yeo, y_eval) < 2e-6

    out["trunk_out_stats"] = mg.stats(y)
    out["trunk_out_crop"] = y.detach()[:, 200:208].numpy()
    out["trunk_eval_stats"] = mg.stats(y_eval)
    out["trunk_eval_crop"] = y_eval[:, 200:208].numpy()
    for name, t in taps.items():
        out["trunk_tap/" + name] = mg.stats(t)
    gk = sorted(gref)
    out["trunk_grad_keys"] = np.array(gk)
    out["trunk_grad_l2"] = np.array([l2(gref[k]) for k in gk])
    for k in ("layer0.1.running_mean", "layer0.4.running_var", "layer3.22.bn3.running_var", "layer4.2.bn2.running_mean"):
        out["trunk_running/" + k] = after[k][:8].numpy()
    return spec


def composition_part(ref, ref_resnet, out):
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    model = ref.MRFPPlus(num_classes=19, criterion=crit)
    r101 = ref_resnet.resnet101(pretrained=False, wt_layer=WT)
    model.layer0 = nn.Sequential(nn.Sequential(r101.conv1, r101.bn1, r101.relu1, r101.conv2, r101.bn2, r101.relu2,
                                               r101.conv3), r101.bn3, r101.relu3, r101.maxpool)
    model.layer1, model.layer2, model.layer3, model.layer4 = r101.layer1, r101.layer2, r101.layer3, r101.layer4
    for n, m in model.layer4.named_modules():          # the D16 surgery MRFPPlus.__init__ applies (deepv3.py:184-189)
        if "conv2" in n:
            m.dilation, m.padding, m.stride = (2, 2), (2, 2), (1, 1)
        elif "downsample.0" in n:
            m.stride = (1, 1)
    model.OClayer1 = nn.Conv2d(128, 64, kernel_size=3, stride=1, padding=1).requires_grad_(False)
    model.OCdeclayer4 = nn.Conv2d(64, 128, kernel_size=3, stride=1, padding=2, dilation=2).requires_grad_(False)
    model.OC4_decbn = nn.BatchNorm2d(128).requires_grad_(False)

    from mrfp_amd import deepv3
    with contextlib.redirect_stdout(io.StringIO()):
        mine = deepv3.MRFPPlus(19, trunk="resnet-101")
    spec = synth.spec_of(mine.state_dict())
    ref_keys = {to_build_key(k): k for k in model.state_dict()}
    assert set(ref_keys) == set(k for k, _ in spec), set(ref_keys) ^ set(k for k, _ in spec)
    for k, s in spec:
        assert tuple(model.state_dict()[ref_keys[k]].shape) == tuple(s), k
    assert sorted(to_build_key(n) for n, p in model.named_parameters() if p.requires_grad) == \
        sorted(n for n, p in mine.named_parameters() if p.requires_grad)
    sd0 = synth.synth_state_dict(spec, seed=0, residual_gain=GAIN)
    load = {ref_keys[k]: v for k, v in sd0.items()}

    B, S = 3, 192
    x, y = synth.synth_batch(B, S, S, seed=61)
    noise = synth.synth_noise(B, seed=62, channels=(128, 256))
    out["comp_B"], out["comp_S"] = np.int64(B), np.int64(S)
    keys = orc.trainable_keys(sd0)
    crop = (slice(None), slice(None), slice(80, 88), slice(40, 48))
    for tag, tg in TAGS:
        model.load_state_dict(load)
        model.train()
        model.zero_grad()
        cap = {}
        model.criterion = mg.CaptureCE(cap)
        with mg.Injector(ref, tg, noise):
            loss_ref = model(x, y, training=True)
        loss_ref.backward()
        gref = {to_build_key(n): p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        after = {to_build_key(k): v.clone() for k, v in model.state_dict().items()}

        leaf = {k: sd0[k].clone().requires_grad_(True) for k in keys}
        work = {k: v.clone() for k, v in sd0.items()}
        work.update(leaf)
        taps, ns = {}, {}
        loss_o = orc.mrfp_forward(work, x, y, training=True, toggles=tg, noise=noise, new_stats=ns, taps=taps)
        go = dict(zip(keys, torch.autograd.grad(loss_o, [leaf[k] for k in keys])))
        assert set(go) == set(gref)
        r_loss = abs(loss_o.item() - loss_ref.item()) / abs(loss_ref.item())
        r_log = mg.rel(taps["logits"], cap["logits"])
        worst = max(((go[k].double() - gref[k].double()).norm() / gref[k].double().norm()).item() for k in keys
                    if gref[k].double().norm() > 1e-9)
        assert r_loss < 2e-6 and r_log < 2e-5 and worst < 1e-3, (tag, r_loss, r_log, worst)
        for k, v in ns.items():
            assert mg.rel(v, after[k]) < 1e-5, k

        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        leaf64 = {k: sd64[k].clone().requires_grad_(True) for k in keys}
        sd64.update(leaf64)
        taps64 = {}
        loss64 = orc.mrfp_forward(sd64, x.double(), y, training=True, toggles=tg,
                                  noise={k: v.double() for k, v in noise.items()}, taps=taps64)
        g64 = dict(zip(keys, torch.autograd.grad(loss64, [leaf64[k] for k in keys])))
        lnoise = mg.rel(cap["logits"], taps64["logits"])
        print(f"[comp {tag}] loss {loss_ref.item():.6f} oracle rel {r_loss:.1e}; logits rel {r_log:.1e}; worst grad "
              f"{worst:.1e}; reference fp32-vs-fp64 logits {lnoise:.2e}")
        out[f"comp_{tag}_loss"] = np.float64(loss_ref.item())
        out[f"comp_{tag}_logits_noise"] = np.float64(lnoise)
        out[f"comp_{tag}_logits_stats"] = mg.stats(cap["logits"])
        out[f"comp_{tag}_logits_crop"] = cap["logits"].detach()[crop].numpy()
        for name, t in taps.items():
            if name != "logits":
                out[f"comp_{tag}_tap/{name}"] = mg.stats(t)
        out[f"comp_{tag}_grad_l2"] = np.array([l2(gref[k]) for k in keys])
        out[f"comp_{tag}_grad_self_noise"] = np.array(
            [((gref[k].double() - g64[k]).norm() / g64[k].norm().clamp_min(1e-30)).item() for k in keys])
        for k in GRAD_KEYS:
            out[f"comp_{tag}_grad_head/{k}"] = gref[k].flatten()[:8].numpy()
        if tag == "ttt":
            for k in ("layer0.1.running_mean", "layer3.22.bn3.running_var", "OC4_decbn.running_mean",
                      "aspp.img_conv.1.running_var"):
                out["comp_ttt_running/" + k] = after[k][:8].numpy()
    out["comp_grad_keys"] = np.array(keys)

    # eval path (model.eval(), training=False)
    model.criterion = crit
    model.load_state_dict(load)
    model.eval()
    with torch.no_grad(), mg.Injector(ref, (True, True, True), None):
        logits_ref = model(x, training=False)
    logits_o = orc.mrfp_forward({k: v.clone() for k, v in sd0.items()}, x, training=False, bn_train=False)
    assert mg.rel(logits_o, logits_ref) < 2e-6
    out["comp_eval_logits_stats"] = mg.stats(logits_ref)
    out["comp_eval_logits_crop"] = logits_ref[crop].numpy()
    hist = orc.eval_hist({k: v.clone() for k, v in sd0.items()}, x, y)
    out["comp_eval_hist"] = hist.astype(np.int64)
    out["comp_eval_miou"] = np.float64(orc.miou_from_hist(hist)[0])
    print("[comp eval] logits rel %.1e  mIoU %.5f" % (mg.rel(logits_o, logits_ref), out["comp_eval_miou"]))
    return spec


def main():
    torch.set_num_threads(8)
    ref = mg.import_reference()
    from network import Resnet as ref_resnet
    out = {"gain": np.float64(GAIN)}
    tspec = trunk_part(ref_resnet, out)
    cspec = composition_part(ref, ref_resnet, out)
    np.savez_compressed(os.path.join(HERE, "r101.npz"), **out)
    spec_path = os.path.join(HERE, "state_dict_spec.json")
    specs = json.load(open(spec_path))
    specs["ResNet3X3_101"] = [[k, list(s)] for k, s in tspec]
    specs["MRFPPlus_r101"] = [[k, list(s)] for k, s in cspec]
    json.dump(specs, open(spec_path, "w"))
    print("wrote r101.npz and the specs")


if __name__ == "__main__":
    main()
