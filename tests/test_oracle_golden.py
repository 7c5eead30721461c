"""The oracle (oracle/mrfp_oracle.py) against the committed golden vectors.

The vectors in tests/golden/mrfp_c1.npz were produced by the *reference* model itself
(tests/golden/make_golden.py, build container only); this test re-derives the inputs from seeds
and checks that the CPU restatement still reproduces them.  Runs without a GPU.
"""
import json
import os

import numpy as np
import pytest
import torch

from mrfp_amd import synth
from oracle import mrfp_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "mrfp_c1.npz"))
SPEC = json.load(open(os.path.join(HERE, "golden", "state_dict_spec.json")))
CROP = (slice(None), slice(None), slice(100, 108), slice(60, 68))
TAGS = {"ttt": (True, True, True), "fff": (False, False, False),
        "tft": (True, False, True), "ftf": (False, True, False)}


def _stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).sum().sqrt().item()])


@pytest.fixture(scope="module")
def c1():
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["MRFPPlus"]], seed=0)
    x, y = synth.synth_batch(2, 256, 256, seed=1)
    return sd, x, y, synth.synth_noise(2, seed=2)


def test_spec_counts():
    assert len(SPEC["MRFPPlus"]) == 431
    n_train = sum(int(np.prod(s)) for k, s in SPEC["MRFPPlus"] if k in set(SPEC["trainable"]))
    assert n_train == 40353203
    assert sorted(orc.trainable_keys(dict((k, None) for k, _ in SPEC["MRFPPlus"]))) == sorted(SPEC["trainable"])


@pytest.mark.parametrize("tag", ["ttt", "ftf"])
def test_train_forward_backward_matches_reference(c1, tag):
    sd, x, y, noise = c1
    keys = orc.trainable_keys(sd)
    leaf = {k: sd[k].clone().requires_grad_(True) for k in keys}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    taps = {}
    loss = orc.mrfp_forward(work, x, y, training=True, toggles=TAGS[tag], noise=noise, taps=taps)
    assert abs(loss.item() - float(G[f"{tag}_loss"])) / float(G[f"{tag}_loss"]) < 1e-5
    np.testing.assert_allclose(taps["logits"].detach()[CROP].numpy(), G[f"{tag}_logits_crop"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(_stats(taps["logits"]), G[f"{tag}_logits_stats"], rtol=1e-4)
    gk = [k[len(tag) + 9:] for k in G.files if k.startswith(f"{tag}_grad_l2/")]
    grads = torch.autograd.grad(loss, [leaf[k] for k in gk])
    for k, g in zip(gk, grads):
        ref = float(G[f"{tag}_grad_l2/{k}"])
        assert abs(g.double().pow(2).sum().sqrt().item() - ref) / ref < 2e-3, k
    if tag == "ttt":
        for name, t in taps.items():
            np.testing.assert_allclose(_stats(t), G[f"ttt_tap/{name}"], rtol=2e-4, err_msg=name)


def test_eval_hist_and_miou(c1):
    sd, x, y, _ = c1
    hist = orc.eval_hist({k: v.clone() for k, v in sd.items()}, x, y)
    # a handful of argmax ties may flip with a different CPU thread count; the histogram is int
    assert np.abs(hist - G["eval_hist"]).sum() <= 8
    assert abs(orc.miou_from_hist(hist)[0] - float(G["eval_miou"])) < 1e-4


def test_train3_low_lr(c1):
    sd, _, _, _ = c1
    sd = {k: v.clone() for k, v in sd.items()}
    toggles = [(True, True, True), (False, True, False), (True, False, True)]
    batches = [synth.synth_batch(2, 256, 256, seed=10 + i) for i in range(3)]
    noises = [synth.synth_noise(2, seed=20 + i) for i in range(3)]
    losses = orc.train_steps(sd, batches, toggles, noises, lr=1e-4)
    np.testing.assert_allclose(losses, G["train3lo_losses"], rtol=2e-5)
    for k in [f[len("train3lo_param_head/"):] for f in G.files if f.startswith("train3lo_param_head/")]:
        np.testing.assert_allclose(sd[k].flatten()[:8].numpy(), G[f"train3lo_param_head/{k}"], rtol=1e-3, atol=1e-6)


def test_plain_deeplab(c1):
    _, x, y, _ = c1
    psd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["simpleDeepV3Plus"]], seed=0)
    loss = orc.mrfp_forward(psd, x, y, training=True, perturb=False)
    assert abs(loss.item() - float(G["plain_loss"])) / float(G["plain_loss"]) < 1e-5


@pytest.mark.parametrize("hw", [(256, 256), (512, 512), (768, 768), (1024, 2048)])
def test_hrfp_size_chain_and_nearest_index(hw):
    """SURVEY section 8 A4 size chains + the ATen nearest rule restated in numpy float32."""
    import math
    import torch.nn.functional as F
    expect = {(256, 256): [64, 77, 92, 110, 128, 128, 107, 85, 64],
              (512, 512): [128, 154, 184, 220, 256, 256, 214, 170, 128],
              (768, 768): [192, 231, 277, 332, 384, 384, 321, 256, 192],
              (1024, 2048): [256, 308, 369, 442, 512, 512, 429, 342, 256]}[hw]
    h = hw[0]
    size = h // 4
    assert size == expect[0]
    for i, (_, _, _, kind, arg) in enumerate(orc.HRFP_STAGES):
        if kind == "scale":
            out, sf = orc.nearest_out_size(size, arg), arg
        elif kind == "half":
            out, sf = int(h / 2), None
        else:
            out, sf = math.ceil(h / 4), None
        assert out == expect[i + 1]
        probe = torch.arange(size, dtype=torch.float32).view(1, 1, 1, size)
        got = (F.interpolate(probe, scale_factor=(1.0, sf)) if sf is not None
               else F.interpolate(probe, size=(1, out))).flatten().long().numpy()
        np.testing.assert_array_equal(got, orc.nearest_src_index(size, out, sf))
        size = out


def test_wider_resnet38_trunk_matches_reference():
    """oracle.wider_resnet_a2 against tests/golden/wrn38.npz (produced by the reference's network/wider_resnet.py,
    tests/golden/make_golden_wrn.py): train mode with the injected Dropout2d masks, backward, eval mode."""
    from wrn_common import GW, stats, trunk_case
    sd, x, gy, masks = trunk_case()
    assert len(sd) == 228 and sum(v.numel() for v in sd.values() if v.is_floating_point()) == 105118656
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    new_stats, taps = {}, {}
    out = orc.wider_resnet_a2(work, x, True, new_stats=new_stats, drop_masks=masks, taps=taps)
    np.testing.assert_allclose(stats(out), GW["out_stats"], rtol=1e-5)
    np.testing.assert_allclose(out[:, 100:108, 2:6, 2:6].detach().numpy(), GW["out_crop"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(stats(taps["mod3"]), GW["mod3_stats"], rtol=1e-5)
    (out * gy).sum().backward()
    for f in GW.files:
        if f.startswith("grad_l2/"):
            k = f[len("grad_l2/"):]
            ref = float(GW[f])
            assert abs(leaf[k].grad.double().pow(2).sum().sqrt().item() - ref) / ref < 1e-4, k
        if f.startswith("running/"):
            np.testing.assert_allclose(new_stats[f[len("running/"):]][:8].numpy(), GW[f], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        oe = orc.wider_resnet_a2({k: v.clone() for k, v in sd.items()}, x, False)
    np.testing.assert_allclose(stats(oe), GW["eval_out_stats"], rtol=1e-5)


def test_well_conditioned_fixture_matches_reference():
    """oracle vs tests/golden/mrfp_wc.npz (reference MRFPPlus at 4x192x192 with residual_gain 0.3 weights; the fixture
    where the reference's own fp32 noise is < 1e-4, so the north_star's 1e-3 is asserted plainly on the GPU)."""
    from golden_common import CROP as C, GWC, TAGS as T, stats, wc_case
    sd, x, y, noise = wc_case()
    keys = orc.trainable_keys(sd)
    assert list(GWC["grad_keys"]) == keys
    for tag in ("ttt",):
        leaf = {k: sd[k].clone().requires_grad_(True) for k in keys}
        work = {k: v.clone() for k, v in sd.items()}
        work.update(leaf)
        taps = {}
        loss = orc.mrfp_forward(work, x, y, training=True, toggles=T[tag], noise=noise, taps=taps)
        assert abs(loss.item() - float(GWC[f"{tag}_loss"])) / float(GWC[f"{tag}_loss"]) < 1e-6
        assert float(GWC[f"{tag}_logits_noise"]) < 4e-4
        np.testing.assert_allclose(taps["logits"].detach()[C].numpy(), GWC[f"{tag}_logits_crop"], rtol=0, atol=2e-5)
        for name, t in taps.items():
            if name != "logits":
                np.testing.assert_allclose(stats(t), GWC[f"{tag}_tap/{name}"], rtol=1e-5, err_msg=name)
        grads = torch.autograd.grad(loss, [leaf[k] for k in keys])
        l2 = np.array([g.double().pow(2).sum().sqrt().item() for g in grads])
        np.testing.assert_allclose(l2, GWC[f"{tag}_grad_l2"], rtol=2e-3, atol=1e-9)


def test_resnet101_trunk_matches_reference():
    """oracle.resnet_trunk against the numbers the reference's resnet101 (ResNet3X3, Resnet.py:338-512, 678-693) produced
    (tests/golden/make_golden_r101.py part 1): train forward + backward, running statistics, eval."""
    from golden_common import GR, r101_trunk_case, stats
    sd, x, gy = r101_trunk_case()
    assert sum(1 for k in sd if k.startswith("layer3.") and k.endswith("conv1.weight")) == 23
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    ns, taps = {}, {}
    out = orc.resnet_trunk(work, x, True, new_stats=ns, taps=taps)
    np.testing.assert_allclose(stats(out), GR["trunk_out_stats"], rtol=1e-5)
    np.testing.assert_allclose(out.detach()[:, 200:208].numpy(), GR["trunk_out_crop"], rtol=1e-4, atol=1e-6)
    for name, t in taps.items():
        np.testing.assert_allclose(stats(t), GR["trunk_tap/" + name], rtol=1e-5, err_msg=name)
    (out * gy).sum().backward()
    l2 = np.array([leaf[k].grad.double().pow(2).sum().sqrt().item() for k in GR["trunk_grad_keys"]])
    np.testing.assert_allclose(l2, GR["trunk_grad_l2"], rtol=1e-3, atol=1e-9)
    for f in GR.files:
        if f.startswith("trunk_running/"):
            np.testing.assert_allclose(ns[f[len("trunk_running/"):]][:8].numpy(), GR[f], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        oe = orc.resnet_trunk({k: v.clone() for k, v in sd.items()}, x, False)
    np.testing.assert_allclose(stats(oe), GR["trunk_eval_stats"], rtol=1e-5)


def test_resnet101_mrfp_composition_matches_reference_forward():
    """oracle.mrfp_forward on the trunk='resnet-101' keys against the reference's own MRFPPlus.forward run on reference
    parts (tests/golden/make_golden_r101.py part 2) -- the network bench.py times."""
    from golden_common import CROP as C, GR, TAGS as T, r101_comp_case, stats
    sd, x, y, noise = r101_comp_case()
    keys = orc.trainable_keys(sd)
    assert list(GR["comp_grad_keys"]) == keys
    tag = "ttt"
    leaf = {k: sd[k].clone().requires_grad_(True) for k in keys}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    taps = {}
    loss = orc.mrfp_forward(work, x, y, training=True, toggles=T[tag], noise=noise, taps=taps)
    assert abs(loss.item() - float(GR[f"comp_{tag}_loss"])) / float(GR[f"comp_{tag}_loss"]) < 1e-6
    np.testing.assert_allclose(taps["logits"].detach()[C].numpy(), GR[f"comp_{tag}_logits_crop"], rtol=0, atol=5e-5)
    for name, t in taps.items():
        if name != "logits":
            np.testing.assert_allclose(stats(t), GR[f"comp_{tag}_tap/{name}"], rtol=2e-5, err_msg=name)
    grads = torch.autograd.grad(loss, [leaf[k] for k in keys])
    l2 = np.array([g.double().pow(2).sum().sqrt().item() for g in grads])
    np.testing.assert_allclose(l2, GR[f"comp_{tag}_grad_l2"], rtol=5e-3, atol=1e-9)
    hist = orc.eval_hist({k: v.clone() for k, v in sd.items()}, x, y)
    assert np.abs(hist - GR["comp_eval_hist"]).sum() <= 8
