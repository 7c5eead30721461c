"""HIP path against the two fixtures of round 2, both produced by the reference's own code:

* tests/golden/mrfp_wc.npz -- reference `MRFPPlus` (ResNet-50) at 4x192x192 with well-conditioned synthetic weights
  (residual_gain 0.3): the reference's own fp32-vs-fp64 logits noise is < 1e-4 there, so the north_star tolerance is
  asserted PLAINLY: loss and logits within 1e-3 relative for all four toggle sets, NP+ on in two of them; every
  per-stage statistic (stem, np1, hrfp0-7, layer1-4, aspp, dec1) within 1e-3, so a drift is attributable to a kernel.
* tests/golden/r101.npz -- the reference's resnet101 (`ResNet3X3`) trunk, and the MRFP+ composition on it run by the
  reference's own `MRFPPlus.forward` (make_golden_r101.py): the network bench.py times.

Gradients: L2 norm of EVERY trainable tensor against the reference's fp32 value, within 3x the reference's own
fp32-vs-fp64 noise for that tensor + 1e-3 (the stem-side gradients in front of an InstanceNorm are differences of large
terms: the reference itself is 1e-2 off there), and the elements stored in the fixture.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

from golden_common import CROP, GR, GWC, SPEC, TAGS, r101_comp_case, r101_trunk_case, relerr, stats, trunk_key, wc_case
from oracle import mrfp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-3          # north_star: logits / loss within 1e-3 relative, fp32
BF16_TOL = 3e-2


def _model(trunk, sd, dtype=torch.float32, fuse_ce=False):
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE, cfg.MODEL.FUSE_UPSAMPLE_CE = dtype, fuse_ce
    with contextlib.redirect_stdout(io.StringIO()):
        m = deepv3.MRFPPlus(19, trunk=trunk, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    m.load_state_dict(sd)
    return m.to(DEV)


def _run_train(model, x, y, noise, tag):
    from mrfp_amd.deepv3 import InjectedRandom
    model.train()
    model.rng = InjectedRandom(TAGS[tag], noise)
    cap, taps = {}, {}
    orig = model._loss
    model._loss = lambda out, g: (cap.__setitem__("logits", out), orig(out, g))[1]
    model._taps = taps
    loss = model(x.to(DEV), y.to(DEV), training=True)
    loss.backward()
    model._taps = None
    return loss, cap["logits"].float(), taps


def _check_train(model, G, pre, tag, x, y, noise, keys, tol, grad_tol):
    loss, logits, taps = _run_train(model, x, y, noise, tag)
    ref = float(G[f"{pre}{tag}_loss"])
    assert abs(loss.item() - ref) / ref < tol, (tag, loss.item(), ref)
    report = {}
    for name, t in taps.items():                       # per-stage statistics: localises a drift
        f = f"{pre}{tag}_tap/{name}"
        if f in G.files:
            got, want = stats(t.float()), G[f]
            report[name] = max(abs(got[1] - want[1]) / want[1], abs(got[2] - want[2]) / want[2])
    bad = {k: v for k, v in report.items() if v >= tol}
    assert not bad, (tag, bad, report)
    assert len(report) >= 14
    e = relerr(logits[CROP], G[f"{pre}{tag}_logits_crop"])
    assert e < tol, (tag, "logits", e, report)
    np.testing.assert_allclose(stats(logits), G[f"{pre}{tag}_logits_stats"], rtol=tol, atol=1e-5)
    params = dict(model.named_parameters())
    ref_l2, self_noise = G[f"{pre}{tag}_grad_l2"], G[f"{pre}{tag}_grad_self_noise"]
    for k, r, nz in zip(keys, ref_l2, self_noise):
        if r < 1e-7:                                  # mathematically-zero gradients (bias in front of a norm)
            continue
        got = params[k].grad.double().pow(2).sum().sqrt().item()
        assert abs(got - r) / r <= 3 * nz + grad_tol, (tag, k, got, r, nz)
    for f in G.files:
        if f.startswith(f"{pre}{tag}_grad_head/"):
            k = f.split("/", 1)[1]
            r = ref_l2[keys.index(k)]
            nz = self_noise[keys.index(k)]
            np.testing.assert_allclose(params[k].grad.flatten()[:8].cpu().numpy(), G[f], rtol=10 * nz + 5 * grad_tol,
                                       atol=(10 * nz + 5 * grad_tol) * r / np.sqrt(params[k].numel()), err_msg=k)
    for n, p in model.named_parameters():
        if n.startswith("OC"):
            assert p.grad is None
    return report


# ---------------------------------------------------------------------------------------------------------------------
# ResNet-50 MRFP+ (the reference class), well-conditioned: plain 1e-3
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["ttt", "fff", "tft", "ftf"])
def test_r50_well_conditioned_plain_1e3(tag):
    sd, x, y, noise = wc_case()
    assert float(GWC[f"{tag}_logits_noise"]) < 4e-4
    model = _model("resnet-50", sd)
    _check_train(model, GWC, "", tag, x, y, noise, list(GWC["grad_keys"]), RTOL, 1e-3)


def test_r50_well_conditioned_bf16():
    sd, x, y, noise = wc_case()
    model = _model("resnet-50", sd, torch.bfloat16)
    try:
        loss, logits, taps = _run_train(model, x, y, noise, "ttt")
    finally:
        from mrfp_amd.config import cfg
        cfg.MODEL.ACT_DTYPE = torch.float32
    ref = float(GWC["ttt_loss"])
    assert abs(loss.item() - ref) / ref < BF16_TOL
    for name, t in taps.items():
        got, want = stats(t.float()), GWC[f"ttt_tap/{name}"]
        assert abs(got[2] - want[2]) / want[2] < BF16_TOL, name


# ---------------------------------------------------------------------------------------------------------------------
# ResNet-101 (ResNet3X3): trunk vs the reference class, composition vs the reference's forward
# ---------------------------------------------------------------------------------------------------------------------
def _r101_trunk(sd, dtype=torch.float32):
    from mrfp_amd.config import cfg
    from mrfp_amd.network import Resnet
    cfg.MODEL.ACT_DTYPE = dtype
    m = Resnet.resnet101(pretrained=False, wt_layer=[0, 0, 4, 4, 4, 0, 0])
    own = m.state_dict()
    load = {trunk_key(k): v for k, v in sd.items()}
    load.update({k: v for k, v in own.items() if k.startswith("fc.")})
    m.load_state_dict(load)
    return m.to(DEV)


def test_r101_trunk_train_vs_reference_golden():
    sd, x, gy = r101_trunk_case()
    m = _r101_trunk(sd).train()
    out = m(x.to(DEV))
    (out.float() * gy.to(DEV)).sum().backward()
    np.testing.assert_allclose(stats(out), GR["trunk_out_stats"], rtol=RTOL)
    assert relerr(out[:, 200:208], GR["trunk_out_crop"]) < RTOL
    params = {k: v for k, v in m.named_parameters()}
    # gradient noise band from the live oracle (fp32 vs fp64), as for the other trunks
    g = {}
    for dtype in (torch.float32, torch.float64):
        leaf = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
        work = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        work.update(leaf)
        (orc.resnet_trunk(work, x.to(dtype), True) * gy.to(dtype)).sum().backward()
        g[dtype] = {k: v.grad.detach().double() for k, v in leaf.items()}
    # Every tensor inside the band -- except for ReLU sign flips: at this size layer4 has 32 pixels per channel, and ONE
    # pre-activation that lands on the other side of zero (|y| below its own fp32 rounding error, ~1e-6 of the cases)
    # moves the bias gradient of that BatchNorm and the weight gradient in front of it by ~1e-2 relative.  The oracle's
    # fp32 run has such events too (layer3.21 here: 1.2e-2), but not on the same pixels, so its noise band cannot cover
    # them (tools/debug_r101_trunk.py prints the table).  Allowed: at most 2 % of the tensors outside the band, each
    # below 3e-2; a wrong kernel fails both counts.
    outside, n_t = [], 0
    for k, ref64 in g[torch.float64].items():
        n64 = ref64.norm().item()
        if n64 < 1e-6:
            continue
        n_t += 1
        noise = (g[torch.float32][k] - ref64).norm().item() / n64
        err = (params[trunk_key(k)].grad.detach().double().cpu() - ref64).norm().item() / n64
        assert err < 3e-2, (k, err, noise)
        if err > 3 * noise + 2e-4:
            outside.append((k, err, noise))
    assert len(outside) <= 0.02 * n_t, outside
    floor = 1e-4 * float(np.median(GR["trunk_grad_l2"]))             # below: mathematically-zero gradients (a bias in
    for k, r in zip(GR["trunk_grad_keys"], GR["trunk_grad_l2"]):     # front of an InstanceNorm), rounding noise only
        if r > floor:                                                # the reference run's own numbers
            got = params[trunk_key(str(k))].grad.double().norm().item()
            assert abs(got - r) / r < 5e-3, (k, got, r)
    msd = m.state_dict()
    for f in GR.files:
        if f.startswith("trunk_running/"):
            np.testing.assert_allclose(msd[trunk_key(f[len("trunk_running/"):])][:8].cpu().numpy(), GR[f], rtol=2e-3, atol=1e-5)


def test_r101_trunk_eval_vs_reference_golden():
    sd, x, _ = r101_trunk_case()
    m = _r101_trunk(sd).eval()
    with torch.no_grad():
        out = m(x.to(DEV))
    np.testing.assert_allclose(stats(out), GR["trunk_eval_stats"], rtol=RTOL)
    assert relerr(out[:, 200:208], GR["trunk_eval_crop"]) < RTOL


@pytest.mark.parametrize("tag", ["ttt", "fff", "tft", "ftf"])
def test_r101_mrfp_plus_vs_reference_forward(tag):
    """The benchmarked network against the reference's own MRFPPlus.forward run on reference ResNet3X3 parts."""
    sd, x, y, noise = r101_comp_case()
    assert [k for k, _ in SPEC["MRFPPlus_r101"]] == list(sd.keys())
    model = _model("resnet-101", sd)
    assert list(model.state_dict().keys()) == list(sd.keys())
    _check_train(model, GR, "comp_", tag, x, y, noise, list(GR["comp_grad_keys"]), RTOL, 1e-3)
    if tag == "ttt":
        msd = model.state_dict()
        for f in GR.files:
            if f.startswith("comp_ttt_running/"):
                np.testing.assert_allclose(msd[f.split("/", 1)[1]][:8].cpu().numpy(), GR[f], rtol=2e-3, atol=1e-5)


def test_r101_mrfp_plus_fused_head_and_bf16():
    """Same network through the production head (fused upsample + CE) in fp32, and with bf16 activations (the bench
    dtype; stated tolerance 3e-2 on the loss and on every stage's L2)."""
    sd, x, y, noise = r101_comp_case()
    from mrfp_amd.config import cfg
    from mrfp_amd.deepv3 import InjectedRandom
    ref = float(GR["comp_ttt_loss"])
    try:
        for dtype, tol in ((torch.float32, RTOL), (torch.bfloat16, BF16_TOL)):
            model = _model("resnet-101", sd, dtype, fuse_ce=True).train()
            model.rng = InjectedRandom(TAGS["ttt"], noise)
            taps = {}
            model._taps = taps
            loss = model(x.to(DEV), y.to(DEV), training=True)
            loss.backward()
            assert abs(loss.item() - ref) / ref < tol, (dtype, loss.item(), ref)
            for name, t in taps.items():
                got, want = stats(t.float()), GR[f"comp_ttt_tap/{name}"]
                assert abs(got[2] - want[2]) / want[2] < tol, (dtype, name)
            keys = list(GR["comp_grad_keys"])
            params = dict(model.named_parameters())
            for k in ("final2.0.weight", "final1.4.weight", "final1.0.weight"):
                r = GR["comp_ttt_grad_l2"][keys.index(k)]
                assert abs(params[k].grad.double().norm().item() - r) / r < 10 * tol, (dtype, k)
            del model
    finally:
        cfg.MODEL.ACT_DTYPE, cfg.MODEL.FUSE_UPSAMPLE_CE = torch.float32, True


def test_r101_eval_hist_miou():
    sd, x, y, _ = r101_comp_case()
    model = _model("resnet-101", sd).eval()
    with torch.no_grad():
        logits = model(x.to(DEV), training=False)
    assert relerr(logits[CROP], GR["comp_eval_logits_crop"]) < RTOL
    np.testing.assert_allclose(stats(logits), GR["comp_eval_logits_stats"], rtol=RTOL, atol=1e-5)
    from mrfp_amd import metrics, ops
    hist, _ = ops.argmax_hist(logits, y.to(DEV))
    hist = hist.cpu().numpy()
    assert np.abs(hist - GR["comp_eval_hist"]).sum() <= 0.002 * hist.sum()
    assert abs(100 * metrics.miou_from_hist(hist) - 100 * float(GR["comp_eval_miou"])) < 0.1


# ---------------------------------------------------------------------------------------------------------------------
# Every gradient tensor of the benchmarked network, ELEMENT-WISE, against the live oracle (gradient norms plus a
# few stored elements would let a permuted or mis-strided gradient with the right norm through).  The oracle is pinned to the
# reference on exactly this case (make_golden_r101.py: every gradient rel 0.0), so "within the oracle's own fp32-vs-fp64
# band" is "as close to the exact gradient as the reference's fp32 run is".
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_grads(sd, x, y, noise, tag):
    keys = orc.trainable_keys(sd)
    out = {}
    for dtype in (torch.float32, torch.float64):
        leaf = {k: sd[k].clone().to(dtype).requires_grad_(True) for k in keys}
        work = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        work.update(leaf)
        nz = {k: v.to(dtype) for k, v in noise.items()}
        loss = orc.mrfp_forward(work, x.to(dtype), y, training=True, toggles=TAGS[tag], noise=nz)
        out[dtype] = (loss.item(), dict(zip(keys, (g.double() for g in torch.autograd.grad(loss, [leaf[k] for k in keys])))))
    return out


def _check_grads_elementwise(params, og, what):
    """Relative L2 distance of every gradient TENSOR to the fp64 evaluation within 3x the oracle's own fp32-vs-fp64 distance
    (+2e-4); the allowance for ReLU sign flips of test_r101_trunk_train_vs_reference_golden (at most 2 % of the tensors outside
    the band, none beyond 3e-2)."""
    g32, g64 = og[torch.float32][1], og[torch.float64][1]
    outside, n_t = [], 0
    for k, ref64 in g64.items():
        n64 = ref64.norm().item()
        if n64 < 1e-7:                                # mathematically-zero gradients (a bias in front of a norm)
            continue
        n_t += 1
        got = params[k].grad
        assert got is not None and tuple(got.shape) == tuple(ref64.shape), (what, k)
        noise = (g32[k] - ref64).norm().item() / n64
        err = (got.detach().double().cpu() - ref64).norm().item() / n64
        assert err < 3e-2 + 2 * noise, (what, k, err, noise)      # (stem-side tensors in front of an InstanceNorm: the reference's
        if err > 3 * noise + 2e-4:                                #  own fp32 run is 2.6e-2 from the fp64 value there)
            outside.append((k, err, noise))
    assert n_t >= 300 and len(outside) <= 0.02 * n_t, (what, n_t, outside)


@pytest.mark.parametrize("tag", ["ttt", "fff", "tft", "ftf"])
def test_r101_mrfp_plus_every_gradient_elementwise(tag):
    sd, x, y, noise = r101_comp_case()
    og = _oracle_grads(sd, x, y, noise, tag)
    assert abs(og[torch.float32][0] - float(GR[f"comp_{tag}_loss"])) / float(GR[f"comp_{tag}_loss"]) < 1e-6   # the pinned loss
    model = _model("resnet-101", sd)
    loss, _, _ = _run_train(model, x, y, noise, tag)
    assert abs(loss.item() - og[torch.float64][0]) / og[torch.float64][0] < RTOL
    _check_grads_elementwise(dict(model.named_parameters()), og, tag)


def test_r101_production_path_pinned_and_switches_bit_identical():
    """(1) The PRODUCTION forward (no `_taps`: one-pass NP+ with the HRFP output added in the apply pass, fused upsample + CE
    head) against the reference-pinned loss and, element-wise, every gradient of the live oracle, in fp32.  (2) The same model
    with bf16 activations: the byte-saving kernels that are on by default (sign mask, gated skip gradient, pointwise and
    row-reuse convolution kernels) against the plain kernels they replace -- loss and EVERY gradient bit-identical, so a kernel
    regression shows at model level, not only in the op-level tests."""
    import os
    import subprocess
    import sys
    from mrfp_amd import ops
    from mrfp_amd.config import cfg
    from mrfp_amd.deepv3 import InjectedRandom
    sd, x, y, noise = r101_comp_case()
    og = _oracle_grads(sd, x, y, noise, "ttt")
    try:
        model = _model("resnet-101", sd, torch.float32, fuse_ce=True).train()
        model.rng = InjectedRandom(TAGS["ttt"], noise)
        assert getattr(model, "_taps", None) is None
        loss = model(x.to(DEV), y.to(DEV), training=True)
        loss.backward()
        ref = float(GR["comp_ttt_loss"])
        assert abs(loss.item() - ref) / ref < RTOL, (loss.item(), ref)
        _check_grads_elementwise(dict(model.named_parameters()), og, "production fp32")
        del model

        def bf16_run():
            m = _model("resnet-101", sd, torch.bfloat16, fuse_ce=True).train()
            m.rng = InjectedRandom(TAGS["ttt"], noise)
            ls = m(x.to(DEV), y.to(DEV), training=True)
            ls.backward()
            return ls.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        l_on, g_on = bf16_run()
        was = ops.SIGN_MASK[0], ops.GATED_SKIP[0]
        ops.SIGN_MASK[0], ops.GATED_SKIP[0] = False, False
        try:
            l_off, g_off = bf16_run()
        finally:
            ops.SIGN_MASK[0], ops.GATED_SKIP[0] = was
        assert torch.equal(l_on, l_off) and g_on.keys() == g_off.keys()
        for k in g_on:
            assert torch.equal(g_on[k], g_off[k]), k
        assert abs(l_on.item() - ref) / ref < BF16_TOL
    finally:
        cfg.MODEL.ACT_DTYPE, cfg.MODEL.FUSE_UPSAMPLE_CE = torch.float32, True
    # the convolution kernel choice is read once per process: the generic kernels in a child process, compared through a file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_parity_pinned_gpu as T\n"
        "from mrfp_amd.deepv3 import InjectedRandom\n"
        "sd, x, y, noise = T.r101_comp_case()\n"
        "m = T._model('resnet-101', sd, torch.bfloat16, fuse_ce=True).train()\n"
        "m.rng = InjectedRandom(T.TAGS['ttt'], noise)\n"
        "ls = m(x.to(T.DEV), y.to(T.DEV), training=True); ls.backward()\n"
        "torch.save({'loss': ls.detach().cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, sys.argv[1])\n"
        % (root, os.path.join(root, "tests")))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        outs = []
        for i, extra in enumerate(({}, {"MRFP_CONV_PW": "0", "MRFP_CONV_RR": "0", "MRFP_WGRAD_DENSE": "0"})):
            f = os.path.join(td, "r%d.pt" % i)
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (extra, r.stderr[-2000:])
            outs.append(torch.load(f))
    assert torch.equal(outs[0]["loss"], l_on.cpu())                       # the child reproduces this process bit for bit
    # generic kernels: other tile shapes and summation orders -> equal to rounding, not to the bit.  Only the loss is compared:
    # at this size (3 x 192 x 192, NP+ on) the gradients of the fp32 model are already 1-3 % from the fp64 evaluation
    # (tools/bf16_grad_noise.py), and with bf16 activations two correct evaluations differ by O(1) per tensor.
    assert abs(outs[1]["loss"].item() - outs[0]["loss"].item()) / abs(outs[0]["loss"].item()) < 5e-3
    for k, g in outs[1]["g"].items():
        assert torch.isfinite(g).all(), k
