"""WiderResNet-38-A2 on the HIP kernels (BASELINE.json configs[4] backbone).

(1) The trunk (mrfp_amd/network/wider_resnet.py) against the golden vectors produced by the reference's own
    network/wider_resnet.py (tests/golden/wrn38.npz) and against the live CPU oracle for every gradient.
(2) The BUILD-DEFINED MRFP+ composition on that trunk (parity unpinned: no reference behaviour exists) against the
    live CPU oracle.  Tolerance as everywhere: 1e-3 relative fp32 on outputs / loss, gradients within the oracle's
    own fp32-vs-fp64 noise band (x3), bf16 looser and stated.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

from mrfp_amd import synth
from oracle import mrfp_oracle as orc
from wrn_common import GW, SPEC, drop_masks, stats, trunk_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _trunk(sd, dtype=torch.float32):
    from mrfp_amd.config import cfg
    from mrfp_amd.network import wider_resnet
    cfg.MODEL.ACT_DTYPE = dtype
    m = wider_resnet.wider_resnet38_a2(classes=0, dilation=True)
    m.load_state_dict(sd)
    return m.to(DEV)


def _oracle_grads(sd, x, gy, masks, dtype):
    leaf = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    work.update(leaf)
    out = orc.wider_resnet_a2(work, x.to(dtype), True, drop_masks={k: v.to(dtype) for k, v in masks.items()})
    (out * gy.to(dtype)).sum().backward()
    return out.detach(), {k: v.grad.detach() for k, v in leaf.items()}


def test_trunk_train_vs_reference_golden_and_oracle():
    from mrfp_amd.network import wider_resnet
    sd, x, gy, masks = trunk_case()
    m = _trunk(sd).train()
    wider_resnet.DROP_MASKS.injected = {k: v.reshape(2, -1) for k, v in masks.items()}
    try:
        out = m(x.to(DEV))
        (out.float() * gy.to(DEV)).sum().backward()
    finally:
        wider_resnet.DROP_MASKS.injected = None
    np.testing.assert_allclose(stats(out), GW["out_stats"], rtol=1e-3)
    assert relerr(out[:, 100:108, 2:6, 2:6], GW["out_crop"]) < 1e-3
    o32, g32 = _oracle_grads(sd, x, gy, masks, torch.float32)
    o64, g64 = _oracle_grads(sd, x, gy, masks, torch.float64)
    assert relerr(out, o64) < max(1e-3, 3 * relerr(o32, o64))
    params = dict(m.named_parameters())
    for k, ref64 in g64.items():
        n64 = ref64.pow(2).sum().sqrt().item()
        if n64 < 1e-6:
            continue
        noise = (g32[k].double() - ref64).pow(2).sum().sqrt().item() / n64
        err = (params[k].grad.detach().double().cpu() - ref64).pow(2).sum().sqrt().item() / n64
        assert err <= 3 * noise + 2e-4, (k, err, noise)
    for f in GW.files:          # numbers stored from the reference run itself
        if f.startswith("grad_l2/"):
            k = f[len("grad_l2/"):]
            ref = float(GW[f])
            assert abs(params[k].grad.double().pow(2).sum().sqrt().item() - ref) / ref < 2e-3, k
    msd = m.state_dict()
    for f in GW.files:
        if f.startswith("running/"):
            np.testing.assert_allclose(msd[f[len("running/"):]][:8].cpu().numpy(), GW[f], rtol=2e-3, atol=1e-5)


def test_trunk_eval_vs_reference_golden():
    sd, x, _, _ = trunk_case()
    m = _trunk(sd).eval()
    with torch.no_grad():
        out = m(x.to(DEV))
    np.testing.assert_allclose(stats(out), GW["eval_out_stats"], rtol=1e-3)
    assert relerr(out[:, 100:108, 2:6, 2:6], GW["eval_out_crop"]) < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_mrfp_plus_on_wrn38_vs_live_oracle(dtype, tol):
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    from mrfp_amd.network import wider_resnet
    cfg.MODEL.ACT_DTYPE = dtype
    with contextlib.redirect_stdout(io.StringIO()):
        model = deepv3.MRFPPlus(19, trunk="wider_resnet38_a2", criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    spec = [(k, tuple(s)) for k, s in SPEC["MRFPPlus_wrn38"]]
    assert [k for k, _ in spec] == list(model.state_dict().keys())
    sd = synth.synth_state_dict(spec, seed=0)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    B, S = 2, 128
    x, y = synth.synth_batch(B, S, S, seed=5)
    noise = synth.synth_noise(B, seed=6, channels=(128, 256))
    masks = drop_masks(B, 7)
    model.rng = deepv3.InjectedRandom((True, True, True), noise)
    wider_resnet.DROP_MASKS.injected = {k: v.reshape(B, -1) for k, v in masks.items()}
    try:
        loss = model(x.to(DEV), y.to(DEV), training=True)
        loss.backward()
    finally:
        wider_resnet.DROP_MASKS.injected = None
        cfg.MODEL.ACT_DTYPE = torch.float32
    keys = [k for k in ("final2.0.weight", "final1.3.weight", "aspp.features.2.0.weight", "mod7.block1.convs.conv3.weight",
                        "mod3.block1.convs.conv1.weight")]
    leaf = {k: sd[k].clone().requires_grad_(True) for k in keys}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    nz = dict(noise)
    nz.update(masks)
    lo = orc.mrfp_forward(work, x, y, training=True, toggles=(True, True, True), noise=nz)
    assert abs(loss.item() - lo.item()) / lo.item() < tol
    grads = torch.autograd.grad(lo, [leaf[k] for k in keys])
    params = dict(model.named_parameters())
    for k, g in zip(keys[:2], grads[:2]):      # the well-conditioned head of the network
        ref = g.double().pow(2).sum().sqrt().item()
        assert abs(params[k].grad.double().pow(2).sum().sqrt().item() - ref) / ref < 10 * tol, k
    for n, p in model.named_parameters():
        if n.startswith("OC"):
            assert p.grad is None
