"""Every BASELINE.json config at its STATED size, once each, on the GPU box.

configs[0] (2x256x256 CPU-reference case) is tests/test_model_gpu.py; here:
  configs[1]  ResNet-50 MRFP+, 8x512x512, fp32 -- against the live CPU oracle on the same 8 images (forward: loss and every
              per-stage statistic within 1e-3; well-conditioned weights, see mrfp_amd/synth.py), plus determinism;
  configs[2]  ResNet-101 MRFP+, 16x768x768, bf16 (the bench workload at its stated batch) -- the oracle is out of reach at
              this size: bitwise reproducibility, finite gradients for every trainable tensor, d(loss)/d(bias) sums to zero;
  configs[4]  WiderResNet-38 MRFP+, 2x1024x2048 (the full-resolution crop), bf16 and float16 -- same properties.
configs[3] (8 ranks) is the driver's; its code path is rehearsed by tests/test_ddp_gpu.py.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

from mrfp_amd import synth
from oracle import mrfp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(trunk, dtype, gain=1.0):
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE, cfg.MODEL.FUSE_UPSAMPLE_CE = dtype, True
    with contextlib.redirect_stdout(io.StringIO()):
        m = deepv3.MRFPPlus(19, trunk=trunk, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    sd = synth.synth_state_dict(synth.spec_of(m.state_dict()), seed=0, residual_gain=gain)
    m.load_state_dict(sd)
    return m.to(DEV).train(), sd


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.abs().mean().item(), t.pow(2).sum().sqrt().item()])


def test_config1_resnet50_8x512_fp32_vs_live_oracle():
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    try:
        model, sd = _make("resnet-50", torch.float32, gain=0.3)
        B, S = 8, 512
        x, y = synth.synth_batch(B, S, S, seed=71)
        noise = synth.synth_noise(B, seed=72)
        runs = []
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            model.rng = deepv3.InjectedRandom((True, True, True), noise)
            taps = {}
            model._taps = taps
            loss = model(x.to(DEV), y.to(DEV), training=True)
            loss.backward()
            model._taps = None
            runs.append((loss.item(), model.layer3[5].conv2.weight.grad.clone(), model.final1[0].weight.grad.clone()))
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
        torch.set_num_threads(16)
        otaps = {}
        with torch.no_grad():
            lo = orc.mrfp_forward({k: v.clone() for k, v in sd.items()}, x, y, training=True, toggles=(True, True, True),
                                  noise=noise, taps=otaps)
        assert abs(runs[0][0] - lo.item()) / lo.item() < 1e-3, (runs[0][0], lo.item())
        for name, t in taps.items():
            got, want = _stats(t.float()), _stats(otaps[name])
            assert np.abs(got - want).max() / want.max() < 1e-3 and abs(got[1] - want[1]) / want[1] < 1e-3, (name, got, want)
        assert len(taps) >= 14
    finally:
        cfg.MODEL.ACT_DTYPE = torch.float32


def _size_independent_properties(trunk, dtype, B, H, W, np1_channels, loss_scale=1.0, drop_masks=None):
    """(1) the same step twice from the same state: identical loss and gradients, bit for bit; (2) every trainable tensor gets
    a finite gradient, the frozen HRFP branch none; (3) d(loss)/d(classifier bias) sums to zero over the classes."""
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    try:
        model, _ = _make(trunk, dtype)
        x, y = synth.synth_batch(B, H, W, seed=81)
        x, y = x.to(DEV), y.to(DEV)
        noise = {k: v.to(DEV) for k, v in synth.synth_noise(B, seed=82, channels=(np1_channels, 256)).items()}
        probe = [n for n, p in model.named_parameters() if p.requires_grad][::17]
        runs = []
        if drop_masks is not None:              # the Dropout2d draws of the WiderResNet, fixed for both runs
            from mrfp_amd.network import wider_resnet
            wider_resnet.DROP_MASKS.injected = {k: v.reshape(B, -1).to(DEV) for k, v in drop_masks.items()}
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            model.rng = deepv3.InjectedRandom((True, True, True), noise)
            loss = model(x, y, training=True)
            loss.backward(torch.full_like(loss, loss_scale))
            params = dict(model.named_parameters())
            runs.append((loss.item(), [params[n].grad.clone() for n in probe]))
        assert np.isfinite(runs[0][0]) and runs[0][0] == runs[1][0]
        for a, b in zip(runs[0][1], runs[1][1]):
            assert torch.equal(a, b)
        for n, p in model.named_parameters():
            if n.startswith("OC"):
                assert p.grad is None
            else:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
        gb = model.final2[0].bias.grad.double()
        assert abs(gb.sum().item()) < 1e-3 * gb.abs().sum().item() + 1e-7 * loss_scale
    finally:
        cfg.MODEL.ACT_DTYPE = torch.float32
        if drop_masks is not None:
            wider_resnet.DROP_MASKS.injected = None


def test_config2_resnet101_16x768_bf16_properties():
    _size_independent_properties("resnet-101", torch.bfloat16, 16, 768, 768, 128)


@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.float16, 65536.0)])
def test_config4_wrn38_2x1024x2048_properties(dtype, scale):
    from wrn_common import drop_masks
    _size_independent_properties("wider_resnet38_a2", dtype, 2, 1024, 2048, 128, loss_scale=scale, drop_masks=drop_masks(2, 83))
