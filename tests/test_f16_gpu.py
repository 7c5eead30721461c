"""float16 activations (MRFP_F16; BASELINE.json configs[4] names an "fp16 MFMA conv path"): the same kernels
instantiated for IEEE half.  Inputs are rounded to float16 first, so only accumulation order and output rounding
differ from the fp32 CPU computation: tolerance 2e-3 of the tensor maximum (11-bit mantissa; bf16 uses 1e-2), model
loss within 5e-3, and one Trainer step with the static loss scale must reproduce the fp32 step's classifier update
to 10 % relative L2 (the bf16 criterion of tests/test_model_gpu.py)."""
import contextlib
import io
import json
import os

import pytest
import torch
import torch.nn.functional as F

from mrfp_amd import synth
from oracle import mrfp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H16 = torch.float16
HERE = os.path.dirname(os.path.abspath(__file__))
SPEC = json.load(open(os.path.join(HERE, "golden", "state_dict_spec.json")))
CL = torch.channels_last


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def r16(t):
    return t.half().float()


@pytest.mark.parametrize("case", [(2, 64, 20, 18, 64, 3, 1, 1, 1, True), (2, 256, 16, 16, 512, 1, 2, 0, 1, False),
                                  (1, 512, 8, 8, 256, 3, 1, 12, 12, False), (2, 256, 12, 10, 19, 1, 1, 0, 1, True),
                                  (2, 128, 240, 240, 256, 3, 1, 1, 1, True), (16, 256, 48, 48, 256, 3, 1, 1, 1, False)])
def test_conv_f16(case):
    from mrfp_amd import conv
    B, Cin, H, W, Cout, k, st, pad, dil, has_bias = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = r16(torch.randn(B, Cin, H, W, generator=g))
    w = r16(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1 if has_bias else None
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yc = F.conv2d(xc, wc, b, st, pad, dil)
    gy = r16(torch.randn(yc.shape, generator=g))
    yc.backward(gy)
    xd = x.to(DEV, H16).contiguous(memory_format=CL).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    yd = conv.conv2d(xd, wd, b.to(DEV) if has_bias else None, st, pad, dil)
    assert yd.dtype == H16
    yd.backward(gy.to(DEV, H16).contiguous(memory_format=CL))
    assert relerr(yd, yc) < 2e-3 and relerr(xd.grad, xc.grad) < 2e-3 and relerr(wd.grad, wc.grad) < 4e-3


def test_row_ops_f16():
    """max-pool -> BatchNorm(train) + residual + ReLU -> bilinear, forward and backward (the max-pool comes first so
    that its arg-max is taken on exactly representable inputs: no rounding-induced ties)."""
    from mrfp_amd import ops
    g = torch.Generator().manual_seed(5)
    x = r16(torch.randn(2, 64, 33, 45, generator=g) * 2 + 0.5)
    res = r16(torch.randn(2, 64, 17, 23, generator=g))
    w, b = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    xc, rc = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yc = F.max_pool2d(xc, 3, 2, 1)
    yc = F.relu(F.batch_norm(yc, None, None, w, b, True, 0.1, 1e-5) + rc)
    yc = F.interpolate(yc, size=(30, 20), mode="bilinear", align_corners=True)
    gy = r16(torch.randn(yc.shape, generator=g))
    yc.backward(gy)
    xd = x.to(DEV, H16).contiguous(memory_format=CL).requires_grad_(True)
    rd = res.to(DEV, H16).contiguous(memory_format=CL).requires_grad_(True)
    yd = ops.batch_norm_act(ops.max_pool_3x3_s2(xd), w.to(DEV), b.to(DEV), None, None, training=True, relu=True, res=rd)
    yd = ops.upsample_bilinear(yd, (30, 20))
    yd.backward(gy.to(DEV, H16).contiguous(memory_format=CL))
    assert yd.dtype == H16
    assert relerr(yd, yc) < 3e-3 and relerr(xd.grad, xc.grad) < 1e-2 and relerr(rd.grad, rc.grad) < 3e-3


def test_upsample_ce_f16():
    from mrfp_amd import ops
    g = torch.Generator().manual_seed(9)
    P = r16(torch.randn(2, 19, 12, 10, generator=g) * 3)
    y = torch.randint(0, 19, (2, 48, 40), generator=g)
    y[0, :3] = 255
    Pc = P.clone().requires_grad_(True)
    lc = orc.cross_entropy_255(orc.upsample_bilinear_ac(Pc, (48, 40)), y)
    lc.backward()
    Pd = torch.zeros(2, 24, 12, 10)
    Pd[:, :19] = P
    Pd = Pd.to(DEV, H16).contiguous(memory_format=CL).requires_grad_(True)
    ld = ops.upsample_cross_entropy(Pd, y.to(DEV), (48, 40), 19)
    ld.backward(torch.tensor(4096.0, device=DEV))          # scaled backward: 1/#pixels gradients survive float16
    assert abs(ld.item() - lc.item()) / lc.item() < 1e-3
    assert relerr(Pd.grad[:, :19].float() / 4096.0, Pc.grad) < 5e-3


def _model(dtype):
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE = dtype
    with contextlib.redirect_stdout(io.StringIO()):
        m = deepv3.MRFPPlus(19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    m.load_state_dict(synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["MRFPPlus"]], seed=0))
    m = m.to(DEV).train()
    m.rng = deepv3.InjectedRandom((True, True, True), synth.synth_noise(2, seed=2))
    return m


def test_model_step_f16_with_loss_scale():
    """One train step in float16 with the static loss scale against the same step in fp32 (same weights, inputs,
    noise): loss within 5e-3, update of the (well-conditioned) classifier parameters within 10 % relative L2 of the fp32 update
    (weights in front of a BatchNorm have a mathematically near-zero gradient: pure noise in any precision)."""
    import numpy as np
    from mrfp_amd.config import cfg
    from mrfp_amd.harness import Trainer
    G = np.load(os.path.join(HERE, "golden", "mrfp_c1.npz"))
    x, y = synth.synth_batch(2, 256, 256, seed=1)
    out = {}
    try:
        for dtype in (torch.float32, H16):
            m = _model(dtype)
            tr = Trainer(m)
            assert tr.loss_scale == (65536.0 if dtype == H16 else 1.0)
            before = {k: p.detach().clone() for k, p in m.named_parameters() if k in ("final2.0.weight", "final2.0.bias")}
            loss = tr.step(x.to(DEV), y.to(DEV))
            out[dtype] = (loss.item(), {k: (dict(m.named_parameters())[k].detach() - v) for k, v in before.items()})
    finally:
        cfg.MODEL.ACT_DTYPE = torch.float32
    ref = float(G["ttt_loss"])
    assert abs(out[torch.float32][0] - ref) / ref < 1e-3
    assert abs(out[H16][0] - ref) / ref < 5e-3
    for k in out[H16][1]:          # relative L2 of the update (batch 2 + NP+ is ill-conditioned: DESIGN.md section 2)
        d16, d32 = out[H16][1][k].double(), out[torch.float32][1][k].double()
        assert ((d16 - d32).pow(2).sum().sqrt() / d32.pow(2).sum().sqrt()).item() < 0.1, k
