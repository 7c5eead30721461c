"""Model-level evidence for the bench dtype (missing #5).  The reference trains in fp32 only (reference main.py:826-864);
the bench runs bf16 activations with fp32 statistics / master weights / accumulation.  At 3 x 192 x 192 with plain synthetic
weights two correct bf16 evaluations differ by O(1) per gradient tensor (tools/bf16_grad_noise.py), so element-wise gradient
checks say nothing there.  This test uses what IS robust: the same 60 SGD steps (reference recipe: lr 1e-2, momentum 0.9, wd 5e-4,
poly schedule, main.py:826-839, 857-864) on one fixed learnable batch, with the same injected toggles / NP+ noise, once in fp32 and
once in bf16 on the HIP path, well-conditioned weights (residual_gain 0.3: the regime of a trained network):
  * the loss curves agree point-wise after the first steps,
  * the final train-batch mIoU agrees,
  * the FIRST-step gradients of the head / ASPP / layer4 / layer3 point the same way (cosine similarity of the fp32 and bf16
    gradient of each parameter group).
"""
import pytest
import torch

from mrfp_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS = 60
TOGGLES = [(True, True, True), (False, True, False), (True, False, True), (True, True, False)]


def _learnable_batch(B=4, S=256, seed=3):
    """19 classes in 32 x 32 blobs, the image = a fixed colour per class + noise (0..255 like the reference's ToTensor without /255)"""
    g = torch.Generator().manual_seed(seed)
    low = torch.randint(0, 19, (B, S // 32, S // 32), generator=g)
    y = low.repeat_interleave(32, 1).repeat_interleave(32, 2)
    colours = torch.rand(19, 3, generator=g) * 255.0
    x = colours[y].permute(0, 3, 1, 2) + 20.0 * torch.randn(B, 3, S, S, generator=g)
    y = y.clone()
    y[torch.rand(B, S, S, generator=g) < 0.03] = 255
    return x.clamp(0, 255).contiguous(), y


def _run(dtype):
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    from mrfp_amd.deepv3 import InjectedRandom
    from mrfp_amd.harness import Trainer, evaluate
    cfg.MODEL.ACT_DTYPE = dtype
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            m = deepv3.MRFPPlus(19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
        m.load_state_dict(synth.synth_state_dict(synth.spec_of(m.state_dict()), seed=0, residual_gain=0.3))
        m = m.to(DEV).train()
        x, y = _learnable_batch()
        x, y = x.to(DEV), y.to(DEV)
        noise = {k: v.to(DEV) for k, v in synth.synth_noise(4, seed=4).items()}
        tr = Trainer(m, lr=1e-2)
        # first-step gradients (nothing is updated by _fwd_bwd)
        m.rng = InjectedRandom(TOGGLES[0], noise)
        tr._fwd_bwd(x, y)
        torch.cuda.synchronize()
        names = {id(p): n for n, p in m.named_parameters()}
        groups = {}
        for p, o in zip(tr.opt.params, tr.opt.offsets):
            n = names[id(p)]
            for g in ("final1", "aspp", "layer4", "layer3"):
                if n.startswith(g + "."):
                    groups.setdefault(g, []).append(tr.opt.flat_g[o:o + p.numel()].double().cpu())
        grads = {g: torch.cat(v) for g, v in groups.items()}
        losses = []
        for i in range(STEPS):
            m.rng = InjectedRandom(TOGGLES[i % len(TOGGLES)], noise)
            losses.append(float(tr.step(x, y)))
        m.rng = InjectedRandom((False, False, False), noise)
        _, miou, _ = evaluate(m, [(x, y)])
        return losses, float(miou), grads
    finally:
        cfg.MODEL.ACT_DTYPE = torch.float32


def test_bf16_training_tracks_fp32():
    l32, m32, g32 = _run(torch.float32)
    l16, m16, g16 = _run(torch.bfloat16)
    cos = {k: float(torch.dot(g32[k], g16[k]) / (g32[k].norm() * g16[k].norm())) for k in g32}
    rel = [abs(a - b) / abs(a) for a, b in zip(l32, l16)]
    print("\n[bf16 fidelity] loss fp32 first/last %.4f %.4f  bf16 %.4f %.4f  max rel diff after step 5: %.4f (at all steps %.4f)"
          % (l32[0], l32[-1], l16[0], l16[-1], max(rel[5:]), max(rel)))
    print("[bf16 fidelity] train-batch mIoU fp32 %.3f  bf16 %.3f   first-step gradient cosine %s" % (m32, m16, cos))
    assert l32[-1] < 0.5 * l32[0] and l16[-1] < 0.5 * l16[0]                 # both runs learn the batch
    assert max(rel[5:]) < 0.03, rel
    assert abs(m32 - m16) <= 0.5, (m32, m16)
    for k, c in cos.items():
        assert c >= 0.98, (k, c)
