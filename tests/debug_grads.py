"""Debug aid: per-parameter gradient error of the HIP model against the live CPU oracle (C1 shape)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrfp_amd import synth, deepv3
from mrfp_amd.config import cfg
from oracle import mrfp_oracle as orc
backend = sys.argv[1] if len(sys.argv) > 1 else "hip"
tg = tuple(c == "t" for c in (sys.argv[2] if len(sys.argv) > 2 else "fff"))
cfg.MODEL.CONV_BACKEND = backend
SPEC = json.load(open(os.path.join(ROOT, "tests/golden/state_dict_spec.json")))
sd = synth.synth_state_dict([(k, tuple(s)) for k, s in SPEC["MRFPPlus"]], seed=0)
model = deepv3.MRFPPlus(19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
model.load_state_dict(sd); model = model.to("cuda:0").train()
x, y = synth.synth_batch(2, 256, 256, seed=1)
noise = synth.synth_noise(2, seed=2)
model.rng = deepv3.InjectedRandom(tg, noise)
loss = model(x.cuda(), y.cuda(), training=True); loss.backward()
keys = orc.trainable_keys(sd)
leaf = {k: sd[k].clone().requires_grad_(True) for k in keys}
work = {k: v.clone() for k, v in sd.items()}; work.update(leaf)
lo = orc.mrfp_forward(work, x, y, training=True, toggles=tg, noise=noise)
grads = torch.autograd.grad(lo, [leaf[k] for k in keys])
params = dict(model.named_parameters())
print("loss", loss.item(), lo.item())
for k, g in zip(keys, grads):
    gm = params[k].grad.detach().cpu().double(); g = g.double()
    err = ((gm - g).pow(2).sum().sqrt() / g.pow(2).sum().sqrt()).item()
    print("%-45s relL2 %.3e" % (k, err))
