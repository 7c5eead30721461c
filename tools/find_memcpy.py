"""Which memcpy / memset activities and host-side synchronisations still happen inside one train step (torch.profiler)."""
import contextlib, io, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import deepv3, synth
from mrfp_amd.config import cfg
from mrfp_amd.harness import Trainer
cfg.MODEL.ACT_DTYPE = torch.bfloat16
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    model = deepv3.MRFPPlus(19, trunk="resnet-101", criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
model.load_state_dict(synth.synth_state_dict(synth.spec_of(model.state_dict()), seed=0))
model = model.to(dev).train()
model.rng = deepv3.InjectedRandom((True, True, True), None, reinit=True)
trainer = Trainer(model)
x, y = synth.synth_batch(16, 768, 768, seed=1)
x, y = x.to(dev), y.to(dev)
for _ in range(2):
    trainer.step(x, y)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    trainer.step(x, y)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "emset" in n or "ynchronize" in n or n in ("aten::item", "aten::_local_scalar_dense", "aten::copy_", "aten::to"):
        c[(n, str(e.input_shapes)[:80])] += 1
for (n, sh), k in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
    print("%5d  %-40s %s" % (k, n[:40], sh))
