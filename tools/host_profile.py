"""cProfile of the host side of one train step (which Python frames the ~28 ms of launch overhead go to)."""
import contextlib, cProfile, io, os, pstats, sys
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
x, y = synth.synth_batch(2, 256, 256, seed=1)
x, y = x.to(dev), y.to(dev)
for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    trainer.step(x, y)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
