"""Host-side (Python + ctypes launch) time of one train step vs its GPU time."""
import contextlib, io, os, sys, time
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 768
x, y = synth.synth_batch(B, S, S, seed=1)
x, y = x.to(dev), y.to(dev)
for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 6
for _ in range(n):
    trainer.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d size %d: host issue %.1f ms/step, total %.1f ms/step" % (B, S, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
