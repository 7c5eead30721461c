"""Per-step wall time of the bench workload from the first step of the FIRST GPU process on a box (is the first run slower, and for how long?):
gpurun -- python tools/cold_start.py [steps]"""
import os, sys, time, contextlib, io
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import synth, deepv3
from mrfp_amd.config import cfg
from mrfp_amd.harness import Trainer
cfg.MODEL.ACT_DTYPE = torch.bfloat16
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    model = deepv3.MRFPPlus(19, trunk="resnet-101", criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
model.load_state_dict(synth.synth_state_dict(synth.spec_of(model.state_dict()), seed=0))
model = model.to(dev).train()
model.rng = deepv3.InjectedRandom((True, True, True), None, reinit=True)
tr = Trainer(model)
x, y = synth.synth_batch(16, 768, 768, seed=1)
x, y = x.to(dev), y.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(x, y)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.1f" % t for t in ts), flush=True)
