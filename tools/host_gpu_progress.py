"""Where is the host relative to the GPU inside a train step?  Host clock and a stream event at the entry of every stage of
MRFPPlus.forward, at the start of backward and at the optimizer step: the lead (event time - host time, both from the step's start) shows
where the launch queue drains.   gpurun -- python tools/host_gpu_progress.py"""
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
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter(), e))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(name)
        return f(*a, **k)
    setattr(obj, attr, g)
for attr in ("_stem", "_hrfp", "_low", "_high", "_final1", "_head", "_head_o2"):
    if hasattr(model, attr): wrap(model, attr, attr)
model.aspp.register_forward_pre_hook(lambda m, a: mark("aspp"))
model.layer3.register_forward_pre_hook(lambda m, a: mark("layer3"))
model.layer4.register_forward_pre_hook(lambda m, a: mark("layer4"))
model.layer2.register_forward_pre_hook(lambda m, a: mark("layer2"))
wrap(model.rng, "reinit_hrfp", "reinit_hrfp")
wrap(tr.opt, "step", "opt.step")
wrap(tr.opt, "zero_grad", "zero_grad")
import mrfp_amd.harness as H
orig_fb = tr._fwd_bwd
for _ in range(4):
    tr.step(x, y)
torch.cuda.synchronize()
marks.clear()
for _ in range(3):
    tr.step(x, y)
mark("end")
torch.cuda.synchronize()
t0h, e0 = marks[0][1], marks[0][2]
for name, th, e in marks:
    print("%-12s host %8.2f ms   gpu %8.2f ms   host lead %7.2f ms" % (name, 1e3 * (th - t0h), e0.elapsed_time(e), e0.elapsed_time(e) - 1e3 * (th - t0h)))
