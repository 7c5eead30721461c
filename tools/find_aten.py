"""Which ATen (non-mrfp) kernels still run inside a train step, and on what shapes (torch.profiler, one step)."""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import deepv3, synth  # noqa: E402
from mrfp_amd.config import cfg  # noqa: E402
from mrfp_amd.harness import Trainer  # noqa: E402

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
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.step(x, y)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if e.key.startswith("aten::") and (dt > 20 or (len(sys.argv) > 1 and e.count >= 8)):
        rows.append((dt, e.key, e.count, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for dt, k, c, sh in (rows if len(sys.argv) > 2 else rows[:60]):
    print("%9.1f us  x%-4d %-28s %s" % (dt, c, k, sh))
