"""Which device-to-device copies (hipMemcpyAsync / __amd_rocclr_copyBuffer: Tensor.copy_, clone, contiguous, to) run inside a train step,
and from where: torch.profiler with stacks, one step."""
import collections
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

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    trainer.step(x, y)
    torch.cuda.synchronize()
names = collections.Counter()
sites = collections.Counter()
for e in prof.events():
    if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::mul", "aten::add", "aten::mul_"):
        st = [s for s in (e.stack or []) if "/mrfp_amd/" in s or "bench.py" in s]
        sites[(e.key, str(e.input_shapes)[:60], st[0][-90:] if st else "?")] += 1
    names[e.key] += 1
for (k, sh, s), c in sorted(sites.items(), key=lambda kv: -kv[1])[:50]:
    print("x%-4d %-16s %-62s %s" % (c, k, sh, s))
print({k: v for k, v in names.items() if "Memcpy" in k or "copy" in k.lower()})
