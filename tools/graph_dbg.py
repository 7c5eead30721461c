import os, sys, time, contextlib, io
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.environ.get("MRFP_DBG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mrfp_amd import synth, deepv3
from mrfp_amd.config import cfg
from mrfp_amd.harness import Trainer
cfg.MODEL.ACT_DTYPE = torch.bfloat16
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    model = deepv3.MRFPPlus(19, trunk="resnet-50", criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
model.load_state_dict(synth.synth_state_dict(synth.spec_of(model.state_dict()), seed=0))
model = model.to(dev).train()
model.rng = deepv3.InjectedRandom((True, True, True), None, reinit=True)
tr = Trainer(model)
if len(sys.argv) > 1 and sys.argv[1] == "graph":
    tr.enable_graph()
x, y = synth.synth_batch(8, 512, 512, seed=1)
x, y = x.to(dev), y.to(dev)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(x, y)
    torch.cuda.synchronize(); print(i, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
