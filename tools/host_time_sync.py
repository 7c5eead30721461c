import contextlib, io, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mrfp_amd import deepv3, synth
from mrfp_amd.config import cfg
from mrfp_amd.harness import Trainer
cfg.MODEL.ACT_DTYPE = torch.bfloat16
dev = torch.device("cuda", 0)
if os.environ.get("MRFP_FORCE_SYNC") == "1" or os.environ.get("MRFP_INIT_PG") == "1":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
with contextlib.redirect_stdout(io.StringIO()):
    model = deepv3.MRFPPlus(19, trunk="resnet-101", criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
model.load_state_dict(synth.synth_state_dict(synth.spec_of(model.state_dict()), seed=0))
model = model.to(dev).train()
model.rng = deepv3.InjectedRandom((True, True, True), None, reinit=True)
trainer = Trainer(model)
x, y = synth.synth_batch(16, 768, 768, seed=1)
x, y = x.to(dev), y.to(dev)
import contextlib as _cl
hp = os.environ.get("MRFP_HIPRIO")
ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(hp))) if hp else _cl.nullcontext()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
ctx.__enter__()
for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 6
for _ in range(n):
    trainer.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ctx.__exit__(None, None, None)
print("hiprio=%s " % hp + "pg=%s " % os.environ.get("MRFP_INIT_PG") + "sync=%s host issue %.1f ms/step, total %.1f ms/step" % (os.environ.get("MRFP_FORCE_SYNC"), 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
if dist.is_initialized(): dist.destroy_process_group()
