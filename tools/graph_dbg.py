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
# usage: tools/graph_dbg.py [graph | graph-old]   (graph-old: the round-2 capture -- on a stream other than the warm-up's)
if len(sys.argv) > 1 and sys.argv[1].startswith("graph"):
    tr.enable_graph()
    tr._debug_capture_on_fresh_stream = sys.argv[1] == "graph-old"
x, y = synth.synth_batch(8, 512, 512, seed=1)
x, y = x.to(dev), y.to(dev)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(x, y)
    torch.cuda.synchronize(); print(i, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
if tr.graph:
    from mrfp_amd.harness import graph_topology
    for key, entry in tr._graphs.items():
        print("captured graph", key[0], graph_topology(entry[0]), flush=True)
