"""Diagnostic: relative L2 distance of every bf16 gradient tensor of the R101 MRFP+ test case to the fp64 oracle, for the
default kernels and for the generic convolution kernels (MRFP_CONV_PW=0 MRFP_CONV_RR=0 set by the caller)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_pinned_gpu as T
from mrfp_amd.deepv3 import InjectedRandom
sd, x, y, noise = T.r101_comp_case()
og = T._oracle_grads(sd, x, y, noise, "ttt")
g64 = og[torch.float64][1]
for dtype in (torch.float32, torch.bfloat16):
    m = T._model("resnet-101", sd, dtype, fuse_ce=True).train()
    m.rng = InjectedRandom(T.TAGS["ttt"], noise)
    ls = m(x.to(T.DEV), y.to(T.DEV), training=True); ls.backward()
    errs = sorted((p.grad.double().cpu() - g64[k]).norm().item() / max(g64[k].norm().item(), 1e-30) for k, p in m.named_parameters() if p.grad is not None and g64[k].norm() > 1e-7)
    print(dtype, "loss", ls.item(), "median", errs[len(errs) // 2], "p90", errs[int(0.9 * len(errs))], "max", errs[-1], flush=True)
