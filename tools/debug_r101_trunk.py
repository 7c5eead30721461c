"""Per-tensor gradient error of the HIP ResNet-101 trunk against the fp64 oracle (diagnostic for tests/test_parity_pinned_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_common import r101_trunk_case, trunk_key
from oracle import mrfp_oracle as orc
from test_parity_pinned_gpu import _r101_trunk
sd, x, gy = r101_trunk_case()
m = _r101_trunk(sd).train()
out = m(x.cuda())
(out.float() * gy.cuda()).sum().backward()
params = dict(m.named_parameters())
g = {}
outs = {}
for dtype in (torch.float32, torch.float64):
    leaf = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    work = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    work.update(leaf)
    o = orc.resnet_trunk(work, x.to(dtype), True)
    outs[dtype] = o.detach()
    (o * gy.to(dtype)).sum().backward()
    g[dtype] = {k: v.grad.detach().double() for k, v in leaf.items()}
print("out err vs 64: hip %.2e  oracle32 %.2e" % (((out.double().cpu() - outs[torch.float64]).norm() / outs[torch.float64].norm()).item(),
      ((outs[torch.float32].double() - outs[torch.float64]).norm() / outs[torch.float64].norm()).item()))
rows = []
for k, ref64 in g[torch.float64].items():
    n64 = ref64.norm().item()
    if n64 < 1e-6:
        continue
    noise = (g[torch.float32][k] - ref64).norm().item() / n64
    err = (params[trunk_key(k)].grad.detach().double().cpu() - ref64).norm().item() / n64
    rows.append((err / (3 * noise + 2e-4), k, err, noise, n64))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%6.1f  %-40s err %.2e noise %.2e |g| %.2e" % r)
