import sys, torch
sys.path.insert(0, '.')
from mrfp_amd import ops
which = sys.argv[1]
dev = 'cuda:0'
s = torch.randn(4, 2, 16, 16, device=dev)
cov = (s @ s.transpose(-1, -2) + 16 * torch.eye(16, device=dev)).requires_grad_(True)
def f_simple(c):
    return (c * 2.0 + 1.0).sum((-1, -2))
def f_softmax(c):
    w = torch.softmax(torch.ones(2, device=dev), 0)
    return c * w[0] + torch.eye(16, device=dev).view(1, 1, 16, 16) * w[1]
def f_isqrt(c):
    return ops.group_isqrt(c, 5)
def f_inplace(c):
    with torch.no_grad():
        buf.mul_(0.9).add_(0.1 * c.mean(0))
    return c + buf
buf = torch.zeros(2, 16, 16, device=dev)
f = {'simple': f_simple, 'softmax': f_softmax, 'isqrt': f_isqrt, 'inplace': f_inplace}[which]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        y = f(cov); torch.autograd.grad(y, cov, torch.ones_like(y))
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = f(cov)
print(which, 'fwd captured', flush=True)
gy = torch.ones_like(y)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, pool=g.pool()):
    gr = torch.autograd.grad(y, cov, gy)
print(which, 'bwd captured', flush=True)
g.replay(); g2.replay(); torch.cuda.synchronize()
print(which, 'ok', float(gr[0].sum()), flush=True)
