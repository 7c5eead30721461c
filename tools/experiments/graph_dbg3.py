import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
from mrfp_amd import ops
from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
which = sys.argv[1]
dev = 'cuda:0'
sw = SwitchWhiten2d(64, num_pergroup=16, sw_type=2).to(dev).train()
N, C, H, W = 4, 64, 12, 10
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
hw = float(H * W)
def algebra(s, M):
    sq = torch.diagonal(M, dim1=-2, dim2=-1).sum((1, 2))
    return sw._transform(s / hw, M, sq, N, C, hw)
s, M = ops._gm_call(x, x)
params = [sw.sw_mean_weight, sw.sw_var_weight, sw.weight, sw.bias]
if which == 'direct':
    g = ops._AlgebraGraph(algebra, s, M, params, (sw.running_mean, sw.running_cov))
    print('direct ok', flush=True)
elif which == 'nobuf':
    sw.eval()   # no running-stat update, uses running stats
    sw.running_cov.copy_(torch.eye(16, device=dev))
    g = ops._AlgebraGraph(algebra, s, M, params, ())
    print('nobuf ok', flush=True)
elif which == 'fwdonly':
    sl, Ml = s.clone().requires_grad_(True), M.clone().requires_grad_(True)
    for _ in range(2):
        with torch.enable_grad():
            algebra(sl, Ml)
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        with torch.enable_grad():
            out = algebra(sl, Ml)
    print('fwdonly ok', flush=True)
elif which == 'nograd':
    sl, Ml = s.clone(), M.clone()
    for _ in range(2):
        with torch.no_grad():
            algebra(sl, Ml)
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        with torch.no_grad():
            out = algebra(sl, Ml)
    print('nograd ok', flush=True)
