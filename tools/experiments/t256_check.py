import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mrfp_amd import conv
from mrfp_amd.config import cfg
import torch.nn.functional as F
cfg.MODEL.ACT_DTYPE = torch.bfloat16
torch.manual_seed(0)
for (B, C, H, W, N, k, pad, dil) in [(16, 128, 96, 96, 256, 3, 1, 1), (4, 256, 192, 192, 256, 3, 2, 2), (16, 512, 48, 48, 2048, 1, 0, 1), (3, 64, 100, 90, 512, 3, 1, 1)]:
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, C, k, k, device="cuda") * 0.05)
    with torch.no_grad():
        y = conv.conv2d(x, w, None, 1, pad, dil)
        st = getattr(y, "_mrfp_colstats", None)
        ref = F.conv2d(x.float(), w.bfloat16().float(), None, 1, pad, dil)
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    s_err = -1.0
    if st is not None:
        rows = st[0].view(st[1], 2, N).double().sum(0)
        yr = y.float().double()
        s_ref = torch.stack([yr.sum((0, 2, 3)), (yr * yr).sum((0, 2, 3))])
        s_err = ((rows - s_ref).abs().max() / s_ref.abs().max()).item()
    print((B, C, H, W, N, k), "max rel err %.2e  stats err %.2e  checksum %.6f" % (err, s_err, y.float().sum().item()))
