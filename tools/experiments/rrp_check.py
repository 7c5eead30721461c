"""Correctness of the RRP kernels (row reuse on the plain staging, conv_igemm_kernel<..., RRP>) against the same launch with MRFP_CONV_RRP=0
(the plain tile: same tile, same epilogue -- results must agree to fp32 accumulation-order rounding; here: bit for bit expected, the K order is the same)."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["MRFP_CONV_RRP"] = sys.argv[1]
    os.environ["MRFP_CONV_RR"] = "0"
    from mrfp_amd import conv
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE = torch.bfloat16
    outs = []
    for (B, C, H, W, N, dil) in [(16, 256, 48, 48, 256, 1), (2, 256, 48, 48, 256, 2), (4, 512, 48, 48, 512, 2), (2, 256, 96, 96, 128, 1), (3, 320, 32, 48, 256, 1)]:
        g = torch.Generator().manual_seed(C + N + dil)
        x = torch.randn(B, C, H, W, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(N, C, 3, 3, generator=g) * 0.05).cuda().requires_grad_(True)
        y = conv.conv2d(x, w, None, 1, dil, dil)
        gy = torch.randn(y.shape, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        ref = torch.nn.functional.conv2d(x.detach().float(), w.detach().bfloat16().float(), None, 1, dil, dil)
        err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        outs += [y.detach().float().cpu(), x.grad.float().cpu()]
        print("shape", (B, C, H, W, N, dil), "rel err vs torch fp32 %.2e" % err, flush=True)
    torch.save(outs, "/tmp/rrp_%s.pt" % sys.argv[1])
    sys.exit(0)
for m in ("0", "2"):
    r = subprocess.run([sys.executable, __file__, m], capture_output=True, text=True)
    print("mode", m, "rc", r.returncode, r.stdout, r.stderr[-400:] if r.returncode else "")
a, b = torch.load("/tmp/rrp_0.pt"), torch.load("/tmp/rrp_2.pt")
for i, (u, v) in enumerate(zip(a, b)):
    print(i, "equal" if torch.equal(u, v) else "max abs diff %.3e (max %.3e)" % ((u - v).abs().max().item(), u.abs().max().item()))
