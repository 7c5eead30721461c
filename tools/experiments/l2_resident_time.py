"""Is the convolutions' fill path limited by COLD misses?  The same per-tile work on an input that fits the L2s vs one that does not."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrfp_amd import conv
from mrfp_amd._lib import call, ptr, dt, stream
CL = torch.channels_last
for (B, C, H, W, N, k, pad, dil) in [(16, 128, 384, 384, 256, 3, 1, 1), (1, 128, 192, 192, 4096, 3, 1, 1), (4, 128, 96, 96, 4096, 3, 1, 1),
                                     (16, 256, 192, 192, 256, 3, 1, 1), (1, 256, 96, 96, 4096, 3, 1, 1), (2, 256, 48, 48, 8192, 3, 1, 1)]:
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
    w = torch.randn(N, C, k, k, device="cuda") * 0.03
    b = torch.randn(N, device="cuda")
    pk = conv.get_pack(w, b, x.dtype, C, N)
    y = torch.empty(B, N, H, W, device="cuda", dtype=x.dtype).contiguous(memory_format=CL)
    def go():
        call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), ptr(pk.bias), ptr(y), dt(x), B, H, W, C, N, N, k, k, H, W, 1, pad, pad, dil, 1, None, None, stream())
    for _ in range(3): go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): go()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("x %6.1f MB  w %5.1f MB  y %7.1f MB  %-40s %.3f ms  %.0f TFLOP/s" % (x.numel() * 2 / 1e6, w.numel() * 2 / 1e6, y.numel() * 2 / 1e6, (B, C, H, W, N, k), ms, 2.0 * B * H * W * N * C * k * k / ms / 1e9))
