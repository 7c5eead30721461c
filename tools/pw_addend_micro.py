"""Pointwise dgrad forms of layer 3 (256 -> 1024 @48^2 and 128 -> 512 @96^2): plain / + addend / + gated addend, microseconds per launch (device events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import _lib
from mrfp_amd._lib import call, ptr, stream
for (B, H, W, C, N) in [(16, 48, 48, 256, 1024), (16, 96, 96, 128, 512), (16, 48, 48, 512, 2048), (16, 192, 192, 64, 256)]:
    x = torch.randn(B, H, W, C, device="cuda").bfloat16()
    wp = (torch.randn(N * C, device="cuda") * 0.05).bfloat16()
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    ad = torch.randn(B, H, W, N, device="cuda").bfloat16()
    mask = torch.randint(0, 256, (B * H * W * N // 8,), dtype=torch.uint8, device="cuda")
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")       # cache flush between reps
    def plain(): call("mrfp_conv_fwd", ptr(x), ptr(wp), None, ptr(y), _lib.BF16, B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1, None, None, stream())
    def add(): call("mrfp_conv_fwd", ptr(x), ptr(wp), None, ptr(y), _lib.BF16, B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1, ptr(ad), None, stream())
    def gated(): call("mrfp_conv_fwd_gated", ptr(x), ptr(wp), None, ptr(y), _lib.BF16, B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1, ptr(ad), ptr(mask), stream())
    out = []
    for fn in (plain, add, gated):
        for cold in (False, True):
            ts = []
            for _ in range(8):
                if cold: big.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            out.append(sorted(ts)[len(ts) // 2])
    mb = (B * H * W * (C + N) * 2) / 1e6
    print("%s  plain %.1f / cold %.1f   +addend %.1f / %.1f   +gated %.1f / %.1f us   (x+y %.0f MB, addend %.0f MB)" % ((H, C, N), *out, mb, B * H * W * N * 2 / 1e6), flush=True)
