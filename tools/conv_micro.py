"""Micro-benchmark of single convolution shapes through the C ABI (for rocprofv3 --pmc runs)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import conv
from mrfp_amd.config import cfg
cfg.MODEL.ACT_DTYPE = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "big3x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = {"big3x3": (16, 256, 192, 192, 256, 3, 1, 1, 1), "l3_3x3": (16, 256, 48, 48, 256, 3, 1, 1, 1),
          "l3_1x1": (16, 1024, 48, 48, 256, 1, 1, 0, 1), "hrfp": (16, 128, 384, 384, 256, 3, 1, 1, 1),
          "exp1x1": (16, 64, 192, 192, 256, 1, 1, 0, 1), "hrfp128": (16, 256, 384, 384, 128, 3, 1, 1, 1), "n64a": (16, 128, 384, 384, 64, 3, 1, 1, 1), "n64b": (16, 64, 192, 192, 64, 3, 1, 1, 1), "l3_exp": (16, 256, 48, 48, 1024, 1, 1, 0, 1), "l2_exp": (16, 128, 96, 96, 512, 1, 1, 0, 1),
          "dec304": (16, 304, 192, 192, 256, 3, 1, 1, 1), "l4_3x3": (16, 512, 48, 48, 512, 3, 1, 2, 2),
          "l4_red": (16, 2048, 48, 48, 512, 1, 1, 0, 1), "l4_exp": (16, 512, 48, 48, 2048, 1, 1, 0, 1), "l2_3x3": (16, 128, 96, 96, 128, 3, 1, 1, 1),
          "hrfp64_128": (16, 64, 384, 384, 128, 3, 1, 1, 1), "hrfp64_64": (16, 64, 384, 384, 64, 3, 1, 1, 1), "head32": (16, 32, 384, 384, 256, 1, 1, 0, 1),
          "aspp12": (16, 2048, 48, 48, 256, 3, 1, 12, 12), "dec256": (16, 256, 192, 192, 256, 3, 1, 1, 1), "hrfp332": (16, 128, 332, 332, 256, 3, 1, 2, 2),
          "l3_down": (16, 512, 96, 96, 1024, 1, 2, 0, 1)}
B, C, H, W, N, k, st, pad, dil = shapes[which]
x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(N, C, k, k, device="cuda") * 0.05).requires_grad_(True)
y = conv.conv2d(x, w, None, st, pad, dil)
gy = torch.randn_like(y)
mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    if mode == "fwd":
        with torch.no_grad():
            y = conv.conv2d(x, w, None, st, pad, dil)
    elif mode == "wgrad":
        from mrfp_amd import _lib
        from mrfp_amd._lib import call, ptr, dt, stream
        Ho, Wo = y.shape[2], y.shape[3]
        if _ == 0:
            ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(B * Ho * Wo, N, k * k * C)), dtype=torch.uint8, device="cuda")
            dw = torch.empty(N, C, k, k, device="cuda")
        call("mrfp_conv_wgrad", ptr(x), ptr(gy), ptr(dw), ptr(ws), dt(x), B, H, W, C, C, N, N, k, k, Ho, Wo, st, pad, pad, dil, stream())
    else:
        y = conv.conv2d(x, w, None, st, pad, dil)
        y.backward(gy)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * B * (H // st) * (W // st) * N * C * k * k * (3 if mode == "all" else 1)
print(which, mode, "%.3f ms  %.1f TF/s" % (dt * 1e3, fl / dt / 1e12))
