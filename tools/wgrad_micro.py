"""Weight-gradient launches of the bench step, single and grouped, under the current environment switches (MRFP_WGRAD_BIG=0/1/2,
MRFP_WGRAD_WGS, ...): microseconds per problem and TFLOP/s.   python tools/wgrad_micro.py [reps [shape index [single|grouped]]]
(one shape, one form: the program behind `rocprofv3 --pmc` in tools/run_pmc_wgrad.sh)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import _lib  # noqa: E402
from mrfp_amd._lib import call, ptr, stream  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# (B, H, W, C, N, k, stride, pad, dil, group)
SHAPES = [(16, 48, 48, 256, 1024, 1, 1, 0, 1, 23), (16, 48, 48, 1024, 256, 1, 1, 0, 1, 22), (16, 48, 48, 256, 256, 3, 1, 1, 1, 22),
          (16, 48, 48, 512, 512, 3, 1, 2, 2, 3), (16, 48, 48, 512, 2048, 1, 1, 0, 1, 3), (16, 48, 48, 2048, 512, 1, 1, 0, 1, 2),
          (16, 192, 192, 256, 256, 3, 1, 1, 1, 1), (16, 192, 192, 320, 256, 3, 1, 1, 1, 1), (16, 48, 48, 2048, 256, 3, 1, 12, 12, 1),
          (16, 48, 48, 1024, 2048, 1, 1, 0, 1, 1), (16, 96, 96, 128, 512, 1, 1, 0, 1, 4), (16, 96, 96, 512, 128, 1, 1, 0, 1, 3),
          (16, 192, 192, 64, 256, 1, 1, 0, 1, 3), (16, 96, 96, 512, 256, 1, 1, 0, 1, 1), (16, 48, 48, 1280, 256, 1, 1, 0, 1, 1),
          (16, 384, 384, 64, 64, 3, 1, 1, 1, 1), (16, 384, 384, 64, 128, 3, 1, 1, 1, 1), (16, 192, 192, 64, 64, 3, 1, 1, 1, 3), (16, 96, 96, 128, 128, 3, 1, 1, 1, 3)]
L = _lib.lib()
if len(sys.argv) > 2:
    SHAPES = [SHAPES[int(sys.argv[2])]]
FORMS = sys.argv[3:4] or ["singles", "grouped"]
for (B, H, W, C, N, k, st, pad, dil, G) in SHAPES:
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    M, Q = B * Ho * Wo, k * k * C
    xs = [torch.randn(B, H, W, C, device="cuda").bfloat16() for _ in range(G)]
    dys = [torch.randn(B, Ho, Wo, N, device="cuda").bfloat16() for _ in range(G)]
    dws = [torch.empty(N, C, k, k, device="cuda") for _ in range(G)]
    arr = ctypes.c_void_p * G
    ax, ay, aw = arr(*[ptr(t) for t in xs]), arr(*[ptr(t) for t in dys]), arr(*[ptr(t) for t in dws])
    wsg = torch.empty(int(L.mrfp_conv_wgrad_grouped_ws_bytes(M, N, Q, G)), dtype=torch.uint8, device="cuda")
    ws1 = torch.empty(int(L.mrfp_conv_wgrad_ws_bytes(M, N, Q)), dtype=torch.uint8, device="cuda")

    def grouped():
        call("mrfp_conv_wgrad_grouped", ax, ay, aw, G, ptr(wsg), _lib.BF16, B, H, W, C, C, N, N, k, k, Ho, Wo, st, pad, pad, dil, stream())

    def singles():
        for g in range(G):
            call("mrfp_conv_wgrad", ptr(xs[g]), ptr(dys[g]), ptr(dws[g]), ptr(ws1), _lib.BF16, B, H, W, C, C, N, N, k, k, Ho, Wo, st, pad, pad, dil,
                 stream())
    out = []
    for fn in (singles, grouped):
        if not any(fn.__name__.startswith(f[:6]) for f in FORMS):
            out.append(float("nan"))
            continue
        fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps / G * 1e3)
        out.append(best)
    fl = 2.0 * M * N * Q
    print("%-44s G %2d  single %7.1f us %5.0f TF/s | grouped %7.1f us/problem %5.0f TF/s  (slab MB: single %.0f, grouped %.0f)" % (
        str((H, C, N, k, st, dil)), G, out[0], fl / out[0] / 1e6, out[1], fl / out[1] / 1e6, ws1.numel() / 1e6, wsg.numel() / 1e6 / G), flush=True)
