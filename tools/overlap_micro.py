"""Can a weight-gradient launch (MFMA-bound) hide inside HBM-bound normalisation passes?  Two streams: stream A runs a chain of
BatchNorm apply passes over a large tensor, stream B one weight-gradient launch; timed serially (B after A on one stream) and
concurrently.  MRFP_WGRAD_LDS=81920 caps the weight-gradient kernel at one workgroup per CU (it then leaves register file and wave
slots to the other stream).   python tools/overlap_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import _lib  # noqa: E402
from mrfp_amd._lib import call, ptr  # noqa: E402

dev = "cuda:0"
B, H, W, C, N, k = 16, 192, 192, 256, 256, 3
x = torch.randn(B, H, W, C, device=dev).bfloat16()
dy = torch.randn(B, H, W, N, device=dev).bfloat16()
dw = torch.empty(N, C, k, k, device=dev)
L = _lib.lib()
ws = torch.empty(int(L.mrfp_conv_wgrad_ws_bytes(B * H * W, N, k * k * C)), dtype=torch.uint8, device=dev)
# normalisation chain: affine_fwd over a 16 x 192 x 192 x 256 tensor (302 MB read + 302 MB written per pass)
a = torch.randn(B, H, W, C, device=dev).bfloat16()
b = torch.empty_like(a)
A, S = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
NPASS = int(os.environ.get("NPASS", "6"))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def norm(stream):
    for _ in range(NPASS):
        call("mrfp_affine_fwd", ptr(a), None, ptr(b), _lib.BF16, B, H, W, C, H, W, None, None, ptr(A), ptr(S), 0, 1, stream.cuda_stream)


def wgrad(stream):
    call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), _lib.BF16, B, H, W, C, C, N, N, k, k, H, W, 1, 1, 1, 1, stream.cuda_stream)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        sa.synchronize(); sb.synchronize()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


def both():
    main = torch.cuda.current_stream()
    sa.wait_stream(main); sb.wait_stream(main)
    norm(sa); wgrad(sb)
    main.wait_stream(sa); main.wait_stream(sb)


def serial():
    main = torch.cuda.current_stream()
    sa.wait_stream(main)
    norm(sa); wgrad(sa)
    main.wait_stream(sa)


def only(fn):
    def f():
        main = torch.cuda.current_stream()
        sa.wait_stream(main)
        fn(sa)
        main.wait_stream(sa)
    return f


tn, tw, ts, tb = timed(only(norm)), timed(only(wgrad)), timed(serial), timed(both)
print("MRFP_WGRAD_LDS=%s  norm chain %.0f us  wgrad %.0f us  serial %.0f us  concurrent %.0f us  (hidden: %.0f us = %.0f %% of the shorter one)" % (
    os.environ.get("MRFP_WGRAD_LDS", "-"), tn, tw, ts, tb, ts - tb, 100 * (ts - tb) / min(tn, tw)))
