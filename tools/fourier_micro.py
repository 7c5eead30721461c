"""Timing of the Fourier amplitude perturbation (mrfp_fourier_mix) against the HBM roofline.

    python tools/fourier_micro.py [B C H W] [reps]

Algorithmic bytes per call: read x, read the partner sample, write y = 3 planes of
B*H*W*C*sizeof(dtype).  Set MRFP_FFT_GENERIC=1 to time the generic four-pass Stockham path instead of the
two-step register path.  Prints one JSON line per (dtype, band).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import _lib, ops  # noqa: E402
from mrfp_amd._lib import call, dt, ptr, stream  # noqa: E402

HBM_PEAK = 8.0e12


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, C, H, W = a[:4] if len(a) >= 4 else (16, 128, 192, 192)
    reps = a[4] if len(a) >= 5 else 20
    dev = "cuda:0"
    for dtype in (torch.bfloat16, torch.float32):
        x = ops.empty_cl(B, C, H, W, dtype, dev)
        x.copy_(torch.randn(B, C, H, W, device=dev))
        y = ops.empty_cl(B, C, H, W, dtype, dev)
        nbytes = int(_lib.lib().mrfp_fourier_spectrum_bytes(B, H, W, C))
        S = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        S3 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ratio = torch.empty(B * H * (W // 2 + 1) * C, dtype=torch.float32, device=dev)
        perm = torch.roll(torch.arange(B, device=dev), 1).contiguous()
        twH, twW = ops._twiddles(H, dev), ops._twiddles(W, dev)
        for high in (0, 1):
            def run():
                call("mrfp_fourier_mix", ptr(x), ptr(y), ptr(perm), ptr(S), ptr(S3), ptr(ratio), 0, ptr(twH), ptr(twW),
                     dt(x), B, H, W, C, 16.0, 1.0, high, stream())
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            alg = 3 * B * H * W * C * x.element_size()
            print(json.dumps({"op": "fourier_mix", "dtype": str(dtype).split(".")[-1], "band": "high" if high else "low",
                              "shape": [B, C, H, W], "ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1),
                              "frac_of_hbm_peak": round(alg / (ms * 1e-3) / HBM_PEAK, 4),
                              "stored_bins": int(_lib.lib().mrfp_fourier_stored_bins(H, W, 16.0, high)),
                              "path": "generic" if os.environ.get("MRFP_FFT_GENERIC") == "1" else "two-step"}), flush=True)


if __name__ == "__main__":
    main()
