"""Timing of the group whitening passes (csrc/whiten.hip) against the HBM roofline.

    python tools/whiten_micro.py [B C H W] [reps]

Algorithmic bytes: group_moments = one read of x; group_apply = one read + one write; backward pair = cross moments
(read dy, x) + fused apply (read dy, x; write dx).  Also times a whole SwitchWhiten2d forward + backward on the group
kernels and on the generic (full Gram + per-image GEMM) passes.  One JSON line per measurement.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import ops  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, C, H, W = a[:4] if len(a) >= 4 else (16, 256, 192, 192)
    reps = a[4] if len(a) >= 5 else 20
    dev = "cuda:0"
    for dtype in (torch.bfloat16, torch.float32):
        x = ops.empty_cl(B, C, H, W, dtype, dev)
        x.copy_(torch.randn(B, C, H, W, device=dev))
        dy = ops.empty_cl(B, C, H, W, dtype, dev)
        dy.copy_(torch.randn(B, C, H, W, device=dev))
        Wm = torch.randn(B, C // 16, 16, 16, device=dev) * 0.2
        sh = torch.randn(B, C, device=dev)
        plane = B * C * H * W * x.element_size()
        for name, fn, nbytes in (("group_moments(x,x)", lambda: ops._gm_call(x, x), plane),
                                 ("group_moments(dy,x)", lambda: ops._gm_call(dy, x), 2 * plane),
                                 ("group_apply", lambda: ops._ga_call(x, Wm, shift=sh), 2 * plane),
                                 ("group_apply fused bwd", lambda: ops._ga_call(dy, Wm, z=x, Vm=Wm, shift=sh), 3 * plane)):
            ms = timed(fn, reps)
            print(json.dumps({"op": name, "dtype": str(dtype).split(".")[-1], "shape": [B, C, H, W], "ms": round(ms, 4),
                              "algorithmic_GBps": round(nbytes / ms / 1e6, 1),
                              "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}), flush=True)
        from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
        for fast in (True, False):
            if not fast and B * C * H * W > 16 * 256 * 96 * 96:
                continue                      # the generic path loops over images with full C x C Grams: small shapes only
            sw = SwitchWhiten2d(C, num_pergroup=16, sw_type=2).to(dev).train()
            sw.use_group_kernels = fast
            xr = x.detach().requires_grad_(True)

            def step():
                sw(xr).backward(dy)
            ms = timed(step, max(3, reps // 4))
            print(json.dumps({"op": "SwitchWhiten2d fwd+bwd (%s)" % ("group kernels" if fast else "generic passes"),
                              "dtype": str(dtype).split(".")[-1], "shape": [B, C, H, W], "ms": round(ms, 3),
                              "algorithmic_GBps": round(8 * plane / ms / 1e6, 1),
                              "frac_of_hbm_peak": round(8 * plane / (ms * 1e-3) / HBM_PEAK, 4)}), flush=True)


if __name__ == "__main__":
    main()
