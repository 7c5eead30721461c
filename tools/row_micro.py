"""HBM-bound row kernels (statistics / apply passes of BatchNorm) against the HBM roofline.

    python tools/row_micro.py [reps]

Every call works on a different buffer of a ring that is larger than the 256 MiB Infinity Cache, so the reads
come from HBM as they do inside a training step.  Prints one JSON line per (shape, kernel): algorithmic bytes
(tensor reads + writes), time, GB/s and the fraction of the 8 TB/s peak.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import ops  # noqa: E402

HBM_PEAK = 8.0e12
DEV = "cuda:0"


def timed(fn, n, reps):
    for i in range(2):
        fn(i % n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dtype = torch.bfloat16
    for (B, C, H, W) in [(16, 256, 192, 192), (16, 64, 384, 384), (16, 1024, 48, 48), (16, 256, 48, 48), (16, 512, 96, 96)]:
        nbytes = B * C * H * W * 2
        n = max(2, int(700e6 // (3 * nbytes)) + 1)          # ring: x, dy, y per slot  > 700 MB in total
        xs = [ops.empty_cl(B, C, H, W, dtype, DEV).normal_() for _ in range(n)]
        dys = [ops.empty_cl(B, C, H, W, dtype, DEV).normal_() for _ in range(n)]
        coef = torch.rand(8 * C, device=DEV) + 0.5
        A, S, P, Q, R, mean = (coef[i * C:(i + 1) * C] for i in range(6))
        tests = {
            "stats_fwd (read x)": (1, lambda i: ops._stats_fwd(xs[i], None)),
            "affine_fwd relu (read x, write y)": (2, lambda i: ops._affine_fwd(xs[i], None, A, S, False, True, None)),
            "stats_bwd remask (read dy, x)": (2, lambda i: ops._stats_bwd(dys[i], xs[i], None, mean, False, None, A, S)),
            "affine_bwd remask (read dy, x, write dx)": (3, lambda i: ops._affine_bwd(dys[i], xs[i], None, P, Q, R, False, None,
                                                                                       False, xs[i], A, S)),
            "affine_bwd +dres (read dy, x, y, write dx, dres)": (5, lambda i: ops._affine_bwd(dys[i], xs[i], xs[(i + 1) % n], P, Q, R,
                                                                                              False, None, True, xs[i])),
        }
        for name, (passes, fn) in tests.items():
            ms = timed(fn, n, reps)
            alg = passes * nbytes
            print(json.dumps({"shape": [B, C, H, W], "kernel": name, "MB": round(alg / 1e6, 1), "ms": round(ms, 4),
                              "GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / (ms * 1e-3) / HBM_PEAK, 3)}),
                  flush=True)
        del xs, dys
        torch.cuda.empty_cache()
    # HRFP stages: conv output [Hs,Ws] -> nearest resize -> BN -> ReLU at [Ho,Wo] (the resize is an index table inside
    # the kernels; algorithmic bytes: source tensor for reads, destination tensor for dy / y)
    for (B, C, Hs, Ho) in [(16, 128, 277, 332), (16, 256, 384, 321), (16, 64, 192, 231)]:
        plan = ops.nearest_plan(Hs, Hs, size=(Ho, Ho), device=DEV)
        sb, db = B * C * Hs * Hs * 2, B * C * Ho * Ho * 2
        n = max(2, int(700e6 // (sb + 2 * db)) + 1)
        xs = [ops.empty_cl(B, C, Hs, Hs, dtype, DEV).normal_() for _ in range(n)]
        dys = [ops.empty_cl(B, C, Ho, Ho, dtype, DEV).normal_() for _ in range(n)]
        coef = torch.rand(8 * C, device=DEV) + 0.5
        A, S, P, Q, R, mean = (coef[i * C:(i + 1) * C] for i in range(6))
        tests = {
            "resize stats_fwd (read x)": (sb, lambda i: ops._stats_fwd(xs[i], plan)),
            "resize affine_fwd relu (read x, write y)": (sb + db, lambda i: ops._affine_fwd(xs[i], None, A, S, False, True, plan)),
            "resize stats_bwd remask (read dy, x)": (sb + db, lambda i: ops._stats_bwd(dys[i], xs[i], None, mean, False, plan, A, S)),
            "resize affine_bwd remask (read dy, x, write dx)": (2 * sb + db, lambda i: ops._affine_bwd(dys[i], xs[i], None, P, Q, R, False,
                                                                                                      plan, False, xs[i], A, S)),
        }
        for name, (alg, fn) in tests.items():
            ms = timed(fn, n, reps)
            print(json.dumps({"shape": [B, C, Hs, Hs], "to": Ho, "kernel": name, "MB": round(alg / 1e6, 1), "ms": round(ms, 4),
                              "GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / (ms * 1e-3) / HBM_PEAK, 3)}),
                  flush=True)
        del xs, dys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
