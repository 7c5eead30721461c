"""Yardstick (tools only -- the product path never dispatches to a vendor library): every distinct convolution launch shape of
the bench step (tools/bench_conv_shapes.json, written by `bench.py --dump-convs`) timed through this library's C ABI and through
the stock ROCm path (MIOpen via F.conv2d / torch.nn.grad.*, hipBLASLt via torch.mm for the pointwise layers, which are plain GEMMs
in NHWC), same box, same process, same timing loop (bf16, channels_last, torch.cuda.Event around `reps` back-to-back launches,
best of `sets`).

    python tools/stock_shapes.py [--shapes tools/bench_conv_shapes.json] [--out gpurun_out/vs_stock] [--reps 20] [--only fwd|wgrad]

Writes <out>.json after every shape (a killed run keeps what it measured) and <out>.md at the end:
    launch | shape | ours us | stock us (best of conv / mm) | ratio ours/stock | ours TFLOP/s | stock TFLOP/s
Launch-argument order (include/mrfp_hip.h):
    mrfp_conv_fwd   [B,H,W,C,N,ldy,R,S,Ho,Wo,stride,pad_h,pad_w,dil,sstride]  (sstride > 1: dgrad of a strided convolution run on dy)
    mrfp_conv_wgrad [B,H,W,C,Ctrue,N,ldn,R,S,Ho,Wo,stride,pad_h,pad_w,dil]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import _lib  # noqa: E402
from mrfp_amd._lib import call, ptr, stream  # noqa: E402

CL = torch.channels_last
DT = torch.bfloat16


def timed(fn, reps, sets):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(sets):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3          # us


def rnd(*shape):
    return torch.randn(*shape, device="cuda", dtype=torch.float32).mul_(0.5).to(DT)


def ours_fwd(a):
    B, H, W, C, N, ldy, R, S, Ho, Wo, st, ph, pw, dil, ss = a
    x = rnd(B, H, W, C)
    wp = rnd(N * R * S * C)
    y = torch.empty(B, Ho, Wo, ldy, device="cuda", dtype=DT)
    return lambda: call("mrfp_conv_fwd", ptr(x), ptr(wp), None, ptr(y), _lib.BF16, B, H, W, C, N, ldy, R, S, Ho, Wo, st, ph, pw, dil, ss,
                        None, None, stream())


def ours_wgrad(a):
    B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo, st, ph, pw, dil = a
    x = rnd(B, H, W, C)
    dy = rnd(B, Ho, Wo, ldn)
    dw = torch.empty(N, Ctrue, R, S, device="cuda", dtype=torch.float32)
    ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(B * Ho * Wo, N, R * S * C)), dtype=torch.uint8, device="cuda")
    return lambda: call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), _lib.BF16, B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo, st,
                        ph, pw, dil, stream())


def stock_fwd(a):
    """-> {label: fn}"""
    B, H, W, C, N, ldy, R, S, Ho, Wo, st, ph, pw, dil, ss = a
    out = {}
    if ss == 1:
        x = rnd(B, C, H, W).contiguous(memory_format=CL)
        w = rnd(N, C, R, S).contiguous(memory_format=CL)
        out["miopen"] = lambda: F.conv2d(x, w, None, st, (ph, pw), dil)
        if R == 1 and S == 1 and st == 1 and ph == 0 and pw == 0:
            x2, w2 = rnd(B * H * W, C), rnd(C, N)
            out["hipblaslt"] = lambda: torch.mm(x2, w2)
    else:
        # dgrad of a convolution with stride ss: the launch reads dy [B,H,W,C] and writes dx [B,Ho,Wo,N]
        dy = rnd(B, C, H, W).contiguous(memory_format=CL)
        w = rnd(C, N, R, S).contiguous(memory_format=CL)            # the forward weight [out = C of this launch][in = N]
        po_h, po_w = dil * (R - 1) - ph, dil * (S - 1) - pw         # the forward convolution's padding
        out["miopen"] = lambda: torch.nn.grad.conv2d_input((B, N, Ho, Wo), w, dy, ss, (po_h, po_w), dil)
    return out


def stock_wgrad(a):
    B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo, st, ph, pw, dil = a
    x = rnd(B, C, H, W).contiguous(memory_format=CL)
    dy = rnd(B, N, Ho, Wo).contiguous(memory_format=CL)
    out = {"miopen": lambda: torch.nn.grad.conv2d_weight(x, (N, C, R, S), dy, st, (ph, pw), dil)}
    if R == 1 and S == 1 and st == 1 and ph == 0 and pw == 0:
        x2, d2 = rnd(B * H * W, C), rnd(B * H * W, N)
        out["hipblaslt"] = lambda: torch.mm(d2.t(), x2)
    return out


def flops(name, a):
    if name == "mrfp_conv_wgrad":
        B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo = a[:11]
        return 2.0 * B * Ho * Wo * N * R * S * Ctrue
    B, H, W, C, N, ldy, R, S, Ho, Wo, st, ph, pw, dil, ss = a
    return 2.0 * B * Ho * Wo * N * R * S * C / float(ss * ss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_conv_shapes.json"))
    ap.add_argument("--out", default="gpurun_out/vs_stock")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--only", default=None)
    ap.add_argument("--find", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen find mode)")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(args.find)
    seen, shapes = set(), []
    for name, a in json.load(open(args.shapes)):
        k = (name, tuple(a))
        if k not in seen:
            seen.add(k)
            shapes.append(k)
    rows = []
    t00 = time.time()
    for i, (name, a) in enumerate(shapes):
        kind = "wgrad" if name == "mrfp_conv_wgrad" else "fwd"
        if args.only and args.only != kind:
            continue
        row = {"name": name, "args": list(a), "gflop": flops(name, a) / 1e9}
        try:
            row["ours_us"] = timed(ours_wgrad(a) if kind == "wgrad" else ours_fwd(a), args.reps, args.sets)
        except Exception as e:          # noqa: BLE001
            row["ours_err"] = str(e)[:200]
        for lab, fn in (stock_wgrad(a) if kind == "wgrad" else stock_fwd(a)).items():
            try:
                row[lab + "_us"] = timed(fn, args.reps, args.sets)
            except Exception as e:      # noqa: BLE001
                row[lab + "_err"] = str(e)[:200]
        rows.append(row)
        torch.cuda.empty_cache()
        with open(args.out + ".json", "w") as f:
            json.dump(rows, f)
        print("[%3d/%d %5.0fs] %s %s ours %.1f miopen %.1f mm %.1f" % (i + 1, len(shapes), time.time() - t00, kind, list(a),
              row.get("ours_us", -1), row.get("miopen_us", -1), row.get("hipblaslt_us", -1)), flush=True)
    write_md(rows, args.out + ".md")


def write_md(rows, path):
    lines = ["| launch | args | GFLOP | ours us | MIOpen us | hipBLASLt us | ours / best stock | ours TFLOP/s | stock TFLOP/s |", "|---|---|---|---|---|---|---|---|---|"]
    so = ss = 0.0
    for r in sorted(rows, key=lambda r: -r.get("ours_us", 0)):
        st = [r[k] for k in ("miopen_us", "hipblaslt_us") if k in r]
        best = min(st) if st else None
        o = r.get("ours_us")
        if o and best:
            so += o
            ss += best
        lines.append("| %s | %s | %.1f | %s | %s | %s | %s | %s | %s |" % (
            r["name"].replace("mrfp_conv_", ""), " ".join(map(str, r["args"])), r["gflop"],
            "%.1f" % o if o else "-", "%.1f" % r["miopen_us"] if "miopen_us" in r else r.get("miopen_err", "-")[:40],
            "%.1f" % r["hipblaslt_us"] if "hipblaslt_us" in r else "-",
            "%.2f" % (o / best) if o and best else "-",
            "%.0f" % (r["gflop"] / o * 1e3) if o else "-", "%.0f" % (r["gflop"] / best * 1e3) if best else "-"))
    lines.append("")
    lines.append("sum over the distinct shapes (each counted once): ours %.1f us, best stock %.1f us" % (so, ss))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
