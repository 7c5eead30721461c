"""Where does a workgroup of the two-group long-K pointwise kernel (conv1x1_longk2_kernel: 1024 -> 256 at 48^2, the layer-3 reduce convolution
and the dgrad of the expand one, 45 launches per step) spend its cycles?  DESIGN.md section 9 item 2 asked for in-kernel stamps of prologue /
steady state / tail.  DIAGNOSTIC build (-DMRFP_CLOCK_STAMP=1 -> csrc/libmrfp_hip_clk.so; in the product build no stamp executes): wave 0 (K
group 0) and wave 4 (K group 1) of every workgroup add up s_memtime cycles per phase (csrc/conv_pwk.hip: PhaseClock).

    python tools/phase_stamp.py [--out gpurun_out/phase_pwk.json] [--seconds 1.0]
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["prologue (first transfers issued, weights -> registers)", "wait + barrier (18 steps)", "issue of 4 transfers (18 steps)",
          "fragment reads + 32 MFMAs (18 steps)", "exchange + convert + store (9 tiles)", "whole kernel (cycles)", "whole kernel (100 MHz ticks)"]
SHAPES = {"fwd 1024->256 @48^2 with statistics": ("stats", [16, 48, 48, 1024, 256, 256, 1, 1, 48, 48, 1, 0, 0, 1, 1]),
          "dgrad form 1024->256 @48^2 (no epilogue extras)": ("plain", [16, 48, 48, 1024, 256, 256, 1, 1, 48, 48, 1, 0, 0, 1, 1])}


def child(args):
    import torch
    from mrfp_amd import _lib, conv
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE = torch.bfloat16
    out = {}
    for name, (kind, a) in SHAPES.items():
        B, H, W, C, N = a[0], a[1], a[2], a[3], a[4]
        x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, C, 1, 1, device="cuda") * 0.05)

        conv.FUSE_STATS[0] = kind == "stats"          # (a bias-free forward convolution asks its epilogue for BatchNorm statistics)

        def fn():
            with torch.no_grad():
                return conv.conv2d(x, w, None, 1, 0, 1)
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < args.seconds:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(20):
            fn()
        a1.record()
        torch.cuda.synchronize()
        us = a0.elapsed_time(a1) / 20 * 1e3
        buf = (ctypes.c_uint64 * (2 * 4096))()
        _lib.call("mrfp_debug_clock_stamps", 3, ctypes.cast(buf, ctypes.c_void_p), 4096)
        rows = {}
        for g in (0, 1):
            per = [[buf[2 * (((b * 2 + g) * 8) + k)] for k in range(7)] for b in range(256)]
            per = [p for p in per if p[6] > 0]
            if not per:
                continue
            ghz = statistics.median(p[5] / p[6] * 0.1 for p in per)
            med = [statistics.median(p[k] for p in per) for k in range(6)]
            rows["wave %d (K group %d)" % (4 * g, g)] = {
                "clock_ghz": round(ghz, 3), "workgroups": len(per),
                "us": {PHASES[k]: round(med[k] / ghz / 1e3, 2) for k in range(6)},
                "unaccounted_us": round((med[5] - sum(med[:5])) / ghz / 1e3, 2)}
        out[name] = {"launch_us_stamped_build": round(us, 1), "waves": rows}
        print(name, "launch %.1f us (stamped build)" % us, flush=True)
        for wv, r in rows.items():
            print("   %s  clock %.3f GHz" % (wv, r["clock_ghz"]))
            for k, v in r["us"].items():
                print("      %-62s %6.2f us" % (k, v))
            print("      %-62s %6.2f us" % ("(after the loop: statistics rows, exit)", r["unaccounted_us"]), flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/phase_pwk.json")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    from mrfp_amd import build
    lib = build.build_variant("clk", ("conv_igemm", "conv_pw", "conv_wgrad", "conv_pwk"), ["-DMRFP_CLOCK_STAMP=1"])
    env = dict(os.environ, MRFP_HIP_LIB=lib)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child", "--out", args.out, "--seconds", str(args.seconds)],
                             env=env, cwd=ROOT))


if __name__ == "__main__":
    main()
