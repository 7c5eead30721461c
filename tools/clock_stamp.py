"""In-kernel clock of the convolution kernels under their own load (MI355X_MICROARCH.md, DVFS give-back item 6): a DIAGNOSTIC build of
the library (-DMRFP_CLOCK_STAMP=1 -> csrc/libmrfp_hip_clk.so, built here; in the product build no stamp executes) in which every
convolution workgroup records d(s_memtime) / d(s_memrealtime) around its main loop.  Per shape: >= 2 s of back-to-back launches on
random data, then the stamps of the last launch; clock = median over workgroups of cycles / ticks x 100 MHz.

    python tools/clock_stamp.py [--out gpurun_out/clock.json] [--seconds 2.0]

Run it with MRFP_HIP_LIB unset: it builds the variant and re-runs itself in a child process with MRFP_HIP_LIB pointing at it."""
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

# name: (kind, launch arguments as in tools/bench_conv_shapes.json, family for mrfp_debug_clock_stamps)
SHAPES = {
    "rr192 3x3 256->256 @192^2 (decoder, row-reuse 192x128)": ("fwd", [16, 192, 192, 256, 256, 256, 3, 3, 192, 192, 1, 1, 1, 1, 1], 0),
    "rr192 3x3 128->256 @384^2 (HRFP)": ("fwd", [16, 384, 384, 128, 256, 256, 3, 3, 384, 384, 1, 1, 1, 1, 1], 0),
    "96x128 3x3 256->256 @48^2 (layer3 conv2)": ("fwd", [16, 48, 48, 256, 256, 256, 3, 3, 48, 48, 1, 1, 1, 1, 1], 0),
    "192x128 3x3 2048->256 d12 @48^2 (ASPP)": ("fwd", [16, 48, 48, 2048, 256, 256, 3, 3, 48, 48, 1, 12, 12, 12, 1], 0),
    "1x1 1024->256 @48^2 (layer3 conv1)": ("fwd", [16, 48, 48, 1024, 256, 256, 1, 1, 48, 48, 1, 0, 0, 1, 1], 0),
    "pointwise 256->1024 @48^2 (layer3 conv3, B-stationary)": ("fwd", [16, 48, 48, 256, 1024, 1024, 1, 1, 48, 48, 1, 0, 0, 1, 1], 1),
    "pointwise 64->256 @192^2 (layer1 conv3, B-stationary)": ("fwd", [16, 192, 192, 64, 256, 256, 1, 1, 192, 192, 1, 0, 0, 1, 1], 1),
    "wgrad 3x3 256->256 @192^2": ("wgrad", [16, 192, 192, 256, 256, 256, 256, 3, 3, 192, 192, 1, 1, 1, 1], 2),
    "wgrad 1x1 1024->256 @48^2": ("wgrad", [16, 48, 48, 1024, 1024, 256, 256, 1, 1, 48, 48, 1, 0, 0, 1], 2),
}


def child(args):
    import torch
    from mrfp_amd import _lib
    from tools.stock_shapes import ours_fwd, ours_wgrad
    L = _lib.lib()
    rows = {}
    for name, (kind, a, fam) in SHAPES.items():
        fn = ours_wgrad(a) if kind == "wgrad" else ours_fwd(a)
        fn()
        torch.cuda.synchronize()
        t0, n = time.time(), 0
        while time.time() - t0 < args.seconds:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(20):
            fn()
        a1.record()
        torch.cuda.synchronize()
        us = a0.elapsed_time(a1) / 20 * 1e3
        buf = (ctypes.c_uint64 * (2 * 4096))()
        _lib.call("mrfp_debug_clock_stamps", fam, ctypes.cast(buf, ctypes.c_void_p), 4096)
        ghz = sorted(buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(4096) if buf[2 * i + 1] > 50)
        # (slots of workgroups that never ran in the LAST launch keep older stamps of the same family: same kernel here)
        if not ghz:
            rows[name] = {"us": us, "clock_ghz": None}
            continue
        rows[name] = {"us_stamped_build": round(us, 1), "launches": n, "workgroups": len(ghz), "clock_ghz": round(statistics.median(ghz), 3),
                      "p10": round(ghz[len(ghz) // 10], 3), "p90": round(ghz[len(ghz) * 9 // 10], 3)}
        print("%-60s %7.1f us  clock %.3f GHz (p10 %.3f p90 %.3f, %d workgroups)" % (name, us, rows[name]["clock_ghz"], rows[name]["p10"],
              rows[name]["p90"], len(ghz)), flush=True)
    clocks = [r["clock_ghz"] for r in rows.values() if r.get("clock_ghz")]
    out = {"method": "d(s_memtime)/d(s_memrealtime) x 100 MHz around the main loop, median over workgroups, after %.1f s of back-to-back "
                     "launches on random data (diagnostic build -DMRFP_CLOCK_STAMP=1)" % args.seconds,
           "shapes": rows, "clock_ghz_median": round(statistics.median(clocks), 3) if clocks else None,
           "clock_ghz_min": min(clocks) if clocks else None, "clock_ghz_max": max(clocks) if clocks else None}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "shapes"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/clock.json")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    from mrfp_amd import build
    lib = build.build_variant("clk", ("conv_igemm", "conv_pw", "conv_wgrad"), ["-DMRFP_CLOCK_STAMP=1"])
    env = dict(os.environ, MRFP_HIP_LIB=lib)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child", "--out", args.out, "--seconds", str(args.seconds)],
                             env=env, cwd=ROOT))


if __name__ == "__main__":
    main()
