#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/measure_traffic.sh into
  gpurun_out/traffic_r02/r02_bench_kernel_stats.{csv,md}   per-kernel device time of the bench command
  gpurun_out/traffic_r02/r02_traffic.json                   HBM bytes per train step by kernel family (PMC)
(copy them into profiles/ to have them judged / read by bench.py).

Corrections, exactly as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950
FETCH_SIZE counts a wide coalesced read at half its bytes -> x2; WRITE_SIZE is exact for 16-byte-per-lane stores.
usage: traffic_report.py OUTDIR STEPS_IN_STATS_TRACE STEPS_IN_PMC_TRACE
"""
import csv
import glob
import json
import os
import subprocess
import sys

FAMILIES = (("conv", ("conv_igemm_kernel", "conv_wgrad_kernel", "wgrad_reduce_kernel", "conv1x1_bstat_kernel", "compact_stats_kernel")),
            ("normalisation", ("stats_kernel", "affine_fwd_kernel", "affine_bwd_kernel", "finalize_kernel", "copy_channels_kernel")),
            )


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def main(out, steps_stats, steps_pmc):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = os.environ.get("MRFP_COMMIT") or \
        subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    # ---- kernel stats
    path = sorted(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam_ms = {}
    for r in rows:
        fam_ms[family(r["Name"])] = fam_ms.get(family(r["Name"]), 0.0) + float(r["TotalDurationNs"]) / 1e6 / steps_stats
    with open(out + "/r02_bench_kernel_stats.csv", "w") as f:
        f.write(open(path).read())
    with open(out + "/r02_bench_kernel_stats.md", "w") as f:
        f.write("# Round 2 -- `MRFP_WGRAD_STREAM=0 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline` under rocprofv3 "
                "--kernel-trace --stats (ResNet-101 MRFP+ 16x768x768 bf16, one MI355X, commit %s)\n\n" % commit)
        f.write("%d train steps in the trace (warm-up, timed, and the per-launch timing step of bench.py); total GPU kernel "
                "time %.2f ms = %.2f ms/step.  By family (ms/step): %s\n\n" % (
                    steps_stats, tot / 1e6, tot / 1e6 / steps_stats, ", ".join("%s %.2f" % kv for kv in sorted(fam_ms.items()))))
        f.write("| kernel | calls | total ms | avg us | % | ms/step |\n|---|---|---|---|---|---|\n")
        for r in rows[:45]:
            t = float(r["TotalDurationNs"])
            f.write("| `%s` | %s | %.2f | %.1f | %.1f | %.2f |\n" % (r["Name"][:110].replace("|", "/"), r["Calls"], t / 1e6,
                                                                   float(r["AverageNs"]) / 1e3, 100 * t / tot, t / 1e6 / steps_stats))
    # ---- PMC passes
    res = {}
    for tag, counter, corr in (("read", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        p = sorted(glob.glob(out + ("/fetch" if tag == "read" else "/write") + "/**/*counter_collection.csv", recursive=True))[-1]
        per = {}
        with open(p) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                per[family(r["Kernel_Name"])] = per.get(family(r["Kernel_Name"]), 0.0) + float(r["Counter_Value"]) * 1024.0 * corr
        res[tag] = {k: v / steps_pmc for k, v in per.items()}
    fams = sorted(set(res["read"]) | set(res["write"]))
    traffic = {"commit": commit, "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/measure_traffic.sh; "
               "KB -> bytes, FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md)",
               "workload": {"trunk": "resnet-101", "size": 768, "width": 768, "batch": 16, "dtype": "bf16"},
               "steps_in_pmc_trace": steps_pmc,
               "hbm_bytes_per_step": {k: {"read": round(res["read"].get(k, 0.0)), "write": round(res["write"].get(k, 0.0))} for k in fams},
               "kernel_ms_per_step": {k: round(v, 3) for k, v in fam_ms.items()}}
    conv = traffic["hbm_bytes_per_step"].get("conv", {"read": 0, "write": 0})
    traffic["conv_family_hbm_bytes_per_step"] = conv["read"] + conv["write"]
    # algorithmic bytes of the conv family: every launch reads its input + weights and writes its output once
    # (+ the skip-gradient addend, + the split-K slabs written and re-read) -- taken from a --dump-convs file when present
    dump = os.path.join(out, "convs.json")
    if os.path.exists(dump):
        alg = 0.0
        for e in json.load(open(dump)):
            a = e["args"]
            if e["name"] == "mrfp_conv_fwd":
                B, H, W, C, N, ldy, R, S, Ho, Wo = a[:10]
                alg += 2.0 * (B * H * W * C + B * Ho * Wo * N + N * R * S * C)
            elif e["name"] == "mrfp_conv_wgrad":
                B, H, W, C, Ct, N, ldn, R, S, Ho, Wo = a[:11]
                alg += 2.0 * (B * H * W * C + B * Ho * Wo * N) + 4.0 * N * R * S * C
        traffic["conv_family_algorithmic_bytes_per_step"] = round(alg)
    json.dump(traffic, open(out + "/r02_traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
