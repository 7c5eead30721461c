#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/measure_traffic.sh into (TAG = $MRFP_ROUND, default r03)
  gpurun_out/traffic_TAG/TAG_bench_kernel_stats.{csv,md}   per-kernel device time of the bench command
  gpurun_out/traffic_TAG/TAG_traffic.json                   HBM bytes per train step by kernel family (PMC), and the conv
                                                            launches of one step classed by their OWN bound (by_class)
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

TAG = os.environ.get("MRFP_ROUND", "r05")
# (round 5, VERDICT r4 weak 10: the fused stem pool's three kernels -- maxpool_fwd with the affine, pool_norm_bwd<0/1> -- are booked in the
#  normalisation family HERE as in bench.py's NORM_CALLS; round 4's summary had them under "other")
FAMILIES = (("conv", ("conv_igemm_kernel", "conv_wgrad_kernel", "conv_wg3_kernel", "conv_wg1_kernel", "wgrad_reduce_kernel", "conv1x1_bstat_kernel", "conv1x1_longk",
                      "conv3x3_c64_kernel", "conv_pw", "compact_stats_kernel")),
            ("normalisation", ("stats_kernel", "affine_fwd_kernel", "affine_fwd_stats_kernel", "affine_bwd_kernel", "finalize_kernel",
                               "copy_channels_kernel", "pool_norm_bwd_kernel", "maxpool_fwd_kernel")),
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
    with open(out + "/%s_bench_kernel_stats.csv" % TAG, "w") as f:
        f.write(open(path).read())
    with open(out + "/%s_bench_kernel_stats.md" % TAG, "w") as f:
        f.write("# " + TAG + " -- `MRFP_WGRAD_STREAM=0 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline` under rocprofv3 "
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
    sys.path.insert(0, root)
    from mrfp_amd import _lib
    traffic = {"commit": commit, "source_sha16": _lib.source_hash(), "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/measure_traffic.sh; "
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
        # algorithmic bytes of the conv family (every launch reads its input + weights (+ addend) and writes its output once)
        # and the per-class breakdown bench.py prints as roofline.by_class: each launch against max(FLOP / 2.5 PF, bytes / 8 TB/s)
        alg, cls = 0.0, {}
        for e in json.load(open(dump)):
            by, fl = e["mbytes"] * 1e6, e["gflop"] * 1e9
            alg += by
            t_f, t_b = fl / 2.5e15, by / 8.0e12
            k = ("mfma_bound" if t_f >= t_b else "hbm_bound") + ("_wgrad" if e["name"] == "mrfp_conv_wgrad" else "_fwd_dgrad")
            g = cls.setdefault(k, {"launches": 0, "ms": 0.0, "tflop": 0.0, "gbytes": 0.0, "bound_ms": 0.0})
            g["launches"] += 1
            g["ms"] += e["ms"]
            g["tflop"] += fl / 1e12
            g["gbytes"] += by / 1e9
            g["bound_ms"] += 1e3 * max(t_f, t_b)
        for g in cls.values():
            g["frac_of_own_bound"] = round(g["bound_ms"] / g["ms"], 4)
            for k in ("ms", "tflop", "gbytes", "bound_ms"):
                g[k] = round(g[k], 3)
        traffic["conv_family_algorithmic_bytes_per_step"] = round(alg)
        traffic["conv_by_class"] = cls
    json.dump(traffic, open(out + "/%s_traffic.json" % TAG, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
