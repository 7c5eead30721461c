"""Distribution of the bf16-fidelity yardsticks over seeds: tests/test_bf16_fidelity_gpu.py compares the
bf16 training trajectory with an fp32 run and with a YARDSTICK run (fp32 arithmetic from weights that carry one bf16 rounding); the
yardstick is one sample of a chaotic quantity, so rounds 3-4 re-based the bar on the sample of the same test run.  This script
measures both quantities over several seeds (weights, batch, NP+ noise all re-drawn per seed) ONCE; the distribution is committed
(profiles/r05_bf16_yardstick.json) and the test asserts fixed constants taken from it.

    python tools/bf16_yardstick_seeds.py [--seeds 6] [--out gpurun_out/bf16_yardstick.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--out", default="gpurun_out/bf16_yardstick.json")
    args = ap.parse_args()
    import test_bf16_fidelity_gpu as T          # (the run function of the test itself: same model, batch recipe, schedule)
    rows = []
    t0 = time.time()
    mean = lambda v: sum(v) / len(v)
    for seed in range(args.seeds):
        l32, m32, g32 = T._run(torch.float32, seed=seed)
        l16, m16, g16 = T._run(torch.bfloat16, seed=seed)
        lrw, mrw, grw = T._run(torch.float32, round_weights=True, seed=seed)
        rel = [abs(a - b) / abs(a) for a, b in zip(l32, l16)][5:]
        rel_rw = [abs(a - b) / abs(a) for a, b in zip(l32, lrw)][5:]
        cos, cos_rw = T._cos(g32, g16), T._cos(g32, grw)
        row = {"seed": seed, "loss_first_last": {"fp32": [l32[0], l32[-1]], "bf16": [l16[0], l16[-1]], "rounded": [lrw[0], lrw[-1]]},
               "dev_bf16_mean": mean(rel), "dev_bf16_max": max(rel), "dev_rounded_mean": mean(rel_rw), "dev_rounded_max": max(rel_rw),
               "miou": {"fp32": m32, "bf16": m16, "rounded": mrw},
               "final_loss_rel_bf16": abs(l16[-1] - l32[-1]) / l32[-1], "final_loss_rel_rounded": abs(lrw[-1] - l32[-1]) / l32[-1],
               "cos_bf16": cos, "cos_rounded": cos_rw}
        rows.append(row)
        print("[%5.0fs] seed %d: bf16 dev mean %.4f max %.4f | rounded-weights dev mean %.4f max %.4f | mIoU %.3f / %.3f / %.3f | cos(final1, layer3) bf16 %.3f %.3f"
              % (time.time() - t0, seed, row["dev_bf16_mean"], row["dev_bf16_max"], row["dev_rounded_mean"], row["dev_rounded_max"],
                 m32, m16, mrw, cos["final1"], cos["layer3"]), flush=True)
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"rows": rows, "what": __doc__.strip().split("\n")[0]}, f, indent=1)
    summ = {k: {"min": min(r[k] for r in rows), "max": max(r[k] for r in rows), "mean": mean([r[k] for r in rows])}
            for k in ("dev_bf16_mean", "dev_bf16_max", "dev_rounded_mean", "dev_rounded_max", "final_loss_rel_bf16", "final_loss_rel_rounded")}
    summ["miou_abs_diff_bf16"] = {"max": max(abs(r["miou"]["fp32"] - r["miou"]["bf16"]) for r in rows)}
    summ["miou_abs_diff_rounded"] = {"max": max(abs(r["miou"]["fp32"] - r["miou"]["rounded"]) for r in rows)}
    with open(args.out, "w") as f:
        json.dump({"rows": rows, "summary": summ, "what": __doc__.strip().split("\n")[0]}, f, indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
