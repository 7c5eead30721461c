#!/bin/bash
# same-box A/B of environment switches on the bench step: tools/ab_env.sh ROUNDS "ENV=.. ENV=.." "ENV=.." ...
# prints ms per step (and the conv / normalisation family times of the per-launch timing step) for every setting, alternated ROUNDS times
cd "$(dirname "$0")/.."
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for mode in "$@"; do
    i=$((i+1))
    env $mode python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --dump-convs gpurun_out/ab_convs_$i.json > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { echo "[$mode] FAILED"; tail -5 gpurun_out/ab_$i.err; continue; }
    python - "$mode" gpurun_out/ab_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print("[%s] %.2f ms/step  conv %.2f ms (%.0f TF/s, wgrad %.2f ms)  norm %.2f ms  other %.2f" % (sys.argv[1], d["ms_per_step"], r["conv_ms_per_step"], r["achieved"],
      sum(v["ms"] for k,v in r["by_class"].items() if "wgrad" in k), r["hbm"]["ms"], r["other_ms_per_step"]), flush=True)
PY
  done
done
