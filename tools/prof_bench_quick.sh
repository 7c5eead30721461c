#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run; prints the kernels matching $1 (regex, default: everything above 0.3 % of the time)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp MRFP_WGRAD_STREAM=0
O=$R/gpurun_out/pbq; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
cd $R && python3 - "$1" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else None
f = sorted(glob.glob("gpurun_out/pbq/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.2f" % (tot / 1e6 / 6))
for r in rows:
    t = float(r["TotalDurationNs"])
    if (pat and pat.search(r["Name"])) or (not pat and t / tot > 0.003):
        print("%-100s calls %5s avg %7.1f us  %.2f ms/step" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, t / 1e6 / 6))
PY
