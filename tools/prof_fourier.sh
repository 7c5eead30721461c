#!/bin/bash
# per-kernel device time of the Fourier amplitude mix (tools/fourier_micro.py B C H W) under rocprofv3 --kernel-trace --stats
#   gpurun -- 'bash tools/prof_fourier.sh 16 128 192 192'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/fftp; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o f -- python3 $R/tools/fourier_micro.py "$@" > $O.log 2>&1
cd $R && grep '"low"' $O.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/fftp/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "band" in r["Name"] or "fft" in r["Name"] or "dft" in r["Name"]:
        print("%-90s calls %s avg %.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
