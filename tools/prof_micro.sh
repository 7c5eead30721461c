#!/bin/bash
# per-kernel device time of one conv_micro shape under several switches (rocprofv3 --kernel-trace --stats)
# usage: tools/prof_micro.sh SHAPE "ENV1=.. ENV2=.." "ENV=.." ...
cd "$(dirname "$0")/.."
shape=$1; shift
export TMPDIR=/tmp
i=0
for mode in "$@"; do
  i=$((i+1))
  d=gpurun_out/pm_${shape}_$i
  rm -rf $d
  for kv in $mode; do export $kv; done
  rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python3 tools/conv_micro.py $shape 30 ${MICRO_MODE:-fwd} > /dev/null 2>&1
  for kv in $mode; do unset ${kv%%=*}; done
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $shape [$mode]"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if "conv" in n or "wgrad" in n:
        print("   %-70s calls %s avg %.1f us"%(n[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
