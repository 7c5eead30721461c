#!/bin/bash
# average device time of the kernels matching PATTERN in the bench step under several settings (rocprofv3 --kernel-trace --stats)
# usage: tools/prof_bench_kernels.sh PATTERN "ENV=.." "ENV=.." ...
cd "$(dirname "$0")/.."
R=$(pwd)
pat=$1; shift
export TMPDIR=/tmp
i=0
for mode in "$@"; do
  i=$((i+1))
  d=$R/gpurun_out/pbk_$i
  rm -rf $d
  for kv in $mode; do export $kv; done
  (cd /tmp && MRFP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $d -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
  for kv in $mode; do unset ${kv%%=*}; done
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== [$mode]"
  python3 - "$f" "$pat" <<'PY'
import csv,sys,re
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    tot+=float(r['TotalDurationNs'])
    if re.search(sys.argv[2], r['Name']):
        print("   %-64s calls %5s avg %7.2f us total %7.2f ms"%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
print("   all kernels: %.2f ms"%(tot/1e6))
PY
done
