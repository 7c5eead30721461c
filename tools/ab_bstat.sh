#!/bin/bash
# A/B of the B-stationary 1x1 kernel against the generic kernel (one GPU box, same run)
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/ab_bstat.log}
: > $out
for shape in l3_exp l2_exp exp1x1; do
  for mode in "MRFP_CONV_BSTAT=0" "MRFP_CONV_BSTAT=1"; do
    echo -n "[$mode] " >> $out
    env $mode python tools/conv_micro.py $shape 50 fwd 2>/dev/null | tail -1 >> $out
  done
done
cat $out
