#!/bin/bash
# A/B of the tile choice of the forward / dgrad kernel on the M = 36 864 shapes (one GPU box, same run)
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/ab_tiles.log}
: > $out
for shape in l3_3x3 l3_1x1 l3_exp big3x3; do
  for mode in "" "MRFP_CONV_T96=0" "MRFP_CONV_T96=0 MRFP_CONV_T192=2" "MRFP_CONV_BIGTILE=2" "MRFP_CONV_T96=2"; do
    echo -n "[$mode] " >> $out
    env $mode python tools/conv_micro.py $shape 40 fwd 2>/dev/null | tail -1 >> $out
  done
done
cat $out
