#!/bin/bash
# A/B of the K-tile staging of the forward / dgrad kernel on the bench workload's main shapes (one GPU box, same run):
# default (one LDS buffer, builtin DMA) vs MRFP_CONV_NBUF=2 / 3 (asynchronous LDS-DMA ring) vs the 256x256 8-wave ring.
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/ab_nbuf.log}
: > $out
for shape in big3x3 hrfp hrfp128 l3_3x3 l3_1x1 l3_exp exp1x1; do
  for mode in "" "MRFP_CONV_NBUF=2" "MRFP_CONV_BIGTILE=1" "MRFP_CONV_T192=0 MRFP_CONV_T96=0 MRFP_CONV_NBUF=2" "MRFP_CONV_T192=0 MRFP_CONV_T96=0" "MRFP_CONV_T96=0" "MRFP_CONV_T192=2"; do
    echo -n "[$mode] " >> $out
    env $mode python tools/conv_micro.py $shape 30 fwd 2>/dev/null | tail -1 >> $out
  done
done
cat $out
