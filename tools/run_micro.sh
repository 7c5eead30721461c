cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in l3_3x3 l3_1x1 l3_exp big3x3 hrfp128; do
for v in "MRFP_CONV_DMA=3" "MRFP_CONV_DMA=5"; do
echo "== $s $v"; env $v python3 $R/tools/conv_micro.py $s 30 fwd 2>&1 | tail -1
done; done
