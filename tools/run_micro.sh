cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in l3_3x3 l3_1x1 l3_exp; do
for v in "MRFP_CONV_T192=1" "MRFP_CONV_T192=2" "MRFP_CONV_T96=0 MRFP_CONV_T192=0"; do
echo "== $s $v"; env $v python3 $R/tools/conv_micro.py $s 40 fwd 2>&1 | tail -1
done; done
