cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "MRFP_CONV_T256X128=0" "MRFP_CONV_T256X128=1"; do
echo "== hrfp128 $v"; env $v python3 $R/tools/conv_micro.py hrfp128 20 fwd 2>&1 | tail -1
done
