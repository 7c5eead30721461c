cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in l3_3x3 l3_1x1 big3x3; do
for d in 0 1 2 3; do
MRFP_DEBUG_DROP=$d rocprofv3 --kernel-trace --stats -d $R/gpurun_out/micw_${s}_$d -o m --output-format csv -- python3 $R/tools/conv_micro.py $s 10 wgrad > $R/gpurun_out/micw_${s}_$d.log 2>&1 || exit 1
done; done
