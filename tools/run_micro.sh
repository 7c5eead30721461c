cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in exp1x1 l3_exp; do
for d in 0 1 3; do
MRFP_DEBUG_DROP=$d rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mic_${s}_$d -o m --output-format csv -- python3 $R/tools/conv_micro.py $s 10 fwd > $R/gpurun_out/mic_${s}_$d.log 2>&1 || exit 1
done; done
