# A/B of the workgroup-wide epilogue staging: default build against `tools/build_variant.sh narrow conv -DMRFP_WIDE_EP=0`
R=$GRAFT_REPO_ROOT; V=$R/mrfp_amd/csrc/libmrfp_hip_narrow.so
cd $R && MRFP_HIP_LIB=$V python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
cd /tmp
for s in l3_exp l3_1x1 l3_3x3 big3x3 exp1x1; do for rep in 1 2; do
echo "== $s base"; python3 $R/tools/conv_micro.py $s 50 fwd 2>&1 | tail -1
echo "== $s narrow"; MRFP_HIP_LIB=$V python3 $R/tools/conv_micro.py $s 50 fwd 2>&1 | tail -1
done; done
bash $R/tools/ab_lib.sh mrfp_amd/csrc/libmrfp_hip_narrow.so
