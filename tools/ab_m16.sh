# A/B of the MFMA shape in the forward / dgrad kernel: default build (16x16x32) against `tools/build_variant.sh m32 -DMRFP_M16=0`
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$R/mrfp_amd/csrc/libmrfp_hip_m32.so
(cd $R && MRFP_HIP_LIB=$V python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3)
for s in l3_3x3 l3_1x1 l3_exp big3x3 hrfp128 exp1x1; do
for rep in 1 2; do
echo "== $s base"; python3 $R/tools/conv_micro.py $s 50 fwd 2>&1 | tail -1
echo "== $s m32"; MRFP_HIP_LIB=$V python3 $R/tools/conv_micro.py $s 50 fwd 2>&1 | tail -1
done; done
cd $R
python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
MRFP_HIP_LIB=$V python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
MRFP_HIP_LIB=$V python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
