# A/B of a build variant of the library against the default build, alternating, on one box:
#   bash tools/ab_lib.sh mrfp_amd/csrc/libmrfp_hip_NAME.so [bench args]
V=$GRAFT_REPO_ROOT/$1; shift
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | cut -c1-140
MRFP_HIP_LIB=$V python bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | cut -c1-140
done
