# per-shape convolution times under every tile override (one box): gpurun -- bash tools/tile_sweep.sh ; then python tools/tile_sweep.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --dump-convs gpurun_out/sweep_$tag.json > /dev/null 2>&1; echo "$tag done"; }
run default MRFP_X=0
run default2 MRFP_X=0
run t96_0 MRFP_CONV_T96=0
run t96_2 MRFP_CONV_T96=2
run t192_0 MRFP_CONV_T192=0
run t192_2 MRFP_CONV_T192=2 MRFP_CONV_RR=0
run rr0 MRFP_CONV_RR=0
run rr3 MRFP_CONV_RR=3
run pw0 MRFP_CONV_PW=0
run t128 MRFP_CONV_T96=0 MRFP_CONV_T192=0 MRFP_CONV_RR=0
