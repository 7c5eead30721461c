# per-shape weight-gradient times under fixed workgroup targets (split-K counts) next to the built-in cost model: gpurun -- bash tools/wgrad_sweep.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --dump-convs gpurun_out/wsweep_$tag.json > /dev/null 2>&1; echo "$tag done"; }
run default MRFP_X=0
run default2 MRFP_X=0
for n in 256 384 512 768 1024 1536 2048; do run wgs$n MRFP_WGRAD_WGS=$n; done
