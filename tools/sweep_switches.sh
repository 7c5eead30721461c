cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['conv_ms_per_step'])"; }
run "X=0"
run "MRFP_CONV_T192=0"
run "MRFP_CONV_BIGTILE=1"
run "MRFP_CONV_T96=0"
run "X=0"
run "MRFP_WGRAD_WGS=384"
run "MRFP_WGRAD_WGS=768"
run "MRFP_WGRAD_DMA=0"
run "X=0"
