cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MRFP_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f16 -o bench -- python3 $R/bench.py --dtype f16 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_f16.log 2>&1
head -25 $R/gpurun_out/prof_f16/bench_kernel_stats.csv | cut -c1-160
