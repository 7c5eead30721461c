# kernel statistics of the bench step with the multi-GPU machinery forced on at one rank (bucket hooks, side stream, RCCL calls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MRFP_FORCE_SYNC=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sync -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_sync.log 2>&1
tail -1 $R/gpurun_out/prof_sync.log | cut -c95-140
grep -i "nccl\|rccl\|copyBuffer\|Generic" $R/gpurun_out/prof_sync/bench_kernel_stats.csv | cut -c1-160
