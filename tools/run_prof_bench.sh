# rocprofv3 kernel statistics of the bench command (weight gradients on the main stream so that kernel durations are
# not inflated by overlap); summary -> profiles/ via tools/summarize_prof.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MRFP_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1u -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r1u.log 2>&1
tail -1 $R/gpurun_out/prof_r1u.log | cut -c1-300
find $R/gpurun_out/prof_r1u -name "*kernel_stats.csv" | head
