#!/bin/bash
# HBM traffic and kernel times of the bench workload (run on the GPU box: `gpurun -- bash tools/measure_traffic.sh`).
#   pass 1: rocprofv3 --kernel-trace --stats            -> per-kernel device time         (profiles/rNN_bench_kernel_stats.{csv,md})
#   pass 2: rocprofv3 --kernel-trace --pmc FETCH_SIZE   -> bytes read from the fabric      } each in its OWN run, no trace domains
#   pass 3: rocprofv3 --kernel-trace --pmc WRITE_SIZE   -> bytes written                   } mixed in (MI355X_MICROARCH.md, HBM)
# tools/traffic_report.py folds passes 2+3 per kernel family and writes rNN_traffic.json (copy to profiles/), which bench.py reads
# for `roofline.traffic`.  The program after `--` is python3 itself (no env / shell hop).  MRFP_COMMIT=<hash> labels the result
# (the GPU box has no .git): `gpurun -- "MRFP_COMMIT=$(git rev-parse --short HEAD) bash tools/measure_traffic.sh"`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${MRFP_ROUND:-r05}; export MRFP_ROUND=$TAG
O=$R/gpurun_out/traffic_$TAG
rm -rf $O && mkdir -p $O
STEPS=3; WARM=2
export MRFP_WGRAD_STREAM=0      # weight gradients on the main stream: kernel durations are not inflated by overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --dump-convs $O/convs.json > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
tail -1 $O/bench.log > $O/bench_line.json
cd $R && python3 tools/traffic_report.py $O $((STEPS + WARM + 1)) 3
