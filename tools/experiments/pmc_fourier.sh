#!/bin/bash
# PMC counters of the band-limited Fourier kernels (three rocprofv3 --pmc passes, one counter group each, no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for g in a b c; do
  case $g in
    a) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";;
    b) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES";;
    c) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM";;
  esac
  rm -rf $R/gpurun_out/pmcf_$g
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcf_$g -o p -- python3 $R/tools/fourier_micro.py 16 128 192 192 3 > $R/gpurun_out/pmcf_$g.log 2>&1 || { tail -3 $R/gpurun_out/pmcf_$g.log; exit 1; }
done
cd $R && python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for g in "abc":
    for f in glob.glob("gpurun_out/pmcf_%s/**/*counter_collection.csv" % g, recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "band" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
for k, cs in acc.items():
    if "float" in k and "bfloat" not in k: continue
    m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}
    print(k[:70], " ".join("%s=%.3g" % (c.replace("SQ_", ""), m[c]) for c in sorted(m)))
PY
