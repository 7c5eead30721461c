#!/bin/bash
# PMC triple (MFMA busy, wait shares, L2) of the weight-gradient kernel on the grouped layer-3 launches and the 192^2 3x3 layer:
#   gpurun -- 'bash tools/run_pmc_wgrad.sh'   -> gpurun_out/pmcw_*.txt   (each counter group in its own rocprofv3 pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for idx in ${WG_IDX:-0 2 6}; do
  O=$R/gpurun_out/pmcw_$idx; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o p -- python3 $R/tools/wgrad_micro.py 3 $idx grouped > $O/a.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -o p -- python3 $R/tools/wgrad_micro.py 3 $idx grouped > $O/b.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_WAIT_INST_LDS --output-format csv -d $O/c -o p -- python3 $R/tools/wgrad_micro.py 3 $idx grouped > $O/c.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python3 $R/tools/wgrad_micro.py 3 $idx grouped > $O/f.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- python3 $R/tools/wgrad_micro.py 3 $idx grouped > $O/w.log 2>&1 || exit 1
  python3 - $O $idx <<'PY'
import csv, glob, sys
from collections import defaultdict
O, idx = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for grp in "abcfw":
    for f in glob.glob(O + "/" + grp + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wgrad" not in k and "wg3" not in k: continue
            per[(k.split("(")[0][-70:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items(): acc[k][c].append(v)
print("== shape %s (%s)" % (idx, open(O + "/a.log").read().strip().splitlines()[-1][:150]))
for k, cs in acc.items():
    m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}
    line = "  %-70s" % k
    wc = m.get("SQ_WAVE_CYCLES")
    if wc: line += " wait_any %.1f%% wait_inst %.1f%% active %.1f%%" % (100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: line += " mfma_busy %.3f bank_conflict %.3g" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], m.get("SQ_LDS_BANK_CONFLICT", 0))
    if "TCC_HIT_sum" in m: line += " L2 hit %.1f%% TCC_REQ %.3g" % (100 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), m["TCC_REQ_sum"])
    if "FETCH_SIZE" in m: line += " read %.1f MB" % (m["FETCH_SIZE"] * 1024 * 2 / 1e6)
    if "WRITE_SIZE" in m: line += " write %.1f MB" % (m["WRITE_SIZE"] * 1024 / 1e6)
    print(line)
PY
done
