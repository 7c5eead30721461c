"""Per-kernel means of the counters collected by tools/run_pmc.sh SHAPE (three rocprofv3 --pmc passes):
python tools/pmc_report.py SHAPE  ->  markdown table rows on stdout."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
shape = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for grp in "abc":
    for f in glob.glob(os.path.join(root, "pmc_%s_%s" % (grp, shape), "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv" not in k:
                continue
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
for k, cs in acc.items():
    print("### `%s` (%d dispatches)" % (k[:110], max(len(v) for v in cs.values())))
    m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}      # first dispatch = warm-up
    for c in sorted(m):
        print("| `%s` | %.4g |" % (c, m[c]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print("| %s / SQ_WAVE_CYCLES | %.1f %% |" % (c, 100 * m[c] / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print("| SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES | %.3f |" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]))
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        print("| L2 hit rate | %.1f %% |" % (100 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
    print()
