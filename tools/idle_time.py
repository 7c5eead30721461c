"""GPU idle time inside a train step from a rocprofv3 kernel trace (all streams): union of the kernels' [start, end] intervals against the wall span.
    gpurun -- 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/idle -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline; python3 $GRAFT_REPO_ROOT/tools/idle_time.py'"""
import csv, glob, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "idle")
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"] if "Stream_Id" in r else "") for r in csv.DictReader(open(f))]
rows.sort()
# steps: delimited by the sgd kernel
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2]]
print("kernels", len(rows), "sgd launches", len(sg))
for a, b in zip(sg[2:-1], sg[3:]):            # steady-state steps (skip the first ones)
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n, st in seg:
        s = max(s, t0)
        if cur_e is None:
            cur_s, cur_e = s, e
            if s > t0: gaps.append((s - t0, n))
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    wall = t1 - t0
    big = sorted(gaps, reverse=True)[:5]
    print("step wall %.2f ms  busy %.2f ms  idle %.2f ms (%d gaps, mean %.2f us)  sum of kernel durations %.2f ms   largest gaps: %s"
          % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(gaps), (wall - busy) / max(1, len(gaps)) / 1e3,
             sum(e - s for s, e, _, _ in seg) / 1e6, ", ".join("%.0fus before %s" % (g / 1e3, n[:30]) for g, n in big)))
