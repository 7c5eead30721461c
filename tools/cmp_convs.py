"""Per-shape comparison of two `bench.py --dump-convs` files: python tools/cmp_convs.py A.json B.json"""
import json
import sys
from collections import defaultdict


def load(path):
    g = defaultdict(lambda: [0, 0.0, 0.0])
    for e in json.load(open(path)):
        k = (e["name"], tuple(e["args"]))
        g[k][0] += 1
        g[k][1] += e["ms"]
        g[k][2] += e["gflop"]
    return g


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((a[k][1] - b[k][1], k) for k in a if k in b), reverse=True)
print("total ms: A %.3f  B %.3f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for d, k in rows[:12] + rows[-12:]:
    print("%-16s %-64s n=%3d  A %7.3f ms %5.0f TF   B %7.3f ms %5.0f TF   A-B %+6.3f" % (
        k[0][5:], str(k[1]), a[k][0], a[k][1], a[k][2] / a[k][1], b[k][1], b[k][2] / b[k][1], d))
