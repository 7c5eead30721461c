#!/usr/bin/env python3
"""Per-shape summary of bench.py --dump-convs output."""
import json, sys
from collections import defaultdict
d = json.load(open(sys.argv[1]))
g = defaultdict(lambda: [0, 0.0, 0.0])
for e in d:
    k = (e['name'], tuple(e['args']))
    g[k][0] += 1; g[k][1] += e['ms']; g[k][2] += e['gflop']
rows = sorted(g.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in g.values())
print("total conv ms %.2f  fwd %.2f  wgrad %.2f" % (tot, sum(v[1] for k, v in g.items() if k[0] == 'mrfp_conv_fwd'), sum(v[1] for k, v in g.items() if k[0] != 'mrfp_conv_fwd')))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for (nme, a), (c, ms, gf) in rows[:n]:
    print("%-11s %-62s x%2d %7.3f ms %7.1f TF/s %5.1f%%" % (nme[5:], str(a), c, ms, gf / ms, 100 * ms / tot))
