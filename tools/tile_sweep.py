"""Reads gpurun_out/sweep_*.json (tools/tile_sweep.sh) and prints, per distinct convolution launch shape of the bench step, the time under
every tile override next to the default choice -- where a fixed override beats the built-in heuristics by more than the run-to-run noise
(two default runs are in the set), the heuristics have something to learn."""
import glob, json, os, collections
runs = {}
for f in sorted(glob.glob("gpurun_out/sweep_*.json")):
    tag = os.path.basename(f)[6:-5]
    g = collections.OrderedDict()
    for r in json.load(open(f)):
        k = (r["name"].replace("mrfp_conv_", ""), tuple(r["args"]))
        e = g.setdefault(k, [0, 0.0])
        e[0] += 1; e[1] += r["ms"]
    runs[tag] = g
tags = [t for t in runs if t not in ("default", "default2")]
base = runs["default"]
gain_total = 0.0
print("%-74s %4s %8s %8s | best override" % ("shape", "n", "default", "default2"))
for k, (n, ms) in sorted(base.items(), key=lambda kv: -kv[1][1]):
    d2 = runs["default2"].get(k, (n, ms))[1]
    ref = min(ms, d2)
    best = min(((runs[t][k][1], t) for t in tags if k in runs[t]), default=(ref, "-"))
    if best[0] < 0.97 * ref and ref - best[0] > 0.005:
        gain_total += ref - best[0]
        print("%-74s %4d %8.3f %8.3f | %-8s %8.3f  (-%.3f ms)" % (str(k)[:74], n, ms, d2, best[1], best[0], ref - best[0]))
print("sum of per-shape gains over the better default run: %.3f ms per step" % gain_total)
for t in ["default", "default2"] + tags:
    print("%-10s conv total %.3f ms" % (t, sum(v[1] for v in runs[t].values())))
