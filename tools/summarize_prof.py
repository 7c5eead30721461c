#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV  ->  a short per-kernel summary (markdown) for profiles/."""
import csv
import glob
import sys


def main(prof_dir, out, title, steps):
    path = sorted(glob.glob(prof_dir + "/**/*_kernel_stats.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("# %s\n\nsource: `%s` (rocprofv3 --kernel-trace --stats); %d train steps in the trace; "
                "total GPU kernel time %.2f ms = %.2f ms/step\n\n" % (title, path.split("/")[-1], steps, tot / 1e6, tot / 1e6 / steps))
        f.write("| kernel | calls | total ms | avg us | % | ms/step |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            t = float(r["TotalDurationNs"])
            f.write("| `%s` | %s | %.2f | %.1f | %.1f | %.2f |\n" % (r["Name"][:100].replace("|", "/"), r["Calls"], t / 1e6,
                                                                   float(r["AverageNs"]) / 1e3, 100 * t / tot, t / 1e6 / steps))
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
