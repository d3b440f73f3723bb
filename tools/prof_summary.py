#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories into one small table per run.

  python tools/prof_summary.py <dir> [<dir> ...]

For *_kernel_stats.csv: per-kernel calls / average / total (the rows of sdso:: kernels first).
For *_counter_collection.csv: per kernel and counter, the mean value per dispatch.  FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read requests at 64 B
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("sdso::", "").replace("void ", "")


def main():
    for d in sys.argv[1:]:
        print("== %s" % d)
        for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
            rows = list(csv.DictReader(open(f)))
            if not any("k_" in r["Name"] for r in rows):
                continue
            print("-- %s" % os.path.relpath(f, d))
            print("%-28s %8s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
            for r in rows:
                print("%-28s %8s %12.2f %12.3f %7s" % (short(r["Name"])[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                a = acc[(short(r["Kernel_Name"]), r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
            if not any(k[0].startswith("k_") for k in acc):
                continue
            print("-- %s" % os.path.relpath(f, d))
            print("%-28s %-14s %8s %16s" % ("kernel", "counter", "disp", "mean/dispatch"))
            for (k, c), (s, n) in sorted(acc.items()):
                print("%-28s %-14s %8d %16.3f" % (k[:28], c, n, s / n))


if __name__ == "__main__":
    main()
