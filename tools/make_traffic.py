#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernels from the rocprofv3 PMC passes -> profiles/<round>_traffic.json.

  python tools/make_traffic.py <round-tag> <workload>=<prof-dir-prefix> ...
  e.g.  python tools/make_traffic.py r01 ba=gpurun_out/prof_r01_ba tracker=gpurun_out/prof_r01_tracker

FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM); the
calibration in tools/calib_fetch.hip shows the same for the sparse 16-B gathers of this code (2.25 requests per
bilinear sample = 128-B lines), so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNEL = {"ba": "k_ba_lin_fused", "tracker": "k_track_eval", "trace": "k_trace_stereo"}


def mean_counter(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"].split("(")[0].replace("sdso::", "").replace("void ", "")]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    out = {}
    for arg in sys.argv[2:]:
        wl, prefix = arg.split("=")
        fetch, write = mean_counter(prefix + "_fetch", "FETCH_SIZE"), mean_counter(prefix + "_write", "WRITE_SIZE")
        kern = [k for k in fetch if k.startswith(KERNEL[wl])]
        if not kern:
            continue
        k = kern[0]
        cfg = None
        jf = prefix + "_bench_under_rocprof.json"
        if os.path.exists(jf) and os.path.getsize(jf):
            cfg = json.loads(open(jf).read().strip().splitlines()[-1])["config"]
        out[wl] = {"kernel": k, "fetch_size_kib_per_launch": fetch[k], "write_size_kib_per_launch": write.get(k, 0.0),
                   "traffic_bytes_per_launch": (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0, "config": cfg,
                   "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: 128-B read requests counted at 64 B; tools/calib_fetch.hip)"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag + "_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
