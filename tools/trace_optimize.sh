#!/bin/bash
# kernel timeline of one device-resident sdso_ba_optimize (8KF / 2000 points): kernel durations vs the gaps between them
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kto
rocprofv3 --kernel-trace --output-format csv -d /tmp/kto -- python3 $root/tools/dbg_optimize_trace.py > /tmp/kto.log 2>&1
tail -3 /tmp/kto.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kto/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last optimize: from the last k_ba_reset_all on
idx = max(i for i, r in enumerate(rows) if "k_ba_reset_all" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("kernels %d, span %.1f us, sum of durations %.1f us" % (len(rows), (t1 - t0) / 1e3, busy / 1e3))
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
prev_end = None
gaps = 0.0
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("sdso::", "").replace("void ", "")[:28]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[n][0] += 1; acc[n][1] += d
    if prev_end is not None: gaps += max(0, int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = int(r["End_Timestamp"])
print("sum of gaps between consecutive kernels %.1f us" % gaps)
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s x%3d  %8.1f us total  %6.1f avg" % (n, c, d, d / c))
PY
