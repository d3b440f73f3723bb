#!/bin/bash
# per-kernel average durations of the default bench (rocprofv3 --kernel-trace --stats), printed as a short table
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$1
shift
rm -rf $OUT; mkdir -p $OUT
env "$@" SDSO_BENCH_SKIP_OTHERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-60s calls %5s avg %8.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('ms/step %.3f value %.3g'%(d['ms_per_step'], d['value']))"
