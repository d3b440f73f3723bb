#!/bin/bash
# durations of the successive k_ba_lin_fused launches of the default bench under rocprofv3 --kernel-trace: tools/lin_durations.sh tag [ENV=..]
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ld_$tag
rm -rf $OUT; mkdir -p $OUT
env "$@" SDSO_BENCH_SKIP_OTHERS=1 SDSO_BENCH_SECONDARY=0 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - "$OUT/kt_kernel_trace.csv" "$tag" <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_ba_lin_fused" in r["Kernel_Name"]][6:]
print(sys.argv[2], "lin_fused launches %d  min %.0f med %.0f mean %.0f max %.0f :" % (len(d), min(d), statistics.median(d), statistics.mean(d), max(d)), " ".join("%.0f" % x for x in d[:20]))
PY
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('   ms/step %.3f value %.3g'%(d['ms_per_step'], d['value']))"
