#!/bin/bash
# step time of the default BA bench against the number of stream groups / the chaining of their accumulate phases
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
mkdir -p gpurun_out
for g in ${GROUPS_LIST:-1 2 3 4}; do
  for nc in 0 1; do
    if [ $g = 1 ] && [ $nc = 1 ]; then continue; fi
    if [ $nc = 1 ]; then export SDSO_BA_NOCHAIN=1; else unset SDSO_BA_NOCHAIN; fi
    SDSO_BA_GROUPS=$g SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 120 python bench.py --steps 30 --no-cpu-baseline > gpurun_out/ab_g${g}_${nc}.log 2>&1
    python - <<PY
import json
l = open("gpurun_out/ab_g${g}_${nc}.log").read().strip().split("\n")[-1]
try:
    d = json.loads(l)
    print("groups $g nochain $nc: ms_per_step %.4f  lin_fused avg %.4f ms  frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("groups $g nochain $nc: failed", l[-300:])
PY
  done
done
