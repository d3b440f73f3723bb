#!/bin/bash
# A/B of environment variants of the default BA bench: tools/ab_env.sh "VAR=1 VAR2=x" "..." ; prints ms/step and the kernel averages (HIP events)
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
for spec in "$@"; do
  env $spec SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 200 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('%-44s ms/step %.3f  value %.3g  lin %.4f frac %.3f  sc %.3f  tail %.3f  resub %s' % ('$spec', d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], e.get('k_ba_sc_avg_ms') or 0, e.get('k_ba_tail_avg_ms') or 0, ('%.3f' % e['k_ba_resub_avg_ms']) if e.get('k_ba_resub_avg_ms') else 'fused'))"
done
