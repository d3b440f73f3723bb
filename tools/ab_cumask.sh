#!/bin/bash
# CU-partition A/B of the 256-window step (round-5 verdict item 4): one group on plain streams against two chained groups whose Schur + tail
# kernels own k CUs (sdso_ctx_partition_cus).  Usage on the GPU box: bash tools/ab_cumask.sh > gpurun_out/r06_cumask_ab.txt
export SDSO_DEBUG_ENV=1
run() {
  env "$@" SDSO_BENCH_SKIP_OTHERS=1 SDSO_BENCH_SECONDARY=${SEC:-1} timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
f=lambda v: ('%.3f' % v) if isinstance(v,(int,float)) else str(v)
print('%-46s ms/step %.3f  value %.4g  lin %.4f  sc %s  tail %s  resub %s' % ('$*', d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], f(e.get('k_ba_sc_avg_ms')), f(e.get('k_ba_tail_avg_ms')), f(e.get('k_ba_resub_avg_ms'))))"
}
run SDSO_BA_GROUPS=1
run SDSO_BA_GROUPS=2
for k in 16 32 48 64; do
  run SDSO_BA_GROUPS=2 SDSO_BA_CUMASK=$k
  run SDSO_BA_GROUPS=2 SDSO_BA_CUMASK=$k,32
done
run SDSO_BA_GROUPS=2 SDSO_BA_CUMASK=96,32
run SDSO_BA_GROUPS=2 SDSO_BA_CUMASK=128,32
run SDSO_BA_GROUPS=3 SDSO_BA_CUMASK=64,32
run SDSO_BA_GROUPS=4 SDSO_BA_CUMASK=64,32
run SDSO_BA_GROUPS=1
