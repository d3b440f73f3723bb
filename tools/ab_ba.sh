#!/bin/bash
# A/B of the fused BA kernel variants on the GPU box: parity test first, then bench lines (kernel avg from HIP events)
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
python -m pytest tests/test_ba_fused_gpu.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/ab_parity.log
for spec in "$@"; do
  env $spec SDSO_BENCH_SKIP_OTHERS=1 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec', 'ms/step %.3f'%d['ms_per_step'], 'lin %.4f'%d['roofline']['kernel_avg_ms'], 'frac %.3f'%d['roofline']['frac'], 'sc %.3f'%d['extra']['k_ba_sc_avg_ms'])"
done
