#!/bin/bash
# k_ba_sc_host: workgroup per host (WPH=0) against wave per host (WPH=1) by batch size.  Usage on the GPU box: bash tools/ab_sc_batch.sh
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
for b in 64 128 160 192 224 256; do for w in 1 0; do
SDSO_BA_SC_WPH=$w SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 200 python bench.py --steps 30 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('batch $b WPH=$w  ms/step %.4f  lin %.4f  sc %.4f  tail %.4f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], e.get('k_ba_sc_avg_ms') or 0, e.get('k_ba_tail_avg_ms') or 0))"
done; done
