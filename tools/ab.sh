#!/bin/bash
# A/B a set of library builds on the default BA bench:  tools/ab.sh lib1.so lib2.so ...   (paths relative to the repo root)
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
for lib in "$@"; do
  for env in "A=1" "SDSO_BA_NO_J=1"; do
    env $env SDSO_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $env', round(d['value']/1e6,1), 'Mres/s step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],4))"
  done
done
