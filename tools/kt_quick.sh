#!/bin/bash
# kernel table of the default BA bench (two groups) and of the single-stream run: tools/kt_quick.sh
export SDSO_DEBUG_ENV=1   # the library reads its A/B switches only behind this gate
root=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for g in 2 1; do
  rm -rf /tmp/kt$g
  SDSO_BA_GROUPS=$g SDSO_BENCH_SKIP_OTHERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$g -- python3 $root/bench.py --steps 30 --no-cpu-baseline > /tmp/kt$g.log 2>&1
  echo "== groups $g: $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt$g.log | head -1)"
  python3 $root/tools/prof_summary.py /tmp/kt$g | grep k_ba | head -9
done
