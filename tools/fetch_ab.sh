#!/bin/bash
# FETCH_SIZE of the BA kernels under two environments: tools/fetch_ab.sh "VAR=1" ...  (first argument "-" = no variable)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  if [ "$spec" != "-" ]; then export $spec; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_fab_$i -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_fab_$i.log 2>&1
  if [ "$spec" != "-" ]; then unset ${spec%%=*}; fi
  echo "== $spec"; python3 $root/tools/prof_summary.py $out/prof_fab_$i | grep "lin_fused"
  rm -rf $out/prof_fab_$i
done
