#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (TCC slots: they do not fit one pass).  Usage on the GPU box: tools/profile_round.sh <tag> [bench args]
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
args="--steps 30 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- python3 $root/bench.py $args > $out/prof_${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_fetch -- python3 $root/bench.py $args > $out/prof_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_${tag}_write -- python3 $root/bench.py $args > $out/prof_${tag}_write.log 2>&1
python3 $root/tools/prof_summary.py $out/prof_${tag}_stats $out/prof_${tag}_fetch $out/prof_${tag}_write > $out/prof_${tag}_summary.txt
grep '^{' $out/prof_${tag}_stats.log > $out/prof_${tag}_bench_under_rocprof.json || true
cp $out/prof_${tag}_stats/*/*_kernel_stats.csv $out/prof_${tag}_kernel_stats.csv 2>/dev/null || true
