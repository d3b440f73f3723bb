#!/bin/bash
# L2 -> fabric read requests of one bench command (TCC_EA0_RDREQ: all / 32-byte ones), L1 -> L2 requests and L2 hit / miss:
# the counters behind "how many bytes does this kernel really pull from HBM".   Usage: tools/profile_ea.sh <tag> [bench args]
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
args="--steps 10 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $out/prof_${tag}_ea -- python3 $root/bench.py $args > $out/prof_${tag}_ea.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/prof_${tag}_l2 -- python3 $root/bench.py $args > $out/prof_${tag}_l2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/prof_${tag}_l1 -- python3 $root/bench.py $args > $out/prof_${tag}_l1.log 2>&1
python3 $root/tools/prof_summary.py $out/prof_${tag}_ea $out/prof_${tag}_l2 $out/prof_${tag}_l1 > $out/prof_${tag}_ea_summary.txt
