#!/bin/bash
# Cache-level counters of one bench command (L1 tag lookups vs L1->L2 read requests; L2 hits vs misses; LDS instructions / bank conflicts).
# Usage: tools/profile_cache.sh <tag> [bench args]
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
args="--steps 10 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/prof_${tag}_l1 -- python3 $root/bench.py $args > $out/prof_${tag}_l1.log 2>&1 || rocprofv3 --pmc TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/prof_${tag}_l1 -- python3 $root/bench.py $args > $out/prof_${tag}_l1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/prof_${tag}_l2 -- python3 $root/bench.py $args > $out/prof_${tag}_l2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/prof_${tag}_lds -- python3 $root/bench.py $args > $out/prof_${tag}_lds.log 2>&1
python3 $root/tools/prof_summary.py $out/prof_${tag}_l1 $out/prof_${tag}_l2 $out/prof_${tag}_lds > $out/prof_${tag}_cache_summary.txt
