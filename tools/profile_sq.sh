#!/bin/bash
# SQ counter pass (wave cycles / stalls / instruction counts) for one bench command.  Usage: tools/profile_sq.sh <tag> [bench args]
set -e -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
args="--steps 10 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/prof_${tag}_sq -- python3 $root/bench.py $args > $out/prof_${tag}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/prof_${tag}_sq2 -- python3 $root/bench.py $args > $out/prof_${tag}_sq2.log 2>&1 || true
python3 $root/tools/prof_summary.py $out/prof_${tag}_sq $out/prof_${tag}_sq2 > $out/prof_${tag}_sq_summary.txt
