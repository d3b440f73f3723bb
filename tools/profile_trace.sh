#!/bin/bash
# traceStereo evidence: kernel stats, SQ counters and L1 / L2 / LDS counters of k_trace_stereo_blk at 20 000 and
# 75 000 points (every textured pixel of the pair), the bench lines at both sizes and the L->R->L boundary call.  Usage on the GPU box: bash tools/profile_trace.sh r06
set -e -o pipefail
tag=${1:?round tag, e.g. r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
python3 bench.py --workload trace --steps 200 --warmup 20 > $out/${tag}_bench_trace.json 2> $out/${tag}_bench_trace.err
python3 bench.py --workload trace --batch 75000 --steps 100 --warmup 10 --no-cpu-baseline > $out/${tag}_bench_trace_75k.json 2> $out/${tag}_bench_trace_75k.err
python3 bench.py --workload match --batch 75000 --steps 30 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_match_75k.json 2> $out/${tag}_bench_match_75k.err
python3 bench.py --workload match --steps 50 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_match_20k.json 2> $out/${tag}_bench_match_20k.err
bash tools/profile_round.sh ${tag}_trace --workload trace
bash tools/profile_sq.sh ${tag}_trace --workload trace
bash tools/profile_cache.sh ${tag}_trace --workload trace
bash tools/profile_cache.sh ${tag}_trace75k --workload trace --batch 75000
