#!/bin/bash
# Round-5 evidence for traceStereo (verdict item 8): kernel stats, SQ counters and L1 / L2 / LDS counters of k_trace_stereo_blk at 20 000 and
# 75 000 points (every textured pixel of the pair), the bench lines at both sizes and the L->R->L boundary call.  Usage on the GPU box: bash tools/profile_r05_trace.sh
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
python3 bench.py --workload trace --steps 200 --warmup 20 > $out/r05_bench_trace.json 2> $out/r05_bench_trace.err
python3 bench.py --workload trace --batch 75000 --steps 100 --warmup 10 --no-cpu-baseline > $out/r05_bench_trace_75k.json 2> $out/r05_bench_trace_75k.err
python3 bench.py --workload match --batch 75000 --steps 30 --warmup 5 --no-cpu-baseline > $out/r05_bench_match_75k.json 2> $out/r05_bench_match_75k.err
python3 bench.py --workload match --steps 50 --warmup 5 --no-cpu-baseline > $out/r05_bench_match_20k.json 2> $out/r05_bench_match_20k.err
bash tools/profile_round.sh r05_trace --workload trace
bash tools/profile_sq.sh r05_trace --workload trace
bash tools/profile_cache.sh r05_trace --workload trace
bash tools/profile_cache.sh r05_trace75k --workload trace --batch 75000
