#!/bin/bash
# one round's numbers for profiles/: bench lines of every workload, single-call latencies, rocprofv3 passes.
# Usage on the GPU box: bash tools/final_round.sh r06   (one script for every round: the per-round copies of rounds 2-5 are in the git history)
tag=${1:?round tag, e.g. r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 2>gpurun_out/${tag}_bench_ba.err | tail -1 > gpurun_out/${tag}_bench_ba.json
SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --batch 128 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_ba_128_windows.json
SDSO_BA_GROUPS=2 SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_ba_two_groups.json
timeout -k 10 300 python bench.py --steps 20 --scaling strong --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_ba_strong.json
SDSO_BA_NO_J=1 SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_ba_noj.json
timeout -k 10 300 python bench.py --workload tracker --steps 50 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_tracker.json
timeout -k 10 300 python bench.py --workload trace --steps 50 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_trace.json
echo bench done
timeout -k 10 300 python tests/diag/bench_latency.py > gpurun_out/${tag}_latency.json 2>gpurun_out/${tag}_latency.err
timeout -k 10 200 python tools/time_optimize.py > gpurun_out/${tag}_optimize_times.txt 2>&1
timeout -k 10 100 python tools/time_track.py > gpurun_out/${tag}_time_track_final.txt 2>&1
echo latency done
if [ "${SKIP_PROFILES:-0}" != "1" ]; then bash tools/profile_all.sh $tag > gpurun_out/profile_${tag}.log 2>&1; fi
echo profiles done
ls gpurun_out | wc -l
