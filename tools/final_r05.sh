#!/bin/bash
# round-5 numbers for profiles/: bench lines of every workload, single-call latencies, rocprofv3 passes
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 2>gpurun_out/r05_bench_ba.err | tail -1 > gpurun_out/r05_bench_ba.json
SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --batch 128 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_ba_128_windows.json
SDSO_BA_GROUPS=2 SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_ba_two_groups.json
timeout -k 10 300 python bench.py --steps 20 --scaling strong --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_ba_strong.json
SDSO_BA_NO_J=1 SDSO_BENCH_SKIP_OTHERS=1 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_ba_noj.json
timeout -k 10 300 python bench.py --workload tracker --steps 50 2>/dev/null | tail -1 > gpurun_out/r05_bench_tracker.json
timeout -k 10 300 python bench.py --workload trace --steps 50 2>/dev/null | tail -1 > gpurun_out/r05_bench_trace.json
echo bench done
timeout -k 10 300 python tests/diag/bench_latency.py > gpurun_out/r05_latency.json 2>gpurun_out/r05_latency.err
timeout -k 10 200 python tools/time_optimize.py > gpurun_out/r05_optimize_times.txt 2>&1
timeout -k 10 100 python tools/time_track.py > gpurun_out/r05_time_track_final.txt 2>&1
echo latency done
bash tools/profile_r05.sh > gpurun_out/profile_r05.log 2>&1
echo profiles done
ls gpurun_out | wc -l
