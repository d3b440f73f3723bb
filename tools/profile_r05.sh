#!/bin/bash
# round-5 evidence: kernel stats + FETCH/WRITE passes for the three workloads, SQ counters for the BA kernels
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
bash tools/profile_round.sh r05_ba && echo ba done
bash tools/profile_round.sh r05_tracker --workload tracker && echo tracker done
bash tools/profile_round.sh r05_trace --workload trace && echo trace done
SDSO_BENCH_SKIP_OTHERS=1 bash tools/profile_sq.sh r05_ba && echo ba sq done
python3 tools/make_traffic.py r05 ba=gpurun_out/prof_r05_ba tracker=gpurun_out/prof_r05_tracker trace=gpurun_out/prof_r05_trace > gpurun_out/r05_traffic_print.txt 2>&1
cp profiles/r05_traffic.json gpurun_out/r05_traffic.json
# only the summaries travel back (the raw rocprofv3 directories exceed gpurun's 64 MiB return limit)
for d in gpurun_out/prof_r05_*; do if [ -d "$d" ]; then rm -rf "$d"; fi; done
du -sh gpurun_out
