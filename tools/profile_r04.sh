#!/bin/bash
# round-4 evidence: kernel stats + FETCH/WRITE passes for the three workloads, SQ counters for the BA kernels
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
bash tools/profile_round.sh r04_ba && echo ba done
bash tools/profile_round.sh r04_tracker --workload tracker && echo tracker done
bash tools/profile_round.sh r04_trace --workload trace && echo trace done
SDSO_BENCH_SKIP_OTHERS=1 bash tools/profile_sq.sh r04_ba && echo ba sq done
python3 tools/make_traffic.py r04 ba=gpurun_out/prof_r04_ba tracker=gpurun_out/prof_r04_tracker trace=gpurun_out/prof_r04_trace > gpurun_out/r04_traffic_print.txt 2>&1
cp profiles/r04_traffic.json gpurun_out/r04_traffic.json
# only the summaries travel back (the raw rocprofv3 directories exceed gpurun's 64 MiB return limit)
for d in gpurun_out/prof_r04_*; do if [ -d "$d" ]; then rm -rf "$d"; fi; done
du -sh gpurun_out
