#!/bin/bash
# round-2 evidence: kernel stats + FETCH/WRITE passes for the three workloads, SQ counters for the BA and trace kernels
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
bash tools/profile_round.sh r02_ba && echo ba done
bash tools/profile_round.sh r02_tracker --workload tracker && echo tracker done
bash tools/profile_round.sh r02_trace --workload trace && echo trace done
SDSO_BENCH_SKIP_OTHERS=1 bash tools/profile_sq.sh r02_ba && echo ba sq done
bash tools/profile_sq.sh r02_trace --workload trace && echo trace sq done
python3 tools/make_traffic.py r02 ba=gpurun_out/prof_r02_ba tracker=gpurun_out/prof_r02_tracker trace=gpurun_out/prof_r02_trace > gpurun_out/r02_traffic_print.txt 2>&1
cp profiles/r02_traffic.json gpurun_out/r02_traffic.json
ls gpurun_out | head -50
# only the summaries travel back (the raw rocprofv3 directories exceed gpurun's 64 MiB return limit)
for d in gpurun_out/prof_r02_*; do if [ -d "$d" ]; then rm -rf "$d"; fi; done
du -sh gpurun_out
