#!/bin/bash
# one round's rocprofv3 evidence: kernel stats + FETCH/WRITE passes for the three workloads, SQ counters for the BA kernels.
# Usage on the GPU box: bash tools/profile_all.sh r06
tag=${1:?round tag, e.g. r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
bash tools/profile_round.sh ${tag}_ba && echo ba done
bash tools/profile_round.sh ${tag}_tracker --workload tracker && echo tracker done
bash tools/profile_round.sh ${tag}_trace --workload trace && echo trace done
SDSO_BENCH_SKIP_OTHERS=1 bash tools/profile_sq.sh ${tag}_ba && echo ba sq done
python3 tools/make_traffic.py $tag ba=gpurun_out/prof_${tag}_ba tracker=gpurun_out/prof_${tag}_tracker trace=gpurun_out/prof_${tag}_trace > gpurun_out/${tag}_traffic_print.txt 2>&1
cp profiles/${tag}_traffic.json gpurun_out/${tag}_traffic.json
# only the summaries travel back (the raw rocprofv3 directories exceed gpurun's 64 MiB return limit)
for d in gpurun_out/prof_${tag}_*; do if [ -d "$d" ]; then rm -rf "$d"; fi; done
du -sh gpurun_out
