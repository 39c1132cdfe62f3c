#!/bin/bash
# round 5, GPU call 12: the round profile (bench with also + CPU baseline, rocprofv3 kernel stats overlapped / serial, FETCH / WRITE / MFMA counters, host profile), SQ wave states, trace gaps; raw rocprof output pruned before the merge (64 MiB limit)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
C3D_ROUND_TAG=r05 bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log
for d in prof_stats prof_stats_serial; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_rocprofv3_kernel_stats${d#prof_stats}.csv; done
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_serial gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/host.prof
bash tools/pmc_sq.sh r05 > gpurun_out/pmc_sq.log 2>&1
bash tools/trace_step.sh > gpurun_out/trace_step_r05.log 2>&1; head -24 gpurun_out/trace_gaps.txt
rm -rf gpurun_out/trace_step
du -sh gpurun_out
