#!/bin/bash
# round 5, GPU call 11: whole GPU suite, then the round profile (bench with also + CPU baseline, rocprofv3 kernel stats overlapped / serial, FETCH / WRITE / MFMA counters, host profile), SQ wave states, trace gaps
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
C3D_TEST_TIMEOUT=2400 bash tools/gpu_tests.sh
C3D_ROUND_TAG=r05 bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log
bash tools/pmc_sq.sh r05 > gpurun_out/pmc_sq.log 2>&1
bash tools/trace_step.sh > gpurun_out/trace_step_r05.log 2>&1; head -30 gpurun_out/trace_gaps.txt
