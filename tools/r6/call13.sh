#!/bin/bash
# round 6, call 13: round profile (bench, rocprofv3 kernel stats overlapped / serial, FETCH / WRITE / MFMA counters, host profile) -> r06_*
cd "${GRAFT_REPO_ROOT:-.}"
C3D_ROUND_TAG=r06 bash tools/profile_round.sh 2>&1 | tail -12
cp gpurun_out/bench.json gpurun_out/r06_bench_bcd.json
bash tools/trace_step.sh > gpurun_out/trace_step_r06.log 2>&1 || true
bash tools/pmc_sq.sh r06 > gpurun_out/pmc_sq_r06.log 2>&1 || true
