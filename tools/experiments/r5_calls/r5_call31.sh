#!/bin/bash
# round 5, call 31: fork / done events of the side stream without the system-scope fence: A/B + the trace of the side queue
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/ab_lib.sh libchange3d_hip_prev.so libchange3d_hip.so 2>&1 | tee gpurun_out/c31_ab.txt
bash tools/trace_step.sh > gpurun_out/trace_step_now.log 2>&1; sed -n 24,40p gpurun_out/trace_side.txt; rm -rf gpurun_out/trace_step
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -3
