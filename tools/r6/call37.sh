#!/bin/bash
# reducers of the cooperative data + weight gradients deferred to the end of a stage pass: tests, A/B against the library before
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "stage or folded or end_to_end or golden or side or dp or allreduce" 2>&1 | tail -4
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 40
