#!/bin/bash
# forward cooperative kernel with two tiles in flight: op tests, A/B against the library before, kernel times
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_c_forward or conv_a_forward" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "cooperative or stage" 2>&1 | tail -4
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 40
bash tools/r6/call33.sh 2>&1 | grep "total\|pw_cfwd"
