#!/bin/bash
# half-vector depthwise forward with branch-free buffer loads / stores (counted waits): tests, A/B against the library before, kernel times
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dw" 2>&1 | tail -3
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 40
bash tools/r6/call33.sh 2>&1 | grep "total\|dw_fwd"
