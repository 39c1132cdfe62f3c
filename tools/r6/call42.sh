#!/bin/bash
# conv_a gradient kernel: one straight-line iteration, two tiles in flight at 216 -> 96 and 54 -> 24: op tests, A/B, kernel times
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 40
bash tools/r6/call33.sh 2>&1 | grep "total\|pw_cdg_a"
