#!/bin/bash
# conv_c gradient kernel of the 48- / 24-channel layers on four-wave workgroups, two per CU: op tests, A/B against the library before, kernel times
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q -m gpu -k cooperative 2>&1 | tail -4
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 40
bash tools/r6/call33.sh 2>&1 | grep "total\|pw_cdg_c"
