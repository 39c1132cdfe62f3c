#!/bin/bash
# prologue round trips: phase clocks of the cooperative kernel, bit-identity tests, step A/B against the library before the change
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
C3D_LIB=$PWD/change3d_amd/lib/libchange3d_hip_cfclk.so python tools/r6/cfwd_clock.py 2>&1 | grep conv_ | tee gpurun_out/c26_clock.txt
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "conv_c_forward or conv_a_forward or stage or folded or se_gate or finalize or dw" 2>&1 | tail -5
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --steps 60
