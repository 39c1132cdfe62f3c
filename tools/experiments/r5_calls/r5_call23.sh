#!/bin/bash
# round 5, GPU call 23: wave-relative buffer resources (no 2 GiB limit) -- op / model parity, the > 2 GiB case, A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c23; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_pw_wg_gpu.py tests/test_wide_ops_gpu.py -x -q -m gpu > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_bf16_fullsize_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -2 $O/pytest_model.txt
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so 2>&1 | tee $O/ab.txt
