#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c21
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "cooperative" 2>&1 | grep "PW_CFWD 0\|passed\|failed\|Error" | tee gpurun_out/r6c21/pytest_e2e.txt
