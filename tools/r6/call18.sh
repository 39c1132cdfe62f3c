#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c18
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -k "cooperative" 2>&1 | grep -v "^$" | tail -40 | tee gpurun_out/r6c18/pytest.txt
