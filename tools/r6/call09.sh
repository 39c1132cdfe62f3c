#!/bin/bash
# round 6, call 9: chained reduction of the separate weight gradients: op tests, A/B (PW_WGRAD_V2 = 3: v2 kernel, reducer launches; 1: chained)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c9
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -8 | tee gpurun_out/r6c9/pytest_wgrad.txt
bash tools/ab_option.sh PW_WGRAD_V2 0 3 1 2>&1 | tee gpurun_out/r6c9/ab.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6c9/pytest_model.txt
