#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c4
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -5 | tee gpurun_out/r6c4/pytest_wgrad.txt
python tools/r6/wgrad_micro.py 0 1 2>&1 | tee gpurun_out/r6c4/micro.txt
bash tools/ab_option.sh PW_WGRAD_V2 0 1 2>&1 | tee gpurun_out/r6c4/ab.txt
