#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c6
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3 | tee gpurun_out/r6c6/pytest_wgrad.txt
python tools/r6/wgrad_micro.py 0 1 2>&1 | grep wgrad | tee gpurun_out/r6c6/micro.txt
python tools/r6/wgrad_micro.py 0 1 2>&1 | grep wgrad | tee -a gpurun_out/r6c6/micro.txt
