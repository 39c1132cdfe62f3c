#!/bin/bash
# whole GPU suite (no -x: report every failure), log under gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout ${C3D_TEST_TIMEOUT:-2700} python -m pytest tests -m gpu -q -s ${C3D_TEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -30
