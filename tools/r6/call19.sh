#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c19
python tools/debug/cfwd_diff.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6c19/diff.txt
python tools/debug/cfwd_diff.py se 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6c19/diff_se.txt
