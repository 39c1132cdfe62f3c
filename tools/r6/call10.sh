#!/bin/bash
# round 6, call 10: which weight gradients to fuse now that the separate kernel is faster; side-stream width of the v2 kernel
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c10
bash tools/ab_option.sh FUSE_WGRAD 3 2 1 0 2>&1 | tee gpurun_out/r6c10/ab_fuse.txt
bash tools/ab_lib.sh libchange3d_hip.so libchange3d_hip_side5.so 2>&1 | tee gpurun_out/r6c10/ab_side5.txt
bash tools/ab_lib.sh libchange3d_hip_side6.so libchange3d_hip_side8.so 2>&1 | tee gpurun_out/r6c10/ab_side68.txt
