#!/bin/bash
# round 6, call 16: ring depth of the backward temporaries (how far the side queue may lag) with the faster weight-gradient kernel
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c16
bash tools/ab_lib.sh libchange3d_hip.so libchange3d_hip_ring4.so 2>&1 | tee gpurun_out/r6c16/ab_34.txt
bash tools/ab_lib.sh libchange3d_hip_ring2.so libchange3d_hip_ring6.so 2>&1 | tee gpurun_out/r6c16/ab_26.txt
