#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c5
for L in libchange3d_hip.so libchange3d_hip_s0.so libchange3d_hip_q12.so libchange3d_hip_q26.so; do
  echo "== $L"; C3D_LIB=$(pwd)/change3d_amd/lib/$L python tools/r6/wgrad_micro.py 1 2>&1 | grep wgrad
done | tee gpurun_out/r6c5/micro.txt
