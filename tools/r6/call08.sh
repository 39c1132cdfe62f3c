#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c8
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3 | tee gpurun_out/r6c8/pytest_wgrad.txt
for rep in 1 2; do for L in libchange3d_hip.so libchange3d_hip_nopk.so; do
  echo "== $L"; C3D_LIB=$(pwd)/change3d_amd/lib/$L python tools/r6/wgrad_micro.py 1 2>&1 | grep wgrad
done; done | tee gpurun_out/r6c8/micro.txt
