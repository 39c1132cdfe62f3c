#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c15
bash tools/ab_lib.sh libchange3d_hip.so libchange3d_hip_dv2.so 2>&1 | tee gpurun_out/r6c15/ab.txt
for L in libchange3d_hip.so libchange3d_hip_dv2.so; do
  C3D_LIB=$(pwd)/change3d_amd/lib/$L timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-also 2>&1 >/dev/null | grep "stem" 
done | tee gpurun_out/r6c15/stem.txt
