#!/bin/bash
# round 5, GPU call 17: register-prefetch fused depthwise backward (32 x 32 maps, stride 2, f32) with buffer-addressed loads / stores -- parity, per-shape and step A/B against the library of the previous commit
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c17; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dw" > $O/pytest_dw.txt 2>&1; tail -2 $O/pytest_dw.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage or reproducible or golden" > $O/pytest_model.txt 2>&1; tail -2 $O/pytest_model.txt
for L in libchange3d_hip_r5a.so libchange3d_hip.so libchange3d_hip_r5a.so libchange3d_hip.so; do echo "== $L"; C3D_LIB=$(pwd)/change3d_amd/lib/$L timeout 300 python tools/bench_ops.py dw 2>&1 | grep "bwd fused"; done > $O/bench_ops_dw.txt 2>&1; cat $O/bench_ops_dw.txt | cut -c1-160
bash tools/ab_lib.sh libchange3d_hip_r5a.so libchange3d_hip.so 2>&1 | tee $O/ab.txt
bash tools/ab_lib.sh libchange3d_hip_r5a.so libchange3d_hip.so --task scd 2>&1 | tee $O/ab_scd.txt
