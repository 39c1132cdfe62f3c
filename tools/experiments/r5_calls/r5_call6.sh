#!/bin/bash
# round 5, GPU call 6: pointwise GEMM with bounds-checked buffer addressing (exact counted waits) -- parity + A/B against the round-4 library
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c6; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_pw_wg_gpu.py tests/test_wide_ops_gpu.py -x -q -m gpu > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage or flag or golden or reproducible" > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so 2>&1 | tee $O/ab.txt
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --kernel-table $O/kt.json > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
