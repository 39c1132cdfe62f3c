#!/bin/bash
# round 5, GPU call 10: one-round-trip parameter loads (block_out_bwd, pw_wgrad); strict tests with the pruned kink search
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c10; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_pw_wg_gpu.py -x -q -m gpu > $O/pytest_ops.txt 2>&1; tail -2 $O/pytest_ops.txt
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so 2>&1 | tee $O/ab.txt
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --kernel-table $O/kt.json > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_cc_gpu.py -q -s -m gpu -k "conditioned or cc_vs_reference_golden" > $O/pytest_strict.txt 2>&1; grep -E "kink:|passed|failed|Error" $O/pytest_strict.txt | cut -c1-250 | tail -30
