#!/bin/bash
# round 5, GPU call 8: rotated f32 depthwise forward under the kink-robust strict tests; what sits in the main-queue bubbles
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c8; mkdir -p $O
timeout 1800 python -m pytest tests/test_model_gpu.py tests/test_cc_gpu.py -x -q -s -m gpu -k "conditioned" > $O/pytest_strict.txt 2>&1; grep -E "conditioned|kink|flip|passed|failed|Error" $O/pytest_strict.txt | cut -c1-400 | tail -40
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dw" > $O/pytest_dw.txt 2>&1; tail -2 $O/pytest_dw.txt
bash tools/trace_bubbles.sh > $O/bubbles.txt 2>&1; head -60 $O/bubbles.txt
timeout 600 python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-kernel-profile 2>/dev/null | tail -1 | cut -c1-200
