#!/bin/bash
# round 6, call 20: the cooperative conv_c forward kernel: op-level bit-identity, end-to-end, SCD / CC A/B
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c20
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv_c_forward" 2>&1 | tail -6 | tee gpurun_out/r6c20/pytest_ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "cooperative" 2>&1 | grep "PW_CFWD\|passed\|failed\|Error" | tee gpurun_out/r6c20/pytest_e2e.txt
for t in scd cc; do
  for v in 0 1; do
    timeout 600 python bench.py --task $t --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option PW_CFWD=$v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t PW_CFWD=$v', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done 2>&1 | tee gpurun_out/r6c20/ab_scd_cc.txt
