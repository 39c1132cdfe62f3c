#!/bin/bash
# round 6, call 7: v2 weight-gradient kernel (multiply of tile t beside the conversion of tile t + 1): A/B, kernel table, model-level tests
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c7
bash tools/ab_option.sh PW_WGRAD_V2 0 1 2>&1 | tee gpurun_out/r6c7/ab.txt
for t in scd cc; do
  for v in 0 1; do
    timeout 600 python bench.py --task $t --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option PW_WGRAD_V2=$v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t PW_WGRAD_V2=$v', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done 2>&1 | tee gpurun_out/r6c7/ab_scd_cc.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_bf16_fullsize_gpu.py tests/test_pw_wg_gpu.py tests/test_round3_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6c7/pytest_model.txt
