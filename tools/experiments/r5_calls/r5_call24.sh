#!/bin/bash
# round 5, call 24: c3d_block_out_bwd folded into the conv_a data gradient (C3D_WG_MASKSUM / add_sums): op tests, model tests,
# A/B through the run-time option (MASK_IN_DGRAD = 1: mask only where the weight gradient is fused, as before; 3: + sums, + res4)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/c24_ops.txt
cat gpurun_out/c24_ops.txt
for rep in 1 2; do
  for o in 1 3; do
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option MASK_IN_DGRAD=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MASK_IN_DGRAD=$o rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s', d['config'].get('final_loss'))"
  done
done | tee gpurun_out/c24_ab.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/c24_model.txt
