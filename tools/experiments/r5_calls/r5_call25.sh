#!/bin/bash
# round 5, call 25: per-kernel tables with and without the folded block-output backward; op tests after the pass fix
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/c25_ops.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "long_walks" 2>&1 | tail -4 | tee -a gpurun_out/c25_ops.txt
for o in 1 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --option MASK_IN_DGRAD=$o --kernel-table gpurun_out/c25_kt$o.json > gpurun_out/c25_b$o.json 2> gpurun_out/c25_b$o.err
  echo "== MASK_IN_DGRAD=$o"; grep "^\[kernels\]" gpurun_out/c25_b$o.err | head -8; tail -1 gpurun_out/c25_b$o.json | cut -c1-200
done
