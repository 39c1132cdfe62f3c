#!/bin/bash
# A/B of the packed weight images (C3D_PW_IMG): operator tests, then the driver-equivalent bench both ways, twice
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "pw_gemm or pack or residual" > gpurun_out/img_ops.log 2>&1
tail -5 gpurun_out/img_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x > gpurun_out/img_model.log 2>&1
tail -5 gpurun_out/img_model.log
for rep in 1 2; do
  for v in 0 1; do
    C3D_PW_IMG=$v timeout 600 python bench.py --steps 60 --warmup 10 > gpurun_out/img_ab_${v}_${rep}.json 2> gpurun_out/img_ab_${v}_${rep}.err
    python -c "import json;d=json.load(open('gpurun_out/img_ab_${v}_${rep}.json'));print('IMG=$v rep $rep', d['ms_per_step'], d['value'], d['roofline']['frac'])"
  done
done
