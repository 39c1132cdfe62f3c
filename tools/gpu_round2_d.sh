#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_tests.sh
C3D_PY_STAGE=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bf16_fullsize_gpu.py -m gpu -q -x > gpurun_out/pytest_gpu_pystage.log 2>&1
echo "py-stage path: $(tail -1 gpurun_out/pytest_gpu_pystage.log)"
for f in 1 0; do
  C3D_FOLD_FIN=$f timeout 600 python bench.py --no-cpu-baseline --no-kernel-profile > gpurun_out/r2d_bench_fold$f.json 2> gpurun_out/r2d_bench_fold$f.err
  echo "FOLD_FIN=$f B=32: $(python -c "import json;d=json.load(open('gpurun_out/r2d_bench_fold$f.json'));print(d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'])")"
  C3D_FOLD_FIN=$f timeout 600 python bench.py --batch 16 --no-cpu-baseline --no-kernel-profile > gpurun_out/r2d_bench_fold${f}_b16.json 2> gpurun_out/r2d_bench_fold${f}_b16.err
  echo "FOLD_FIN=$f B=16: $(python -c "import json;d=json.load(open('gpurun_out/r2d_bench_fold${f}_b16.json'));print(d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'])")"
done
