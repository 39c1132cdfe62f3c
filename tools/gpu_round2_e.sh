#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
C3D_TEST_ARGS="-x" bash tools/gpu_tests.sh
for f in 1 0; do
  C3D_FOLD_FIN=$f timeout 600 python bench.py --no-cpu-baseline --kernel-table gpurun_out/r2e_kernels_fold$f.json > gpurun_out/r2e_bench_fold$f.json 2> gpurun_out/r2e_bench_fold$f.err
  echo "FOLD_FIN=$f B=32: $(python -c "import json;d=json.load(open('gpurun_out/r2e_bench_fold$f.json'));print(d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'], d['kernel_time_ms_eager_step'])")"
done
