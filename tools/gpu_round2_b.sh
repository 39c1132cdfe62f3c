#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_tests.sh
for mode in 0 1; do
  C3D_PY_STAGE=$mode timeout 600 python bench.py --no-cpu-baseline --no-kernel-profile > gpurun_out/r2b_bench_py$mode.json 2> gpurun_out/r2b_bench_py$mode.err
  echo "PY_STAGE=$mode: $(python -c "import json;d=json.load(open('gpurun_out/r2b_bench_py$mode.json'));print(d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'])")"
done
for b in 8 16; do
  timeout 600 python bench.py --batch $b --no-cpu-baseline --no-kernel-profile > gpurun_out/r2b_bench_b$b.json 2> gpurun_out/r2b_bench_b$b.err
  echo "B=$b: $(python -c "import json;d=json.load(open('gpurun_out/r2b_bench_b$b.json'));print(d['value'],d['ms_per_step'],d['config']['host_enqueue_ms_per_step'])")"
done
