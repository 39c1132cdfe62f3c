#!/bin/bash
# bench with the per-kernel table; usage: tools/gpu_bench_kt.sh <tag> [extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
timeout 900 python bench.py --no-cpu-baseline --kernel-table gpurun_out/${tag}_kernels.json "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('${tag}:', d['value'], 'img/s', d['ms_per_step'], 'ms/step host', d['config']['host_enqueue_ms_per_step'], 'kernel sum', d.get('kernel_time_ms_eager_step'))
k=json.load(open('gpurun_out/${tag}_kernels.json'))
for r in k['kernels'][:32]: print(f"  {r['kernel']:26s} x{r['launches']:4d} {r['ms_total']:8.3f} ms {r['GBps']:8.1f} GB/s")
PY
