#!/bin/bash
# A/B of a run-time library option in ONE GPU call: tools/ab_option.sh FUSE_WGRAD 3 1 0  [-- extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
name=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ $# -gt 0 ] && shift
for rep in 1 2; do
  for v in "${vals[@]}"; do
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option $name=$v "$@" > gpurun_out/ab_${name}_${v}_${rep}.json 2> gpurun_out/ab_${name}_${v}_${rep}.err
    python -c "import json;d=json.load(open('gpurun_out/ab_${name}_${v}_${rep}.json'));print('$name=$v rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
