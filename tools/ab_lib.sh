#!/bin/bash
# A/B of two built libraries in ONE GPU call: tools/ab_lib.sh libA.so libB.so [bench args]   (paths under change3d_amd/lib/)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    C3D_LIB=$(pwd)/change3d_amd/lib/$L timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
