#!/bin/bash
# scheduling-pattern variants of the cooperative conv_a data + weight gradient (library variants, one call)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do
for L in libchange3d_hip.so libchange3d_hip_cd_sched0.so libchange3d_hip_cd_vpm6.so libchange3d_hip_cd_vpm16.so; do
  C3D_LIB=$(pwd)/change3d_amd/lib/$L timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done
