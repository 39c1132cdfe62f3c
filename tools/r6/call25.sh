#!/bin/bash
# fork / join packets of the backward pass: deferred weight-gradient forks, batched joins (env A/B in one call)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage or side" 2>&1 | tail -5
run() { # name, env...
  n=$1; shift
  for rep in 1 2 3; do
    env "$@" timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile > gpurun_out/c25_$n.json 2> gpurun_out/c25_$n.err
    python -c "import json;d=json.load(open('gpurun_out/c25_$n.json'));print('$n rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
}
run old C3D_SIDE_DEFER=0 C3D_SIDE_JOIN_EVERY=1 C3D_BWD_RING=3
run defer C3D_SIDE_DEFER=1 C3D_SIDE_JOIN_EVERY=1 C3D_BWD_RING=4
run defer_j2 C3D_SIDE_DEFER=1 C3D_SIDE_JOIN_EVERY=2 C3D_BWD_RING=5
run defer_j3 C3D_SIDE_DEFER=1 C3D_SIDE_JOIN_EVERY=3 C3D_BWD_RING=6
run old C3D_SIDE_DEFER=0 C3D_SIDE_JOIN_EVERY=1 C3D_BWD_RING=3
run defer_j2 C3D_SIDE_DEFER=1 C3D_SIDE_JOIN_EVERY=2 C3D_BWD_RING=5
