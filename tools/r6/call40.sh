#!/bin/bash
# re-sweep of the older run-time options at the final kernels (they now only touch the first blocks of a stage / the decoder)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for rep in 1 2; do
  run
  run --option FUSE_WGRAD=0
  run --option FUSE_WGRAD=1
  run --option SIDE_STREAM=0
  run --option MASK_IN_DGRAD=1
done
