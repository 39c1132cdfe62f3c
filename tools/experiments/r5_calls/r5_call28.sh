#!/bin/bash
# round 5, call 28: run-time options re-checked at the 7/8-wide side-stream weight gradient
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], 'ms', d['value'], 'img/s')"; }
for rep in 1 2; do
  run base
  run side_off --option SIDE_STREAM=0
  run fuse0 --option FUSE_WGRAD=0
  run fuse1 --option FUSE_WGRAD=1
  run fuse2 --option FUSE_WGRAD=2
  run mask1 --option MASK_IN_DGRAD=1
  run ring0 --option DW_RING=0
done 2>&1 | tee gpurun_out/c28_ab.txt
