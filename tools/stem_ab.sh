#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in 0 1 2 3; do
  C3D_STEM_WX_DBG=$m python bench.py --no-cpu-baseline --steps 6 --warmup 3 2> gpurun_out/stem_$m.err | tail -1 > gpurun_out/stem_$m.json
  echo "DBG=$m"; grep -E "stem_bwd_wx" gpurun_out/stem_$m.err | head -5
done
