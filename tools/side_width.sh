#!/bin/bash
# How wide should the side-stream weight-gradient kernels be?  (they are single-round kernels: one long-running
# workgroup per CU blocks every main-stream kernel that becomes ready meanwhile)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/sw_$label.err | tail -1 > gpurun_out/sw_$label.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sw_$label.json"))
    print("$label", d["ms_per_step"], "ms", d["value"], "img/s", "host", d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$label", "failed", e)
PY
}
run base C3D_BWD_RING=2
run prio C3D_SIDE_PRIO=1
run early_pw192 C3D_WGC_EARLY=1 C3D_WG_BLOCKS=192
run early_pw160 C3D_WGC_EARLY=1 C3D_WG_BLOCKS=160
run early_pw192_prio C3D_WGC_EARLY=1 C3D_WG_BLOCKS=192 C3D_SIDE_PRIO=1
run early_pw128_prio C3D_WGC_EARLY=1 C3D_WG_BLOCKS=128 C3D_SIDE_PRIO=1
run base2 C3D_BWD_RING=2
