#!/bin/bash
# Upper bounds for launch folding (C3D_WHATIF in csrc/stage_driver.hip: launches skipped, results wrong on purpose).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/whatif_$label.err | tail -1 > gpurun_out/whatif_$label.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/whatif_$label.json"))
    print("$label", d["ms_per_step"], "ms", d["value"], "img/s", "host", d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$label", "failed", e)
PY
}
run base C3D_WHATIF=0
run fwdSE C3D_WHATIF=2
run bwdSE C3D_WHATIF=8
run serial_base C3D_WHATIF=0 C3D_WGRAD_SIDE=0
run serial_bwdSE C3D_WHATIF=8 C3D_WGRAD_SIDE=0
run serial_fwdSE C3D_WHATIF=2 C3D_WGRAD_SIDE=0
run serial_plain C3D_WHATIF=1 C3D_WGRAD_SIDE=0
