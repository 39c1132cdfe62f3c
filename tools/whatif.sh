#!/bin/bash
# Upper bounds for launch folding (C3D_WHATIF in csrc/stage_driver.hip: launches skipped, results wrong on purpose).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/whatif_$label.err | tail -1 > gpurun_out/whatif_$label.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/whatif_$label.json")); print("$label", d["ms_per_step"], "ms", d["value"], "img/s", "host", d["config"].get("host_enqueue_ms_per_step"))
except Exception as e:
    print("$label", "failed", e)
PY
}
run warm C3D_WHATIF=0
run base C3D_WHATIF=0
run skip_fwdSE C3D_WHATIF=2
run skip_bwdSE C3D_WHATIF=8
run skip_bothSE C3D_WHATIF=10
run base2 C3D_WHATIF=0
