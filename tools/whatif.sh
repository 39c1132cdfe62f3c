#!/bin/bash
# Upper bounds for launch folding (C3D_WHATIF in csrc/stage_driver.hip: launches skipped, results wrong on purpose).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in 0 1 2 4 3 7; do
  C3D_WHATIF=$w python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/whatif_$w.err | tail -1 > gpurun_out/whatif_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/whatif_$w.json"))
print("WHATIF=$w", d["ms_per_step"], "ms", d["value"], "img/s", "host", d["config"].get("host_enqueue_ms_per_step"))
PY
done
