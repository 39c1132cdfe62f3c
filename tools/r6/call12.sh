#!/bin/bash
# round 6, call 12: full GPU suite + smoke at the library with pw_wgrad_v2 (chained), half-vector depthwise forward, advisor items
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c12
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r6c12/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6c12/smoke.txt
