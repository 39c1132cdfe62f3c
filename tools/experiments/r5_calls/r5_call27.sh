#!/bin/bash
# round 5, call 27: full GPU suite + smoke at the library with the folded block-output backward, then the round profile
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/c27_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/c27_smoke.txt
bash tools/r5_call12.sh 2>&1 | tail -40
