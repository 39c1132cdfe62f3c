#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
C3D_DW_TZ=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "dw333_fwd_bwd" 2>&1 | tail -2
for m in 0 1 2 4; do echo "DBG=$m"; C3D_DW_TZ=1 C3D_DW_TZ_DBG=$m timeout 300 python tools/bench_ops.py dw 2>&1 | grep -E "dw fwd" | cut -c1-90; done
