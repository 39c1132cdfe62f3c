#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "conditioned_weights_every_gradient and scd" 2>&1 | grep -E "conditioned: worst|passed|failed" | cut -c1-400; }
run A=1
run A=2
run C3D_STEM_MFMA=0
run C3D_WGC_EARLY=0
run C3D_WGRAD_SIDE=0
run C3D_DWWG_SIDE_WGS=256
