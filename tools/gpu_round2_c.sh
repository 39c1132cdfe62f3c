#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_tests.sh
for f in 1 0; do
  C3D_FOLD_BN=$f timeout 600 python tools/infer_bench.py 32 30 > gpurun_out/r2c_infer_fold$f.json 2> gpurun_out/r2c_infer_fold$f.err
  echo "FOLD_BN=$f: $(cat gpurun_out/r2c_infer_fold$f.json)"
done
