#!/bin/bash
# round 6, call 1: pw_wgrad_v2 op tests, same-call A/B of C3D_OPT_PW_WGRAD_V2 (BCD), per-shape kernel table with v2
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c1
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -15 | tee gpurun_out/r6c1/pytest_wgrad.txt
bash tools/ab_option.sh PW_WGRAD_V2 0 1 2>&1 | tee gpurun_out/r6c1/ab.txt
for v in 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --option PW_WGRAD_V2=$v --kernel-table gpurun_out/r6c1/kt_v$v.json > gpurun_out/r6c1/bench_v$v.json 2> gpurun_out/r6c1/bench_v$v.err
done
python tools/kt_diff.py gpurun_out/r6c1/kt_v0.json gpurun_out/r6c1/kt_v1.json 0.02 | tee gpurun_out/r6c1/kt_diff.txt
