#!/bin/bash
# round 6, call 14: stride-2 depthwise forward on eight waves (C3D_OPT_DW_FWD_HV bit 2): op tests, A/B (1 = off, 5 = on), per-shape table
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c14
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "dw" 2>&1 | tail -5 | tee gpurun_out/r6c14/pytest_dw.txt
bash tools/ab_option.sh DW_FWD_HV 1 5 2>&1 | tee gpurun_out/r6c14/ab.txt
for v in 1 5; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --option DW_FWD_HV=$v --kernel-table gpurun_out/r6c14/kt_v$v.json > gpurun_out/r6c14/bench_v$v.json 2> gpurun_out/r6c14/bench_v$v.err
done
python tools/kt_diff.py gpurun_out/r6c14/kt_v1.json gpurun_out/r6c14/kt_v5.json 0.02 | grep "dw333_fwd\|^sum" | tee gpurun_out/r6c14/kt_diff.txt
