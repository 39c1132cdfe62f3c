#!/bin/bash
# round 6, call 17: conv_c forward on the workgroup-cooperative kernel (C3D_OPT_PW_CFWD): stage test, A/B, per-shape table
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c17
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "cooperative or res_stage" 2>&1 | tail -12 | tee gpurun_out/r6c17/pytest.txt
bash tools/ab_option.sh PW_CFWD 0 1 2>&1 | tee gpurun_out/r6c17/ab.txt
for v in 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --option PW_CFWD=$v --kernel-table gpurun_out/r6c17/kt_v$v.json > gpurun_out/r6c17/bench_v$v.json 2> gpurun_out/r6c17/bench_v$v.err
done
python tools/kt_diff.py gpurun_out/r6c17/kt_v0.json gpurun_out/r6c17/kt_v1.json 0.02 | grep "pro=1\|^sum" | tee gpurun_out/r6c17/kt_diff.txt
