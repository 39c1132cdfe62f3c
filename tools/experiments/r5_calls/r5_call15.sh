#!/bin/bash
# round 5, GPU call 15: new side-stream width / walk cap defaults, f32 dense rows on the buffer-addressed loop -- op + model tests, A/B vs the round-4 library, f32 / SCD / CC benches
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c15; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_pw_wg_gpu.py tests/test_wide_ops_gpu.py -x -q -m gpu > $O/pytest_ops.txt 2>&1; tail -2 $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -2 $O/pytest_model.txt
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so 2>&1 | tee $O/ab.txt
bash tools/ab_lib.sh libchange3d_hip_base.so libchange3d_hip.so --dtype f32 2>&1 | tee $O/ab_f32.txt
for t in scd cc; do timeout 600 python bench.py --task $t --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], 'ms', d['value'], 'img/s')"; done | tee $O/tasks.txt
