#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out/r5c4; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_pw_wg_gpu.py -x -q -m gpu > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage or flag or e2e" > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
C3D_SWEEP_ARGS="--no-also" bash tools/knob_sweep.sh "skew2 C3D_PW_SKEW=2" "skew4 C3D_PW_SKEW=4" "skew8 C3D_PW_SKEW=8" "skew16 C3D_PW_SKEW=16" 2>&1 | tee $O/skew.txt
for rep in 1 2; do for r in 5 13; do
  timeout 600 python bench.py --task scd --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option DW_RING=$r 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SCD DW_RING=$r rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option DW_RING=$r 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BCD DW_RING=$r rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done | tee $O/ring_small.txt
python tools/pw_phase_clock.py --fb --ring=13 > $O/fb_ring13.txt 2>&1; grep -v "^/opt" $O/fb_ring13.txt | head -34
