#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dw" > $O/pytest_dw.txt 2>&1; tail -5 $O/pytest_dw.txt
for r in 0 1 5 0 1 5; do echo "== DW_RING=$r"; C3D_BENCH_OPT=DW_RING=$r timeout 300 python tools/bench_ops.py dw 2>&1 | grep "bwd fused"; done > $O/bench_ops_dw.txt 2>&1
cat $O/bench_ops_dw.txt
for rep in 1 2; do for r in 0 1 5; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option DW_RING=$r 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BCD DW_RING=$r rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done | tee $O/step_ab.txt
for rep in 1 2; do for r in 0 1 5; do
  timeout 600 python bench.py --task scd --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-kernel-profile --option DW_RING=$r 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SCD DW_RING=$r rep $rep', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done | tee $O/step_ab_scd.txt
