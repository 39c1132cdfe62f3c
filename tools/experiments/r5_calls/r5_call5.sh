#!/bin/bash
# round 5, GPU call 5: state after container re-creation -- VALU rate probe, default bench with kernel table, whole GPU suite
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c5; mkdir -p $O
tools/micro/bin/valu_rate > $O/valu_rate.txt 2>&1
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --kernel-table $O/kt.json > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
C3D_TEST_TIMEOUT=2400 bash tools/gpu_tests.sh; cp gpurun_out/pytest_gpu.log $O/
