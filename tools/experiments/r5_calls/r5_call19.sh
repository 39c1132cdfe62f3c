#!/bin/bash
# round 5, GPU call 19: per-kernel tables of the B=16 workloads (CC, SCD) at the round-5 kernels; the new alignment test
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c19; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "misaligned or block_out" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for t in cc scd; do timeout 600 python bench.py --task $t --steps 20 --warmup 5 --no-cpu-baseline --no-also --kernel-table $O/kt_$t.json > $O/bench_$t.json 2> $O/bench_$t.err; grep "kernels\]" $O/bench_$t.err | head -24; done
