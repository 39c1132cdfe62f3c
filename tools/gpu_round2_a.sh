#!/bin/bash
# first GPU call of round 2: whole GPU suite, gradient error report, bench (+ world-2 launcher check on one GPU)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a_pytest.log
tail -5 gpurun_out/r2a_pytest.log
timeout 600 python tools/grad_error_report.py 64 > gpurun_out/grad_error_report.txt 2> gpurun_out/grad_error_report.err
head -12 gpurun_out/grad_error_report.txt
timeout 900 python bench.py --kernel-table gpurun_out/r2a_kernels.json > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
tail -c 1500 gpurun_out/r2a_bench.json
C3D_DIST_DEVICE=0 C3D_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/r2a_bench_dp2.json 2> gpurun_out/r2a_bench_dp2.err
echo "dp2 rc=$?"; tail -c 800 gpurun_out/r2a_bench_dp2.json; tail -5 gpurun_out/r2a_bench_dp2.err
