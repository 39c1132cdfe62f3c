#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_tests.sh
timeout 600 python bench.py --task cc --no-cpu-baseline --kernel-table gpurun_out/r2m_cc_kernels.json > gpurun_out/r2m_bench_cc.json 2> gpurun_out/r2m_bench_cc.err
tail -c 1200 gpurun_out/r2m_bench_cc.json; tail -3 gpurun_out/r2m_bench_cc.err
grep "kernels\]" gpurun_out/r2m_bench_cc.err | head -24
timeout 300 python -m change3d_amd.scripts.train_CC --batch_size 4 --max_steps 6 --print_freq 2 2>&1 | tail -4
