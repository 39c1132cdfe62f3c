#!/bin/bash
# What happens on the host and on the copy engines during the largest main-queue gaps of a train step?
# rocprofv3 kernel + memory-copy + HIP runtime API traces of a short bench run (no counters: trace-only), then
# tools/trace_bubbles.py lists every API call / copy / other-queue kernel that overlaps each of the largest gaps.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/trace_bub
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d gpurun_out/trace_bub -- python bench.py --no-cpu-baseline --no-kernel-profile --no-also --steps 8 --warmup 4 "$@" > gpurun_out/trace_bub.log 2>&1
python tools/trace_bubbles.py gpurun_out/trace_bub | tee gpurun_out/trace_bubbles.txt
find gpurun_out/trace_bub -name "*.csv" -size +20M -delete
