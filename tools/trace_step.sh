#!/bin/bash
# rocprofv3 kernel trace of a short bench run + idle/overlap analysis (tools/trace_gaps.py)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -- python bench.py --no-cpu-baseline --no-kernel-profile --no-also --steps 10 --warmup 4 "$@" > gpurun_out/trace_step.log 2>&1
f=$(find gpurun_out/trace_step -name "*kernel_trace.csv" | xargs ls -S | head -1)
python tools/trace_gaps.py "$f" 8 | tee gpurun_out/trace_gaps.txt
python tools/trace_side.py "$f" | tee gpurun_out/trace_side.txt
tail -1 gpurun_out/trace_step.log | cut -c1-200
# keep the merged output small
find gpurun_out/trace_step -name "*.csv" -size +30M -delete
