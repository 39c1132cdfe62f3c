#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -- python bench.py --no-cpu-baseline --no-kernel-profile --no-also --steps 10 --warmup 4 > gpurun_out/trace_step.log 2>&1
f=$(find gpurun_out/trace_step -name "*kernel_trace.csv" | xargs ls -S | head -1)
python tools/trace_gaps.py "$f" 8 > gpurun_out/trace_gaps.txt
python tools/trace_window.py "$f" 700 200 0 | tee gpurun_out/trace_window0.txt
python tools/trace_window.py "$f" 150 150 1 | tee gpurun_out/trace_window1.txt
python tools/trace_window.py "$f" 150 150 2 | tee gpurun_out/trace_window2.txt
find gpurun_out/trace_step -name "*.csv" -size +30M -delete
