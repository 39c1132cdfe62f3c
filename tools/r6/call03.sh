#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c3
python tools/r6/wgrad_micro.py 0 1 3 5 7 9 2>&1 | tee gpurun_out/r6c3/micro.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/tools/r6/wgrad_micro.py 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1); head -8 "$f" | tee gpurun_out/r6c3/stats.txt
