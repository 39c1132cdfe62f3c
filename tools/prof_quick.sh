#!/bin/bash
# Quick per-kernel profile on the GPU box: tools/prof_quick.sh <tag> [bench.py args...]  (environment knobs pass through)
#   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, side stream as configured by the caller)
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
root=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_$tag
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- \
    python $root/bench.py --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 2 "$@" > /tmp/prof_$tag.log 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
tail -1 /tmp/prof_$tag.log | cut -c1-300
