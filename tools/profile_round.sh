#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun):
#   1. bench.py (default flags)                         -> gpurun_out/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same cmd -> gpurun_out/prof_stats/
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE need 3 + 2 of the 4 TCC slots) of one eager
#      step                                             -> gpurun_out/pmc_fetch/, gpurun_out/pmc_write/
# tools/summarize_rocprof.py turns 2 and 3 into the small JSON files committed under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 2000 gpurun_out/bench.json
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- \
    python bench.py --no-cpu-baseline --no-also > gpurun_out/prof_stats.log 2>&1
# same command with the side stream off: kernels run one at a time (the mode bench.py's per-kernel
# HIP-event table is taken in)
rm -rf gpurun_out/prof_stats_serial
C3D_WGRAD_SIDE=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_serial -- \
    python bench.py --no-cpu-baseline --no-also > gpurun_out/prof_stats_serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  C3D_WGRAD_SIDE=0 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python bench.py --no-cpu-baseline --no-also --no-graph --no-kernel-profile --steps 1 --warmup 1 > $d.log 2>&1
done
# MFMA utilisation of the pointwise kernels (own pass)
rm -rf gpurun_out/pmc_mfma
C3D_WGRAD_SIDE=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d gpurun_out/pmc_mfma -- python bench.py --no-cpu-baseline --no-also --no-graph --no-kernel-profile --steps 1 --warmup 1 > gpurun_out/pmc_mfma.log 2>&1
python tools/summarize_rocprof.py gpurun_out ${C3D_ROUND_TAG:-r04}
# host-side profile of the step loop (where the enqueue time goes)
timeout 600 python -m cProfile -o gpurun_out/host.prof bench.py --no-cpu-baseline --no-also --no-kernel-profile --steps 30 --warmup 5 > /dev/null 2> gpurun_out/host_prof.err
python -c "
import pstats
p = pstats.Stats('gpurun_out/host.prof'); p.sort_stats('cumulative').print_stats(45)" > gpurun_out/host_prof.txt 2>&1
ls -la gpurun_out/*.json
