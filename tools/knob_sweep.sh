#!/bin/bash
# environment-knob sweep around the defaults through the INSTRUMENTED build (python __graft_entry__.py --tuning);
# each label: one 40-step bench, defaults interleaved.  Usage: tools/knob_sweep.sh "label VAR=val [VAR=val]" ...
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export C3D_LIB=$(pwd)/change3d_amd/lib/libchange3d_hip_tune.so
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 5 ${C3D_SWEEP_ARGS:-} 2> gpurun_out/ks_$label.err | tail -1 > gpurun_out/ks_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/ks_$label.json')); print('$label', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ks_$label.err; }
run warm A=1
run base A=1
for spec in "$@"; do
  set -- $spec
  run "$@"
done
run base_end A=1
