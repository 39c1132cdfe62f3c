#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10 2> gpurun_out/sw_$label.err | tail -1 > gpurun_out/sw_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/sw_$label.json')); print('$label', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
run warm A=1
run base A=1
run over200 C3D_PW_OVERSUB=200
run over150 C3D_PW_OVERSUB=150
run over75 C3D_PW_OVERSUB=75
run base2 A=1
run over400 C3D_PW_OVERSUB=400
run dwlate C3D_DWWG_EARLY=0
run base3 A=1
