#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/fin_$label.err | tail -1 > gpurun_out/fin_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/fin_$label.json')); print('$label', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
run warm A=1
run cons1 C3D_FIN_CONSUMER=1
run cons0 C3D_FIN_CONSUMER=0
run grid512 C3D_BOF_GRID=512
run grid2048 C3D_BOF_GRID=2048
run grid256 C3D_BOF_GRID=256
run cons1b C3D_FIN_CONSUMER=1
