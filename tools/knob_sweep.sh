#!/bin/bash
# environment-knob sweep around the defaults (each label: one 60-step bench), defaults interleaved
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10 2> gpurun_out/ks_$label.err | tail -1 > gpurun_out/ks_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/ks_$label.json')); print('$label', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ks_$label.err; }
run warm A=1
run base A=1
run dwbd_tpw8 C3D_DWBD_TPW=8
run dwbd_tpw12 C3D_DWBD_TPW=12
run dwbd_tpw6 C3D_DWBD_TPW=6
run base2 A=1
run dwbd_tpw8b C3D_DWBD_TPW=8
run dwbd_tpw10 C3D_DWBD_TPW=10
run dwbd8_dwf12 C3D_DWBD_TPW=8 C3D_DW_TPW=12
run base3 A=1
