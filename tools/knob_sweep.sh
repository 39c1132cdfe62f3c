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
run pw4wave C3D_PW_FORCE8=-1
run pw_force8_all C3D_PW_FORCE8=2
run pw_round C3D_PW_ROUND=1
run base2 A=1
run dwwg_nodot2 C3D_DWWG_NODOT2=1
run stem_dv_tpw8 C3D_STEM_DV_TPW=8
run head_tpw C3D_HEAD_TPW=2
run base3 A=1
