#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 2> gpurun_out/sw_$label.err | tail -1 > gpurun_out/sw_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/sw_$label.json')); print('$label', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
run warm A=1
run base A=1
run pw224 C3D_WG_BLOCKS=224
run pw192 C3D_WG_BLOCKS=192
run pw160 C3D_WG_BLOCKS=160
run dw112 C3D_DWWG_SIDE_WGS=112
run dw144 C3D_DWWG_SIDE_WGS=144
run pw192_dw144 C3D_WG_BLOCKS=192 C3D_DWWG_SIDE_WGS=144
run base2 A=1
