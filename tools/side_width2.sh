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
run pw208 C3D_WG_BLOCKS=208
run pw192 C3D_WG_BLOCKS=192
run pw176 C3D_WG_BLOCKS=176
run pw160 C3D_WG_BLOCKS=160
run base2 A=1
run pw144 C3D_WG_BLOCKS=144
run pw128 C3D_WG_BLOCKS=128
run pw192_dw112 C3D_WG_BLOCKS=192 C3D_DWWG_SIDE_WGS=112
run pw192_dw144 C3D_WG_BLOCKS=192 C3D_DWWG_SIDE_WGS=144
run pw176_dw144 C3D_WG_BLOCKS=176 C3D_DWWG_SIDE_WGS=144
run pw192_ring2 C3D_WG_BLOCKS=192 C3D_BWD_RING=2
run base3 A=1
