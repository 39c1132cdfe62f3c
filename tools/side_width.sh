#!/bin/bash
# sweep of the side-stream kernel widths / ring depth (interleaved with the default configuration)
set -u
# environment knobs exist in the instrumented build only (python __graft_entry__.py --tuning)
cd "${GRAFT_REPO_ROOT:-.}"
export C3D_LIB=${C3D_LIB:-$(pwd)/change3d_amd/lib/libchange3d_hip_tune.so}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local label=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10 2> gpurun_out/sw_$label.err | tail -1 > gpurun_out/sw_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/sw_$label.json')); print('$label', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
run warm A=1
run base A=1
run pw224 C3D_WG_BLOCKS=224
run pw192 C3D_WG_BLOCKS=192
run dw96 C3D_DWWG_SIDE_WGS=96
run dw112 C3D_DWWG_SIDE_WGS=112
run base2 A=1
run dw144 C3D_DWWG_SIDE_WGS=144
run dw160 C3D_DWWG_SIDE_WGS=160
run ring2 C3D_BWD_RING=2
run ring4 C3D_BWD_RING=4
run prio C3D_SIDE_PRIO=1
run base3 A=1
