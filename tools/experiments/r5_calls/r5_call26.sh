#!/bin/bash
# round 5, call 26: fused depthwise backward with 4 x 8 tiles / 256 threads (two workgroups per CU) against the shipped 8 x 8 / 512
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=$(pwd)/change3d_amd/lib
C3D_LIB=$L/libchange3d_hip_fb4.so timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "dw333_fwd_bwd or dw333_backward_walks" 2>&1 | tail -4 | tee gpurun_out/c26_ops.txt
run() {  # name lib options...
  local name=$1 lib=$2; shift 2
  C3D_LIB=$L/$lib timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also "$@" 2> gpurun_out/c26_$name.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  grep "c3d_dw333_bwd_fused" gpurun_out/c26_$name.err
}
for rep in 1 2; do
  run base libchange3d_hip.so
  run base_noring libchange3d_hip.so --option DW_RING=0
  run fb4 libchange3d_hip_fb4.so
done 2>&1 | tee gpurun_out/c26_ab.txt
