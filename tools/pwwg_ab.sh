#!/bin/bash
# A/B of the side-stream width of the pointwise weight gradient (C3D_PWWG_SIDE_WGS): BCD, SCD, CC, B=16
set -u
# environment knobs exist in the instrumented build only (python __graft_entry__.py --tuning)
cd "${GRAFT_REPO_ROOT:-.}"
export C3D_LIB=${C3D_LIB:-$(pwd)/change3d_amd/lib/libchange3d_hip_tune.so}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -k "pw_wgrad or e2e or res_stage or reproducible" 2>&1 | tail -3
run() { local label=$1; shift; env "$@" 2> gpurun_out/pwwg_$label.err | tail -1 > gpurun_out/pwwg_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/pwwg_$label.json')); print('$label', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run bcd_256_$rep C3D_PWWG_SIDE_WGS=256 python bench.py --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10
run bcd_192_$rep python bench.py --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10
done
run scd_256 C3D_PWWG_SIDE_WGS=256 python bench.py --task scd --no-cpu-baseline --no-kernel-profile
run scd_192 python bench.py --task scd --no-cpu-baseline --no-kernel-profile
run cc_256 C3D_PWWG_SIDE_WGS=256 python bench.py --task cc --no-cpu-baseline --no-kernel-profile
run cc_192 python bench.py --task cc --no-cpu-baseline --no-kernel-profile
run b16_256 C3D_PWWG_SIDE_WGS=256 python bench.py --batch 16 --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10
run b16_192 python bench.py --batch 16 --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10
run b16_128 C3D_PWWG_SIDE_WGS=128 python bench.py --batch 16 --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 10
