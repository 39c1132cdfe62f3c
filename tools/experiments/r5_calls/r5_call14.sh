#!/bin/bash
# round 5, GPU call 14: confirm the two knobs that moved (side-stream weight-gradient width, fused depthwise backward walk cap), alone and together, repeated
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c14; mkdir -p $O
C3D_SWEEP_ARGS="--no-also" bash tools/knob_sweep.sh "side160 C3D_PWWG_SIDE_WGS=160" "dwbf64 C3D_DWBF_MAX=64" "both C3D_PWWG_SIDE_WGS=160 C3D_DWBF_MAX=64" "base2 A=1" \
  "side144 C3D_PWWG_SIDE_WGS=144" "side176 C3D_PWWG_SIDE_WGS=176" "both_b C3D_PWWG_SIDE_WGS=160 C3D_DWBF_MAX=64" "dwbf128 C3D_DWBF_MAX=128" "base3 A=1" \
  "both_c C3D_PWWG_SIDE_WGS=160 C3D_DWBF_MAX=64" "both144 C3D_PWWG_SIDE_WGS=144 C3D_DWBF_MAX=64" 2>&1 | tee $O/sweep.txt
C3D_SWEEP_ARGS="--no-also --task scd" bash tools/knob_sweep.sh "scd_both C3D_PWWG_SIDE_WGS=160 C3D_DWBF_MAX=64" "scd_dwbf64 C3D_DWBF_MAX=64" 2>&1 | tee $O/sweep_scd.txt
C3D_SWEEP_ARGS="--no-also --task cc" bash tools/knob_sweep.sh "cc_both C3D_PWWG_SIDE_WGS=160 C3D_DWBF_MAX=64" 2>&1 | tee $O/sweep_cc.txt
rm -f gpurun_out/ks_*.err
