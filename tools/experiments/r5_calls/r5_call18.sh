#!/bin/bash
# round 5, GPU call 18: B=16 workloads (SCD, CC) against the pointwise workgroup-shape knobs (their res4 launches are 1.5 tiles per wave)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c18; mkdir -p $O
C3D_SWEEP_ARGS="--no-also --task scd" bash tools/knob_sweep.sh "scd_f8m1 C3D_PW_FORCE8=-1" "scd_round1 C3D_PW_ROUND=1" "scd_base2 A=1" "scd_f8_2 C3D_PW_FORCE8=2" "scd_base3 A=1" 2>&1 | tee $O/sweep_scd.txt
C3D_SWEEP_ARGS="--no-also --task cc" bash tools/knob_sweep.sh "cc_f8m1 C3D_PW_FORCE8=-1" "cc_round1 C3D_PW_ROUND=1" "cc_base2 A=1" "cc_f8_2 C3D_PW_FORCE8=2" "cc_side96 C3D_PWWG_SIDE_WGS=96" "cc_bob256 C3D_BOB_GRID=256" 2>&1 | tee $O/sweep_cc.txt
rm -f gpurun_out/ks_*.err
