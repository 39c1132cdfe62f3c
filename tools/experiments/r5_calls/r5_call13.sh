#!/bin/bash
# round 5, GPU call 13: knob sweep through the instrumented build now that the pointwise kernels' waits are exact (the defaults were tuned against the round-4 kernels)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c13; mkdir -p $O
C3D_SWEEP_ARGS="--no-also" bash tools/knob_sweep.sh "f8m1 C3D_PW_FORCE8=-1" "f8_2 C3D_PW_FORCE8=2" "f8_1 C3D_PW_FORCE8=1" "round1 C3D_PW_ROUND=1" \
  "side128 C3D_PWWG_SIDE_WGS=128" "side160 C3D_PWWG_SIDE_WGS=160" "side256 C3D_PWWG_SIDE_WGS=256" "prio1 C3D_SIDE_PRIO=1" \
  "bob256 C3D_BOB_GRID=256" "bob512 C3D_BOB_GRID=512" "dwbf16 C3D_DWBF_MAX=16" "dwbf64 C3D_DWBF_MAX=64" "ring2 C3D_BWD_RING=2" "ring4 C3D_BWD_RING=4" \
  "wgmt64 C3D_WG_MT=64" "dwtpw C3D_DW_TPW=16" 2>&1 | tee $O/sweep.txt
rm -f gpurun_out/ks_*.err
