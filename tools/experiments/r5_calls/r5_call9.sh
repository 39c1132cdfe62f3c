#!/bin/bash
# round 5, GPU call 9: kink-robust strict tests (BCD / SCD / CC) with the matching-pursuit fit; CC fixtures with the corrected float64 yardstick; swish instruction probe
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/r5c9; mkdir -p $O
tools/micro/bin/valu_rate 2>&1 | grep -E "swish|v_fma_f32|v_exp_f32|v_rcp_f32|v_med3" > $O/valu_rate_swish.txt; cat $O/valu_rate_swish.txt
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_cc_gpu.py -q -s -m gpu -k "conditioned or cc_vs_reference_golden" > $O/pytest_strict.txt 2>&1; grep -E "^(bcd|scd|conditioned CC|CC s)|kink:|sigma of zero|passed|failed|Error" $O/pytest_strict.txt | cut -c1-330 | tail -60
