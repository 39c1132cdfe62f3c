#!/bin/bash
# BatchNorm_b of conv_c's input from the batch totals the depthwise forward accumulates (C3D_OPT_PW_CFWD bit 2): tests, A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_c_forward or conv_a_forward or dw" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_cpu.py -x -q -m gpu -k "stage or folded or end_to_end or golden or cooperative" 2>&1 | tail -4
bash tools/ab_option.sh PW_CFWD 3 7
