#!/bin/bash
# conv_a forward on the cooperative kernel: op-level bit identity, then the step A/B (C3D_OPT_PW_CFWD 1 = conv_c only, 3 = both)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_a_forward or conv_c_forward" 2>&1 | tail -15
bash tools/ab_option.sh PW_CFWD 1 3
