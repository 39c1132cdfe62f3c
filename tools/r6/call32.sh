#!/bin/bash
# cooperative conv_c data + weight gradient: op tests, stage / model tests, step A/B (C3D_OPT_PW_CDG 1 = conv_a only, 3 = both)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pw_wg_gpu.py -x -q -m gpu 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "stage or folded or end_to_end or golden" 2>&1 | tail -6
bash tools/ab_option.sh PW_CDG 1 3
