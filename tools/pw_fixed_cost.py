#!/usr/bin/env python
"""Fixed per-launch cost of c3d_pw_gemm: tiny-M launches under `rocprofv3 --kernel-trace` (GPU-side
durations; host timing of such short kernels only measures the Python launch rate).
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python tools/pw_fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd import ops  # noqa: E402

DEV, DT = "cuda:0", torch.bfloat16
dt = ops.dt_code(DT)
for (K, N) in [(96, 216), (24, 54)]:
    for M in [16, 4096, 98304]:
        x = torch.randn(M, K, device=DEV).to(DT)
        y = torch.empty(M, ops.cpad(N), device=DEV, dtype=DT)
        w = torch.randn(N, K, device=DEV) * 0.1
        stats = torch.zeros(16 * 2 * 256, dtype=torch.float64, device=DEV)
        for epi in (ops.EPI_STORE, ops.EPI_STATS):
            for _ in range(20):
                ops.pw_gemm(x, w, y, M=M, K=K, N=N, w_sn=K, w_sk=1, dtype=dt, epi_mode=epi,
                            stats=stats if epi == ops.EPI_STATS else None)
            torch.cuda.synchronize()
