#!/usr/bin/env python
"""MFMA stem kernels vs the scalar-FMA ones on the same inputs (c3d_set_option C3D_OPT_STEM_MFMA toggled per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from change3d_amd import ops

DEV = "cuda:0"
def run(mfma, T, dtype, B=2, H=40, W=72):
    ops.set_option(ops.OPT_STEM_MFMA, 1 if mfma else 0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, T, H, W, generator=g).to(DEV)
    w_t = (torch.randn(24, 3, 1, 3, 3, generator=g) * 0.3).to(DEV)
    w_xy = (torch.randn(24, 1, 5, 1, 1, generator=g) * 0.5).to(DEV)
    dt = ops.dt_code(dtype)
    u = torch.zeros(B, T, H, W, 24, dtype=dtype, device=DEV)
    sums = torch.zeros(48, dtype=torch.float64, device=DEV)
    ops.stem_fwd(x, w_t, w_xy, u, sums, B, T, H, W, dt)
    g0 = torch.randn(B, T, H, W, 24, generator=g).to(DEV).to(dtype)
    coef = torch.randn(72, generator=g).to(DEV)
    dv = torch.zeros_like(u)
    dw_xy = torch.zeros(24, 5, device=DEV)
    ops.stem_bwd_dv(x, w_t, w_xy, g0, u, coef, dv, dw_xy, B, T, H, W, dt)
    dw_t = torch.zeros(24, 27, device=DEV)
    dP = torch.zeros(B, 3, T, H, W, device=DEV)
    ops.stem_bwd_wx(x, w_t, dv, dw_t, dP, B, T, H, W, 0, T, True, dt)
    dP2 = torch.zeros(3, T - 2, H, W, device=DEV)
    dw_t2 = torch.zeros(24, 27, device=DEV)
    ops.stem_bwd_wx(x, w_t, dv, dw_t2, dP2, B, T, H, W, 1, T - 2, False, dt)
    torch.cuda.synchronize()
    return dict(u=u.float(), sums=sums.float(), dv=dv.float(), dw_xy=dw_xy, dw_t=dw_t, dP=dP, dP2=dP2, dw_t2=dw_t2)

import itertools
for (T, dtype, (B, H, W)) in itertools.product((3, 5), (torch.float32, torch.bfloat16), ((2, 64, 64), (2, 40, 72), (3, 128, 128))):
    if True:
        a, b = run(True, T, dtype, B, H, W), run(False, T, dtype, B, H, W)
        out = []
        for k in a:
            d = (a[k] - b[k]).abs().max().item()
            s = b[k].abs().max().item()
            out.append(f"{k} {d / max(s, 1e-30):.1e}")
        print(T, dtype, (B, H, W), " ".join(out))
