#!/usr/bin/env python
"""Run ONE kernel shape a few times (for rocprofv3 --pmc).  usage: one_kernel.py dwfwd|dwbwd|dwwgrad|pwa|pwc [stage]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd import ops
DEV = "cuda:0"; B, T = 32, 3; DT = torch.bfloat16; dt = ops.dt_code(DT)
STAGES = {1: (128, 24, 54, 24), 2: (64, 48, 108, 48), 3: (32, 96, 216, 96)}
which = sys.argv[1]; st = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, Cin, Ci, Co = STAGES[st]; Cip = ops.cpad(Ci); M = B * T * H * H
rt = lambda *s: torch.randn(*s, device=DEV).to(DT)
a_, b_, t1, t2 = (rt(B, T, H, H, Cip) for _ in range(4))
w = torch.randn(Ci, 27, device=DEV) * 0.1; ss = torch.rand(2 * Cip, device=DEV)
nc = torch.zeros(B * Cip * 2, dtype=torch.float64, device=DEV); ds = torch.zeros(16 * 2 * Ci, dtype=torch.float64, device=DEV)
cA, cC, cB = torch.rand(Cip, device=DEV), torch.rand(Cip, device=DEV), torch.rand(B * Cip, device=DEV)
dw = torch.zeros(Ci, 27, device=DEV)
x = rt(M, Cin); wa = torch.randn(Ci, Cin, device=DEV) * 0.1; wc = torch.randn(Co, Ci, device=DEV) * 0.1
stats = torch.zeros(16 * 2 * 256, dtype=torch.float64, device=DEV); gate = torch.rand(B * Cip, device=DEV)
for _ in range(3):
    if which == "dwfwd": ops.dw_fwd(a_, ss, w, b_, nc, B, T, H, H, Ci, 1, dt)
    elif which == "dwbwd": ops.dw_bwd_fused(t1, b_, cA, cB, cC, w, a_, ss, ss, t2, ds, dw, B, T, H, H, Ci, dt, 1)
    elif which == "pwa": ops.pw_gemm(x, wa, a_.view(M, Cip), M=M, K=Cin, N=Ci, w_sn=Cin, w_sk=1, dtype=dt, epi_mode=ops.EPI_STATS, stats=stats)
    elif which == "pwc": ops.pw_gemm(b_.view(M, Cip), wc, x, M=M, K=Ci, N=Co, w_sn=Ci, w_sk=1, dtype=dt, pro_mode=ops.PRO_BN_SE_SWISH, pro_p=ss, pro_gate=gate, rows_per_sample=T * H * H, epi_mode=ops.EPI_STATS, stats=stats)
torch.cuda.synchronize()
if which in ("pwwg", "pwwgc"):
    coef3 = torch.rand(3 * Cip, device=DEV); coefo = torch.rand(3 * Co, device=DEV)
    dwa, dwc = torch.zeros(Ci, Cin, device=DEV), torch.zeros(Co, Ci, device=DEV)
    for _ in range(3):
        if which == "pwwg": ops.pw_wgrad(a_.view(M, Cip), x, dwa, M=M, K=Cin, N=Ci, dw_sn=Cin, dw_sk=1, dtype=dt, p2=b_.view(M, Cip), p_coef=coef3)
        else: ops.pw_wgrad(x if Cin == Co else rt(M, Co), b_.view(M, Cip), dwc, M=M, K=Ci, N=Co, dw_sn=Ci, dw_sk=1, dtype=dt, p2=x if Cin == Co else rt(M, Co), p_coef=coefo, q_mode=ops.PRO_BN_SE_SWISH, q_ss=ss, q_gate=gate, rows_per_sample=T * H * H)
    torch.cuda.synchronize()
