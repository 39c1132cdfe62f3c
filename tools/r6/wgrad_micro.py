#!/usr/bin/env python
"""Round-6 micro-benchmark of c3d_pw_wgrad on the res4 layers (B=32, 32 x 32 maps, bf16): HIP-event time per call
(kernel + reducer) for a list of C3D_OPT_PW_WGRAD_V2 values.  usage: wgrad_micro.py [v ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from change3d_amd import ops
DEV = "cuda:0"; DT = torch.bfloat16; dt = ops.dt_code(DT)
B, T, H = 32, 3, 32
M = B * T * H * H
Ci, Co = 216, 96
rt = lambda *s: torch.randn(*s, device=DEV).to(DT)
t2, a_, b_ = rt(M, Ci), rt(M, Ci), rt(M, Ci)
g, c, x = rt(M, Co), rt(M, Co), rt(M, Co)
coef_a, coef_c = torch.rand(3 * Ci, device=DEV), torch.rand(3 * Co, device=DEV)
ss = torch.rand(2 * Ci, device=DEV); gate = torch.rand(B * Ci, device=DEV)
dwa, dwc = torch.zeros(Ci, Co, device=DEV), torch.zeros(Co, Ci, device=DEV)
def conv_a(): ops.pw_wgrad(t2, x, dwa, M=M, K=Co, N=Ci, dw_sn=Co, dw_sk=1, dtype=dt, p2=a_, p_coef=coef_a)
def conv_c(): ops.pw_wgrad(g, b_, dwc, M=M, K=Ci, N=Co, dw_sn=Ci, dw_sk=1, dtype=dt, p2=c, p_coef=coef_c, q_mode=ops.PRO_BN_SE_SWISH,
                           q_ss=ss, q_gate=gate, rows_per_sample=T * H * H)
vals = [int(v) for v in sys.argv[1:]] or [0, 1]
for name, fn in (("conv_a wgrad N=216 K=96", conv_a), ("conv_c wgrad N=96 K=216", conv_c)):
    for v in vals:
        ops.set_option(ops.OPT_PW_WGRAD_V2, v)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}  opt={v:3d}  {e0.elapsed_time(e1) / n * 1e3:7.1f} us per call")
