#!/usr/bin/env python
"""c3d_pw_wgrad at the residual-stage shapes (B=32, bf16): today's fused-prologue modes against the plain mode (operands
already in GEMM form): what a VALU-free weight-gradient kernel could gain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from change3d_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
B, T = 32, 3
shapes = [("res2", 128, 24, 54), ("res3", 64, 48, 108), ("res4", 32, 96, 216)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, hw, co, ci in shapes:
    M = B * T * hw * hw
    cip, cop = ops.cpad(ci), ops.cpad(co)
    t2 = torch.randn(M, cip, device=dev).to(dt)
    a = torch.randn(M, cip, device=dev).to(dt)
    x = torch.randn(M, cop, device=dev).to(dt)
    g = torch.randn(M, cop, device=dev).to(dt)
    c = torch.randn(M, cop, device=dev).to(dt)
    b = torch.randn(M, cip, device=dev).to(dt)
    coef_a = torch.randn(3 * cip, device=dev)
    coef_c = torch.randn(3 * cop, device=dev)
    ss_b = torch.randn(2 * cip, device=dev)
    gate = torch.rand(B * cip, device=dev)
    dwa = torch.zeros(ci, co, device=dev)
    dwc = torch.zeros(co, ci, device=dev)
    d = ops.dt_code(dt)
    rps = T * hw * hw
    r = {}
    r["a_fused"] = timeit(lambda: ops.pw_wgrad(t2, x, dwa, M=M, K=co, N=ci, dw_sn=co, dw_sk=1, dtype=d, p2=a, p_coef=coef_a))
    r["a_plain"] = timeit(lambda: ops.pw_wgrad(t2, x, dwa, M=M, K=co, N=ci, dw_sn=co, dw_sk=1, dtype=d))
    r["c_fused"] = timeit(lambda: ops.pw_wgrad(g, b, dwc, M=M, K=ci, N=co, dw_sn=ci, dw_sk=1, dtype=d, p2=c, p_coef=coef_c,
                                               q_mode=ops.PRO_BN_SE_SWISH, q_ss=ss_b, q_gate=gate, rows_per_sample=rps))
    r["c_plain"] = timeit(lambda: ops.pw_wgrad(g, b, dwc, M=M, K=ci, N=co, dw_sn=ci, dw_sk=1, dtype=d))
    by = M * (cip + cop) * 2
    print(name, f"M={M}", {k: round(v, 1) for k, v in r.items()}, "us;  plain bytes", by // 1000000, "MB ->",
          {k: round(by / v / 1e6, 2) for k, v in r.items() if 'plain' in k}, "TB/s")
