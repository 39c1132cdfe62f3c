#!/usr/bin/env python
"""Per-shape kernel microbenchmark (GPU box): times the real BCD layer shapes at B=32 through
the C ABI and prints us / algorithmic GB/s per launch.  Usage: python tools/bench_ops.py [family ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd import ops  # noqa: E402

DEV = "cuda:0"
B, T = int(os.environ.get("C3D_BENCH_B", "32")), 3   # C3D_BENCH_B=1 C3D_BENCH_DIV=16: tiny launches (fixed costs, under rocprofv3)
DIV = int(os.environ.get("C3D_BENCH_DIV", "1"))
DT = torch.bfloat16
dt = ops.dt_code(DT)
for _ov in filter(None, os.environ.get("C3D_BENCH_OPT", "").split(",")):   # C3D_BENCH_OPT=DW_RING=0,FOLD_SE=1: c3d_set_option before timing
    _n, _v = _ov.split("=")
    ops.set_option(getattr(ops, "OPT_" + _n.upper()), int(_v))
# (stage, H(in), Cin, Ci, Co) for identity blocks; block 0 has stride 2 from H*2
STAGES = [(1, 128 // DIV, 24, 54, 24), (2, 64 // DIV, 48, 108, 48), (3, 32 // DIV, 96, 216, 96)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def rt(*shape):
    return torch.randn(*shape, device=DEV).to(DT)


def report(name, us, nbytes):
    print(f"{name:58s} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / 1e6:7.1f} MB)", flush=True)


def pw_family():
    for st, H, Cin, Ci, Co in STAGES:
        M = B * T * H * H
        Cip = ops.cpad(Ci)
        x, a_, b_, c_ = rt(M, Cin), rt(M, Cip), rt(M, Cip), rt(M, Co)
        wa, wc = torch.randn(Ci, Cin, device=DEV) * 0.1, torch.randn(Co, Ci, device=DEV) * 0.1
        stats = torch.zeros(16 * 2 * 256, dtype=torch.float64, device=DEV)
        ss = torch.rand(2 * Cip, device=DEV)
        gate = torch.rand(B * Cip, device=DEV)
        coef3 = torch.rand(3 * Cip, device=DEV)
        coefo = torch.rand(3 * Co, device=DEV)
        nc3 = torch.zeros(B * Cip * 3, dtype=torch.float64, device=DEV)
        es = 2
        us = timeit(lambda: ops.pw_gemm(x, wa, a_, M=M, K=Cin, N=Ci, w_sn=Cin, w_sk=1, dtype=dt, epi_mode=ops.EPI_STATS, stats=stats))
        report(f"s{st} conv_a fwd   {Cin}->{Ci} M={M} NONE+STATS", us, M * (Cin + Cip) * es)
        us = timeit(lambda: ops.pw_gemm(b_, wc, c_, M=M, K=Ci, N=Co, w_sn=Ci, w_sk=1, dtype=dt, pro_mode=ops.PRO_BN_SE_SWISH,
                                        pro_p=ss, pro_gate=gate, rows_per_sample=T * H * H, epi_mode=ops.EPI_STATS, stats=stats))
        report(f"s{st} conv_c fwd   {Ci}->{Co} SWISH+STATS", us, M * (Cip + Co) * es)
        us = timeit(lambda: ops.pw_gemm(c_, wc, a_, M=M, K=Co, N=Ci, w_sn=1, w_sk=Ci, dtype=dt, x2=c_, pro_mode=ops.PRO_AFFINE2,
                                        pro_p=coefo, epi_mode=ops.EPI_SWISH_SE_BWD, e1=b_, epi_p=ss, epi_gate=gate, epi_q=ss,
                                        stats=nc3, rows_per_sample=T * H * H))
        report(f"s{st} conv_c bwd-d {Co}->{Ci} AFFINE2+SWISH_SE_BWD", us, M * (2 * Co + 2 * Cip) * es)
        us = timeit(lambda: ops.pw_gemm(a_, wa, x, M=M, K=Ci, N=Cin, w_sn=1, w_sk=Cin, dtype=dt, x2=b_, pro_mode=ops.PRO_AFFINE2,
                                        pro_p=coef3, epi_mode=ops.EPI_ADD, e1=c_ if Cin == Co else x, res_mode=0))
        report(f"s{st} conv_a bwd-d {Ci}->{Cin} AFFINE2+ADD", us, M * (2 * Cip + 2 * Cin) * es)
        dwc, dwa = torch.zeros(Co, Ci, device=DEV), torch.zeros(Ci, Cin, device=DEV)
        us = timeit(lambda: ops.pw_wgrad(c_, b_, dwc, M=M, K=Ci, N=Co, dw_sn=Ci, dw_sk=1, dtype=dt, p2=c_, p_coef=coefo,
                                         q_mode=ops.PRO_BN_SE_SWISH, q_ss=ss, q_gate=gate, rows_per_sample=T * H * H))
        report(f"s{st} conv_c wgrad N={Co} K={Ci}", us, M * (2 * Co + Cip) * es)
        us = timeit(lambda: ops.pw_wgrad(a_, x, dwa, M=M, K=Cin, N=Ci, dw_sn=Cin, dw_sk=1, dtype=dt, p2=b_, p_coef=coef3))
        report(f"s{st} conv_a wgrad N={Ci} K={Cin}", us, M * (2 * Cip + Cin) * es)


def dw_family():
    for st, H, Cin, Ci, Co in STAGES:
        Cip = ops.cpad(Ci)
        a_, b_, t1, t2 = (rt(B, T, H, H, Cip) for _ in range(4))
        w = torch.randn(Ci, 27, device=DEV) * 0.1
        ss = torch.rand(2 * Cip, device=DEV)
        nc = torch.zeros(B * Cip * 2, dtype=torch.float64, device=DEV)
        dsums = torch.zeros(2 * Ci, dtype=torch.float64, device=DEV)
        cA, cC, cB = torch.rand(Cip, device=DEV), torch.rand(Cip, device=DEV), torch.rand(B * Cip, device=DEV)
        dw = torch.zeros(Ci, 27, device=DEV)
        n = a_.numel() * 2
        us = timeit(lambda: ops.dw_fwd(a_, ss, w, b_, nc, B, T, H, H, Ci, 1, dt))
        report(f"s{st} dw fwd C={Ci} {H}x{H}", us, 2 * n)
        us = timeit(lambda: ops.dw_bwd_fused(t1, b_, cA, cB, cC, w, a_, ss, ss, t2, dsums, dw, B, T, H, H, Ci, dt, 1))
        report(f"s{st} dw bwd fused (data + weight gradient)", us, 4 * n)


if __name__ == "__main__":
    fams = sys.argv[1:] or ["pw", "dw"]
    if "pw" in fams:
        pw_family()
    if "dw" in fams:
        dw_family()


def dw_stride2_family():
    """First block of each stage: stride-2 depthwise conv on the previous stage's resolution."""
    for st, H, Ci in [(1, 256 // DIV, 54), (2, 128 // DIV, 108), (3, 64 // DIV, 216)]:
        Cip = ops.cpad(Ci)
        a_ = rt(B, T, H, H, Cip)
        b_ = rt(B, T, H // 2, H // 2, Cip)
        w = torch.randn(Ci, 27, device=DEV) * 0.1
        ss = torch.rand(2 * Cip, device=DEV)
        nc = torch.zeros(B * Cip * 2, dtype=torch.float64, device=DEV)
        us = timeit(lambda: ops.dw_fwd(a_, ss, w, b_, nc, B, T, H, H, Ci, 2, dt), iters=10)
        report(f"s{st} dw fwd stride 2 C={Ci} {H}x{H}", us, (a_.numel() + b_.numel()) * 2)


if "dws2" in sys.argv[1:]:
    dw_stride2_family()
