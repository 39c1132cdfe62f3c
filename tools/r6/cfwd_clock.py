#!/usr/bin/env python
"""Phase clocks of the cooperative pointwise forward kernel (csrc/pw_cfwd.hip built with -DC3D_CF_CLOCK:
tools/r6/mkvariant.sh cfclk pw_cfwd.hip "-DC3D_CF_CLOCK -Wno-dangling-else"; run with C3D_LIB=.../libchange3d_hip_cfclk.so).
Per layer shape of the B=32 step: s_memtime (100 MHz) deltas of wave 0, averaged over the workgroups, and the HIP-event time."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from change3d_amd import ops, _lib as L
DEV = "cuda:0"
h = L.lib()
rd = h.c3d_debug_cf_clock; rd.restype = C.c_int; rd.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (1024 * 8))()
B, T = 32, 3
ptr = lambda t: t.data_ptr()
def run(kind, K, N, H, se):
    rps = T * H * H; M = B * rps
    Kp, Np = ops.cpad(K), ops.cpad(N)
    x = torch.randn(M, Kp, device=DEV).to(torch.bfloat16); x2 = torch.randn(M, Kp, device=DEV).to(torch.bfloat16)
    w = torch.randn(N, K, device=DEV) * 0.1
    img = torch.empty(ops.pw_weight_image_bytes(N, K, ops.DT_BF16), dtype=torch.uint8, device=DEV)
    ops.pw_pack_weights([(w, img, N, K, K, 1)], ops.DT_BF16)
    y = torch.empty(M, Np, device=DEV, dtype=torch.bfloat16); po = torch.empty(M, Kp, device=DEV, dtype=torch.bfloat16)
    stats = torch.zeros(ops.STAT_STRIPES * 2 * N, dtype=torch.float64, device=DEV)
    gamma, beta = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    rm, rv, nbt = torch.zeros(K, device=DEV), torch.ones(K, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    ss, mr = torch.zeros(2 * Kp, device=DEV), torch.zeros(2 * Kp, device=DEV)
    Cr = 16
    w1, b1, w2, b2 = torch.randn(Cr, K, device=DEV) * 0.1, torch.zeros(Cr, device=DEV), torch.randn(K, Cr, device=DEV) * 0.1, torch.zeros(K, device=DEV)
    gate, hid = torch.zeros(B, Kp, device=DEV), torch.zeros(B, Cr, device=DEV)
    a = L.PwArgs()
    a.x, a.y, a.w, a.w_img = ptr(x), ptr(y), ptr(w), ptr(img)
    a.M, a.K, a.Kp, a.N, a.Np, a.w_sn, a.w_sk = M, K, Kp, N, Np, K, 1
    a.dtype, a.epi_mode, a.pro_p, a.stats = ops.DT_BF16, ops.EPI_STATS, ptr(ss), ptr(stats)
    f = L.BnFin()
    f.gamma, f.beta, f.running_mean, f.running_var, f.nbt, f.ss, f.mr = ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(ss), ptr(mr)
    f.count, f.momentum, f.eps, f.training = float(M), 0.1, 1e-5, 1
    if kind == "conv_c":
        xs = x.float().view(B, rps, Kp).double()
        sums = torch.stack([xs.sum(1), (xs * xs).sum(1)], dim=2).contiguous()
        a.pro_mode, a.rows_per_sample, f.batch = ops.PRO_BN_SE_SWISH, rps, B
        if se:
            a.pro_gate = ptr(gate)
            a.se_w1, a.se_b1, a.se_w2, a.se_b2, a.se_hid, a.se_cr = ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hid), Cr
    else:
        xd = x.float().double()[:, :K]
        tot = torch.stack([xd.sum(0), (xd * xd).sum(0)])
        sums = (tot[None] / ops.STAT_STRIPES).repeat(ops.STAT_STRIPES, 1, 1).contiguous()
        a.pro_mode, a.x2, a.pro_out, f.batch = ops.PRO_AFFINE2, ptr(x2), ptr(po), 0
    f.sums = ptr(sums)
    a.fin = f
    st = ops._stream()
    for _ in range(5): assert h.c3d_pw_gemm(C.byref(a), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): h.c3d_pw_gemm(C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    assert rd(buf) == 0
    import numpy as np
    c = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    live = c[:, 4] > c[:, 0]
    c = c[live]
    d = lambda i, j: ((c[:, i] - c[:, j]).mean() / 100.0)   # s_memtime runs at 100 MHz
    span = (c[:, 4].max() - c[:, 0].min()) / 100.0
    print(f"{kind} K={K} N={N} H={H} se={int(se)}: {us:6.1f} us/launch | wgs {live.sum():3d}  first entry -> last exit {span:6.1f} us | "
          f"setup {d(6,0):5.2f} bn {(d(7,6) if kind == 'conv_c' else d(5,6)):5.2f} gate {(d(5,7) if kind == 'conv_c' else 0):5.2f}  wait+barrier {d(1,5):5.2f}  first convert {d(2,1):5.2f}  tile loop {d(3,2):6.2f}  stats {d(4,3):5.2f}  "
          f"| entry spread {(c[:,0].max()-c[:,0].min())/100.0:5.2f} us")
for kind, K, N, H, se in [("conv_c", 216, 96, 32, True), ("conv_c", 216, 96, 32, False), ("conv_c", 108, 48, 64, True), ("conv_c", 54, 24, 128, True),
                          ("conv_a", 96, 216, 32, False), ("conv_a", 48, 108, 64, False), ("conv_a", 24, 54, 128, False)]:
    run(kind, K, N, H, se)
