#!/usr/bin/env python
"""Phase clocks of the cooperative conv_c data + weight gradient (csrc/pw_cdgrad.hip built with -DC3D_CD_CLOCK:
tools/r6/mkvariant.sh cdclk pw_cdgrad.hip "-DC3D_CD_CLOCK -Wno-dangling-else"; run with C3D_LIB=.../libchange3d_hip_cdclk.so)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from change3d_amd import ops, _lib as L
DEV, DT = "cuda:0", torch.bfloat16
h = L.lib()
rd = h.c3d_debug_cd_clock; rd.restype = C.c_int; rd.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (1024 * 8))()
B, T = 32, 3
dt = ops.dt_code(DT)
for Co, Ci, H in [(96, 216, 32), (48, 108, 64), (24, 54, 128)]:
    rows = T * H * H; M = B * rows
    Cop, Cip = ops.cpad(Co), ops.cpad(Ci)
    rt = lambda *s: torch.randn(*s, device=DEV).to(DT)
    g, c, b = rt(M, Cop), rt(M, Cop), rt(M, Cip)
    w = torch.randn(Co, Ci, device=DEV) * 0.1
    coef, ss, mr = torch.rand(3 * Cop, device=DEV), torch.rand(2 * Cip, device=DEV), torch.rand(2 * Cip, device=DEV)
    gate = torch.rand(B, Cip, device=DEV)
    img = torch.zeros(ops.pw_weight_image_bytes(Ci, Co, dt), dtype=torch.uint8, device=DEV)
    ops.pw_pack_weights([(w, img, Ci, Co, 1, Ci)], dt)
    t1 = torch.empty(M, Cip, dtype=DT, device=DEV)
    nc3 = torch.zeros(B * Cip * 3, dtype=torch.float64, device=DEV)
    dw = torch.zeros(Co, Ci, device=DEV)
    fn = lambda: ops.pw_gemm(g, w, t1, M=M, K=Co, N=Ci, w_sn=1, w_sk=Ci, dtype=dt, x2=c, pro_mode=ops.PRO_AFFINE2, pro_p=coef,
                             epi_mode=ops.EPI_SWISH_SE_BWD, e1=b, epi_p=ss, epi_gate=gate, epi_q=mr, stats=nc3, rows_per_sample=rows,
                             w_img=img, wg_mode=ops.WG_SWISH, wg_dw=dw)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    assert rd(buf) == 0
    import numpy as np
    k = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    k = k[k[:, 5] > k[:, 0]]
    d = lambda i, j: (k[:, i] - k[:, j]).mean() / 2100.0   # us at ~2.1 GHz
    print(f"conv_c dgrad+wgrad {Co}->{Ci} H={H}: {us:6.1f} us/launch (kernel + reducer) | wgs {len(k)} | prologue {d(1,0):5.1f}  first convert {d(2,1):4.1f}  "
          f"tile loop {d(3,2):6.1f}  last wgrad + partials {d(4,3):4.1f}  sums flush {d(5,4):4.1f} us")
