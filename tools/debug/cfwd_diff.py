#!/usr/bin/env python
"""conv_c forward: csrc/pw_cfwd.hip against the wave-private-tile kernel on the same device buffers (C3D_OPT_PW_CFWD),
per shape: are the outputs bit-identical, where do they differ, how far apart are the statistics.
usage: cfwd_diff.py [se]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from change3d_amd import ops, _lib as L
DEV = "cuda:0"
use_se = len(sys.argv) > 1 and sys.argv[1] == "se"
p = lambda t: t.data_ptr() if t is not None else None
for (K, N, B, rps) in ((216, 96, 8, 768), (108, 48, 4, 3072), (54, 24, 4, 3072), (216, 96, 32, 3072)):
    Kp, Np = ops.cpad(K), ops.cpad(N)
    M = B * rps
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(M, Kp); x[:, :K] = torch.randn(M, K, generator=g)
    x = x.to(DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    xs = x.float().view(B, rps, Kp)
    nc = torch.zeros(B, Kp, 2, dtype=torch.float64, device=DEV)
    nc[:, :, 0] = xs.double().sum(1); nc[:, :, 1] = (xs.double() ** 2).sum(1)
    gamma = (torch.rand(K, generator=g) + 0.5).to(DEV); beta = (torch.randn(K, generator=g) * 0.2).to(DEV)
    Cr = 16
    w1 = (torch.randn(Cr, K, generator=g) * 0.1).to(DEV); b1 = torch.zeros(Cr, device=DEV)
    w2 = (torch.randn(K, Cr, generator=g) * 0.1).to(DEV); b2 = torch.zeros(K, device=DEV)
    img = torch.empty(ops.pw_weight_image_bytes(N, K, ops.DT_BF16), dtype=torch.uint8, device=DEV)
    ops.pw_pack_weights([(w, img, N, K, K, 1)], ops.DT_BF16)
    res = {}
    for opt in (0, 1):
        ops.set_option(ops.OPT_PW_CFWD, opt)
        y = torch.full((M, Np), 7.0, device=DEV).to(torch.bfloat16)
        stats = torch.zeros(ops.STAT_STRIPES * 2 * N, dtype=torch.float64, device=DEV)
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
        ss, mr = torch.zeros(2 * Kp, device=DEV), torch.zeros(2 * Kp, device=DEV)
        gate, hid = torch.zeros(B, Kp, device=DEV), torch.zeros(B, Cr, device=DEV)
        a = L.PwArgs()
        a.x, a.y, a.w, a.w_img = p(x), p(y), p(w), p(img)
        a.M, a.K, a.Kp, a.N, a.Np, a.w_sn, a.w_sk = M, K, Kp, N, Np, K, 1
        a.rows_per_sample, a.dtype = rps, ops.DT_BF16
        a.pro_mode, a.epi_mode = ops.PRO_BN_SE_SWISH, ops.EPI_STATS
        a.pro_p, a.pro_gate, a.stats = p(ss), (p(gate) if use_se else None), p(stats)
        f = L.BnFin()
        f.gamma, f.beta, f.running_mean, f.running_var, f.nbt, f.ss, f.mr = p(gamma), p(beta), p(rm), p(rv), p(nbt), p(ss), p(mr)
        f.count, f.momentum, f.eps, f.training, f.batch, f.sums = float(M), 0.1, 1e-5, 1, B, p(nc)
        a.fin = f
        if use_se:
            a.se_w1, a.se_b1, a.se_w2, a.se_b2, a.se_hid, a.se_cr = p(w1), p(b1), p(w2), p(b2), p(hid), Cr
        rc = L.lib().c3d_pw_gemm(C.byref(a), ops._stream())
        torch.cuda.synchronize()
        assert rc == 0, rc
        res[opt] = (y.float().cpu(), stats.cpu().view(ops.STAT_STRIPES, 2, N).sum(0), ss.cpu(), gate.cpu(), rm.cpu(), int(nbt))
    y0, y1 = res[0][0], res[1][0]
    d = (y0 != y1)
    print(f"K={K} N={N} B={B} rps={rps} se={use_se}: y differs in {int(d.sum())} of {d.numel()} elements; "
          f"rows affected {int(d.any(1).sum())}, columns {d.any(0).nonzero().flatten().tolist()[:12]}; first rows {d.any(1).nonzero().flatten().tolist()[:8]}")
    print("   stats rel diff", ((res[0][1] - res[1][1]).abs() / res[0][1].abs().clamp_min(1e-9)).max().item(),
          " ss equal", torch.equal(res[0][2], res[1][2]), " gate equal", torch.equal(res[0][3], res[1][3]),
          " running_mean equal", torch.equal(res[0][4], res[1][4]), " nbt", res[0][5], res[1][5])
ops.set_option(ops.OPT_PW_CFWD, 1)
