"""Where does the C3D_WG_MASKSUM output differ from (c3d_pw_gemm EPI_ADD, c3d_block_out_bwd)?  (debug helper)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from test_ops_gpu import DEV, padc, q, rnd
from change3d_amd import ops
DT = torch.bfloat16
for (M, Ci, Cin) in [(777, 120, 80), (768, 120, 80), (777, 216, 80), (777, 120, 96), (777, 96, 96), (777, 112, 112), (1000, 216, 96), (50, 216, 112)]:
    dt = ops.dt_code(DT)
    Cip, Cinp = ops.cpad(Ci), ops.cpad(Cin)
    t2, a_ = q(rnd((M, Ci), 41), DT), q(rnd((M, Ci), 42), DT)
    A, Bc, Cc = rnd((Ci,), 43), rnd((Ci,), 44, 0.1), rnd((Ci,), 45, 0.1)
    w = rnd((Ci, Cin), 46, 0.2)
    y_prev = torch.relu(q(rnd((M, Cin), 47), DT)); res = q(rnd((M, Cin), 48), DT); cten = q(rnd((M, Cin), 49), DT)
    t2d, ad = (padc(t, Cip).to(DEV, DT).contiguous() for t in (t2, a_))
    yd, rd, cd_ = (padc(t, Cinp).to(DEV, DT).contiguous() for t in (y_prev, res, cten))
    coef = torch.cat([padc(A, Cip), padc(Bc, Cip), padc(Cc, Cip)]).to(DEV)
    mr = torch.cat([padc(rnd((Cin,), 50, 0.5), Cinp), padc(rnd((Cin,), 51).abs() + 0.5, Cinp)]).to(DEV)
    wd = w.to(DEV)
    dx = torch.full((M, Cinp), float("nan"), dtype=DT, device=DEV)
    ops.pw_gemm(t2d, wd, dx, M=M, K=Ci, N=Cin, w_sn=1, w_sk=Cin, dtype=dt, x2=ad, pro_mode=ops.PRO_AFFINE2, pro_p=coef, epi_mode=ops.EPI_ADD, e1=rd)
    g_ref = torch.where(yd > 0, dx, torch.zeros_like(dx))
    g = torch.full((M, Cinp), float("nan"), dtype=DT, device=DEV)
    s = torch.zeros(2 * Cin, dtype=torch.float64, device=DEV)
    try:
        ops.pw_gemm(t2d, wd, g, M=M, K=Ci, N=Cin, w_sn=1, w_sk=Cin, dtype=dt, x2=ad, pro_mode=ops.PRO_AFFINE2, pro_p=coef,
                    epi_mode=ops.EPI_ADD, e1=rd, wg_mode=ops.WG_MASKSUM, wg_x3=yd, add_c=cd_, add_mr=mr, add_sums=s)
    except Exception as e:
        print((M, Ci, Cin), "refused", e); continue
    torch.cuda.synchronize()
    bad = (g.view(torch.int16) != g_ref.view(torch.int16))
    gq = g_ref[:, :Cin].double().cpu()
    chat = (cten.double() - mr[:Cin].double().cpu()) * mr[Cinp:Cinp + Cin].double().cpu()
    want = torch.cat([gq.sum(0), (gq * chat).sum(0)])
    print((M, Ci, Cin), "mismatches", int(bad.sum()), "rows", sorted(set((bad.nonzero()[:, 0] // 16).tolist()))[:20],
          "cols", sorted(set(bad.nonzero()[:, 1].tolist()))[:20], "nan", int(torch.isnan(g.float()).sum()),
          "sum err", float((s.cpu() - want).abs().max()), "scale", float(want.abs().max()))
    if bad.any():
        i, j = bad.nonzero()[0].tolist()
        print("   first", i, j, float(g[i, j]), float(g_ref[i, j]), float(dx[i, j]), float(yd[i, j]))
