#!/usr/bin/env python
"""Where does the cooperative conv_c data gradient differ from the first kernel's?  (debug aid for tests/test_pw_wg_gpu.py)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from test_ops_gpu import DEV, padc, q, rnd
from change3d_amd import ops
DT = torch.bfloat16
Co, Ci, rows, B, ragged, gated = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (96, 216, 64, 37, 0, 1)
M = B * rows - ragged
dt = ops.dt_code(DT)
Cop, Cip = ops.cpad(Co), ops.cpad(Ci)
g, c = q(rnd((M, Co), 81), DT), q(rnd((M, Co), 82), DT)
A, Bc, Cc = rnd((Co,), 83), rnd((Co,), 84, 0.1), rnd((Co,), 85, 0.1)
w = rnd((Co, Ci), 86, 0.2)
b = q(rnd((M, Ci), 87), DT)
scale, shift = rnd((Ci,), 88).abs() + 0.5, rnd((Ci,), 89, 0.3)
gate = torch.sigmoid(rnd((B, Ci), 90))
mean, rstd = rnd((Ci,), 91, 0.5), rnd((Ci,), 92).abs() + 0.5
gd, cd = (padc(t, Cop).to(DEV, DT).contiguous() for t in (g, c))
bd = padc(b, Cip).to(DEV, DT).contiguous()
coef = torch.cat([padc(A, Cop), padc(Bc, Cop), padc(Cc, Cop)]).to(DEV)
ss = torch.cat([padc(scale, Cip), padc(shift, Cip)]).to(DEV)
mr = torch.cat([padc(mean, Cip), padc(rstd, Cip)]).to(DEV)
gt = padc(gate, Cip).to(DEV).contiguous() if gated else None
wd = w.to(DEV)
img = torch.zeros(ops.pw_weight_image_bytes(Ci, Co, dt), dtype=torch.uint8, device=DEV)
ops.pw_pack_weights([(wd, img, Ci, Co, 1, Ci)], dt)
base = dict(M=M, K=Co, N=Ci, w_sn=1, w_sk=Ci, dtype=dt, x2=cd, pro_mode=ops.PRO_AFFINE2, pro_p=coef, epi_mode=ops.EPI_SWISH_SE_BWD,
            e1=bd, epi_p=ss, epi_gate=gt, epi_q=mr, rows_per_sample=rows, w_img=img)
out = {}
for key, opt in ((0, 0), (3, 3), ("0b", 0), ("3b", 3)):
    ops.set_option(ops.OPT_PW_CDG, opt)
    t1 = torch.full((M, Cip), float("nan"), dtype=DT, device=DEV)
    nc3 = torch.zeros(B * Cip * 3, dtype=torch.float64, device=DEV)
    dw = torch.ones((Co, Ci), dtype=torch.float32, device=DEV)
    if opt == 0 and Co > 48:
        ops.pw_gemm(gd, wd, t1, stats=nc3, **base)
    else:
        ops.pw_gemm(gd, wd, t1, stats=nc3, wg_mode=ops.WG_SWISH, wg_dw=dw, **base)
    torch.cuda.synchronize()
    out[key] = (t1.cpu(), nc3.cpu())
print("first kernel run-to-run differing:", int((out[0][0].view(torch.int16) != out["0b"][0].view(torch.int16)).sum()), " cooperative run-to-run:", int((out[3][0].view(torch.int16) != out["3b"][0].view(torch.int16)).sum()))
d = out[0][0].view(torch.int16) != out[3][0].view(torch.int16)
print("differing elements", int(d.sum()), "of", d.numel())
rowsd = d.any(1).nonzero().flatten()
print("rows", rowsd[:40].tolist(), "... count", len(rowsd))
colsd = d.any(0).nonzero().flatten()
print("cols", colsd[:60].tolist(), "... count", len(colsd))
if len(rowsd):
    r = int(rowsd[0]); cs = d[r].nonzero().flatten()[:8]
    print("row", r, "cols", cs.tolist(), "first", out[0][0][r, cs].tolist(), "coop", out[3][0][r, cs].tolist())
n0, n3 = out[0][1].view(B, Cip, 3), out[3][1].view(B, Cip, 3)
print("sums max abs diff", (n0 - n3).abs().max().item(), "scale", n0.abs().max().item())
