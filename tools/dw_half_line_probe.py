import sys, os, torch
sys.path.insert(0, os.getcwd())
from change3d_amd import ops
DEV, DT = "cuda:0", torch.bfloat16
dt = ops.dt_code(DT)
B, T, H = 32, 3, 64
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for Ci in (32, 64, 96, 128, 24, 56, 112):
    Cip = ops.cpad(Ci)
    rt = lambda *s: torch.randn(*s, device=DEV).to(DT)
    a_, b_, t1, t2 = (rt(B, T, H, H, Cip) for _ in range(4))
    w = torch.randn(Ci, 27, device=DEV) * 0.1
    ss = torch.rand(2 * Cip, device=DEV)
    nc = torch.zeros(B * Cip * 2, dtype=torch.float64, device=DEV)
    dsums = torch.zeros(2 * Ci, dtype=torch.float64, device=DEV)
    cA, cC, cB = torch.rand(Cip, device=DEV), torch.rand(Cip, device=DEV), torch.rand(B * Cip, device=DEV)
    n = a_.numel() * 2
    dw = torch.zeros(Ci, 27, device=DEV)
    us = timeit(lambda: ops.dw_bwd_fused(t1, b_, cA, cB, cC, w, a_, ss, ss, t2, dsums, dw, B, T, H, H, Ci, dt, 1))
    usf = timeit(lambda: ops.dw_fwd(a_, ss, w, b_, nc, B, T, H, H, Ci, 1, dt))
    print(f"C={Ci:4d} Cp={Cip:4d}: bwd-data {us:8.1f} us {4*n/us/1e3:8.1f} GB/s ({us/Cip:6.3f} us/ch)   fwd {usf:8.1f} us {2*n/usf/1e3:8.1f} GB/s ({usf/Cip:6.3f} us/ch)")
