import sys, os, torch
sys.path.insert(0, os.getcwd())
from change3d_amd import ops
DEV, DT = "cuda:0", torch.bfloat16
dt = ops.dt_code(DT)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, NC, sig in [(32, 1, 1), (16, 7, 0)]:
    H = W = 256
    x = torch.randn(B, H, W, 24, device=DEV).to(DT)
    w = torch.randn(NC, 24, 3, 3, device=DEV) * 0.1
    dout = torch.randn(B, NC, H, W, device=DEV)
    prob = torch.rand(B, NC, H, W, device=DEV)
    dx = torch.empty_like(x)
    dw = torch.zeros(NC, 24, 3, 3, device=DEV)
    us = timeit(lambda: ops.head_bwd(dout, prob, x, w, dx, dw, B, H, W, 24, NC, sig, dt))
    dw.zero_(); ops.head_bwd(dout, prob, x, w, dx, dw, B, H, W, 24, NC, sig, dt); torch.cuda.synchronize()
    print(f"B={B} NC={NC}: {us:.1f} us  dw checksum {dw.double().sum().item():.6e} |dw| {dw.double().abs().sum().item():.6e} dx checksum {dx.double().abs().sum().item():.6e}")
