import sys, torch
sys.path.insert(0, '.')
from change3d_amd import ops
DEV='cuda:0'; B,T,H,W=32,3,256,256
dt=ops.dt_code(torch.bfloat16)
x=torch.randn(B,3,T,H,W,device=DEV); w_t=torch.randn(24,3,1,3,3,device=DEV)*0.3
dv=torch.randn(B,T,H,W,24,device=DEV).to(torch.bfloat16)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
dw=torch.zeros(24,27,device=DEV); dPs=torch.zeros(3,1,H,W,device=DEV); dPf=torch.zeros(B,3,T,H,W,device=DEV)
print('summed dP (product mode)', timeit(lambda: ops.stem_bwd_wx(x,w_t,dv,dw,dPs,B,T,H,W,1,1,False,dt)))
print('no dP                   ', timeit(lambda: ops.stem_bwd_wx(x,w_t,dv,dw,None,B,T,H,W,1,1,False,dt)))
print('per-sample dP           ', timeit(lambda: ops.stem_bwd_wx(x,w_t,dv,dw,dPf,B,T,H,W,0,T,True,dt)))
