#!/usr/bin/env python
"""Phase clocks of the Toeplitz-MFMA depthwise forward experiment (csrc/dw_toeplitz.hip, C3D_DW_TZ_CLK=1)."""
import ctypes as C, os, sys
os.environ["C3D_DW_TZ"], os.environ["C3D_DW_TZ_CLK"] = "1", "1"
# the experiment is linked into the instrumented build only (python __graft_entry__.py --tuning)
os.environ.setdefault("C3D_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "change3d_amd", "lib", "libchange3d_hip_tune.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from change3d_amd import _lib, ops
PH = ["issue loads", "wait loads", "convert + LDS write", "barrier 1", "MFMA + results", "statistics", "barrier 2", "gather LDS reads",
      "perm + stores", "loop head"]
h = _lib.lib()
h.c3d_debug_tz_clock.restype = C.c_int
buf = (C.c_ulonglong * 11)()
dt = torch.bfloat16
for (B, H, Cc) in ((32, 128, 54), (32, 64, 108), (32, 32, 216)):
    Cp = ops.cpad(Cc)
    a = torch.randn(B, 3, H, H, Cp, device="cuda").to(dt)
    ss = torch.cat([torch.ones(Cp), torch.zeros(Cp)]).cuda()
    w = torch.randn(Cc, 27, device="cuda") * 0.2
    y = torch.empty_like(a)
    nc = torch.zeros(B * Cp * 2, dtype=torch.float64, device="cuda")
    for nch in ("16", "8"):
        os.environ["C3D_DW_TZ_NCH"] = nch
        ops.dw_fwd(a, ss, w, y, nc, B, 3, H, H, Cc, 1, ops.dt_code(dt)); h.c3d_debug_tz_clock(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.dw_fwd(a, ss, w, y, nc, B, 3, H, H, Cc, 1, ops.dt_code(dt)); e1.record(); torch.cuda.synchronize()
        h.c3d_debug_tz_clock(buf)
        v = list(buf); waves = max(v[10], 1); tot = sum(v[:10])
        print(f"{H}x{H} C={Cc} NCH={nch}: {e0.elapsed_time(e1) * 1e3:.1f} us, {waves} waves, {tot / waves:.0f} clk per wave")
        for i, n in enumerate(PH):
            print(f"    {n:22s} {v[i] / waves:9.0f} clk {100.0 * v[i] / tot:5.1f} %")
