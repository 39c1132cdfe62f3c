#!/usr/bin/env python
"""Where does c3d_se_bn_bwd_coef spend its ~28 us?  Builds libchange3d_hip_seclk.so (bn_se.hip with -DC3D_SE_CLOCK: s_memtime
stamps of workgroup 0 / thread 0 at the phase boundaries) and prints the clocks per phase for the res2 / res3 / res4 / res5 shapes.
  python tools/se_phase_clock.py --build   (CPU container)      python tools/se_phase_clock.py   (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLK_LIB = os.path.join(ROOT, "change3d_amd", "lib", "libchange3d_hip_seclk.so")
PH = ["loads: w2, hid, du / zz", "barrier", "hidden gradient dh", "stage w1 (+2 barriers)", "dz", "FC parameter gradients",
      "barrier", "BN coefficients"]


def build():
    import __graft_entry__ as g
    g.build(verbose=False)
    objdir = os.path.join(g.LIBDIR, "obj")
    o = os.path.join(objdir, "bn_se_seclk.o")
    subprocess.check_call([g.HIPCC] + g.FLAGS + ["-DC3D_SE_CLOCK", "-c", os.path.join(g.CSRC, "bn_se.hip"), "-o", o])
    objs = [o if s == "bn_se.hip" else os.path.join(objdir, s.replace(".hip", ".o")) for s in g.SOURCES]
    subprocess.check_call([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", CLK_LIB] + objs)
    print("built", CLK_LIB)


def main():
    os.environ["C3D_LIB"] = CLK_LIB
    import torch
    from change3d_amd import _lib
    h = _lib.lib()
    rd = h.c3d_debug_se_clock
    rd.restype, rd.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
    dev = "cuda:0"
    f = h.c3d_se_bn_bwd_coef
    for name, B, C_, Cr, rps in [("res2", 32, 54, 8, 3 * 128 * 128), ("res3", 32, 108, 8, 3 * 64 * 64), ("res4", 32, 216, 16, 3 * 32 * 32),
                                 ("res5 (CC, B=16)", 16, 432, 32, 3 * 16 * 16)]:
        Cp = (C_ + 7) // 8 * 8
        g = torch.Generator().manual_seed(1)
        r = lambda *s, dt=torch.float32: torch.randn(*s, generator=g).to(dev).to(dt)  # noqa: E731
        nc3, ncf = r(B, Cp, 3, dt=torch.float64), r(B, Cp, 2, dt=torch.float64).abs()
        gamma, mr, ss = r(C_), r(2 * Cp).abs() + 0.5, r(2 * Cp)
        w1, w2, gate, hid = r(Cr, C_), r(C_, Cr), torch.sigmoid(r(B, Cp)), r(B, Cr)
        cA, cC, cB = r(Cp), r(Cp), r(B, Cp)
        grads = [torch.zeros(n, device=dev) for n in (C_, C_, Cr * C_, Cr, C_ * Cr, C_)]
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def launch():
            rc = f(p(nc3), p(ncf), B, float(rps), p(gamma), p(mr), p(ss), C_, Cp, p(w1), p(w2), p(gate), p(hid), Cr, p(cA), p(cC), p(cB),
                   p(grads[0]), p(grads[1]), p(grads[2]), p(grads[3]), p(grads[4]), p(grads[5]), st)
            assert rc == 0, rc
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        rd(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        torch.cuda.synchronize()
        rd(buf, 0)
        tot = sum(buf[i] for i in range(8))
        print(f"{name}: B={B} C={C_} Cr={Cr}: {1e3 * e0.elapsed_time(e1) / n:.1f} us per back-to-back launch; workgroup 0 thread 0: "
              f"{tot / n:.0f} s_memtime ticks (100 MHz: {tot / n / 100:.1f} us)")
        for i, nm in enumerate(PH):
            print(f"    {nm:28s} {buf[i] / n:8.0f} ticks  {100.0 * buf[i] / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
