#!/usr/bin/env python
"""Where does a pointwise GEMM launch spend its time?  Builds libchange3d_hip_clk.so (pw_gemm.hip with
-DC3D_PW_CLOCK: s_memtime stamps around every phase of the wave loop, summed over waves) and prints the
per-wave average of each phase for the stage-1..3 layer shapes.

  python tools/pw_phase_clock.py --build        (CPU container: cross-compile, the .so travels with gpurun)
  python tools/pw_phase_clock.py                (GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLK_LIB = os.path.join(ROOT, "change3d_amd", "lib", "libchange3d_hip_clk.so")
PHASES = ["zero X region", "wait loads", "convert+prologue", "issue prefetch", "mfma", "stage Os", "epilogue+store",
          "loop exit", "stats flush", "params+first issue", "zero W, Pp", "barrier 1", "W load+scatter", "barrier 2",
          "fused weight gradient"]
NP = len(PHASES)


def build():
    import __graft_entry__ as g
    g.build(verbose=False)
    objdir = os.path.join(g.LIBDIR, "obj")
    clk = ["pw_gemm.hip", "pw_gemm_wg.hip", "pw_wgrad.hip", "dw_bwd_fused.hip"]
    procs = []
    for src in clk:
        o = os.path.join(objdir, src.replace(".hip", "_clk.o"))
        procs.append(subprocess.Popen([g.HIPCC] + g.FLAGS + ["-DC3D_PW_CLOCK", "-c", os.path.join(g.CSRC, src), "-o", o]))
    for p_ in procs:
        assert p_.wait() == 0
    objs = [os.path.join(objdir, s.replace(".hip", "_clk.o" if s in clk else ".o")) for s in g.SOURCES]
    subprocess.check_call([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", CLK_LIB] + objs)
    print("built", CLK_LIB)


def main():
    os.environ["C3D_LIB"] = CLK_LIB
    import torch
    from change3d_amd import _lib, ops
    h = _lib.lib()
    readers = {}
    for nm in ("c3d_debug_pw_clock", "c3d_debug_pw_wg_clock"):   # one clock array per translation unit (plain / fused variants)
        f = getattr(h, nm)
        f.restype, f.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
        readers[nm] = f
    buf = (C.c_ulonglong * (8192 * 16))()
    state = {"reader": "c3d_debug_pw_clock"}

    def read(reset=True):
        torch.cuda.synchronize()
        assert readers[state["reader"]](buf, 1 if reset else 0) == 0
        import numpy as np
        a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.float64)
        a = a[a[:, 15] > 0]
        v = list(a[:, :15].sum(0)) + [a[:, 15].sum()]
        per = a[:, :15].sum(1) / a[:, 15]
        return v, (per.min(), per.mean(), per.max()) if len(per) else (0, 0, 0)

    DEV, DT, B, T = "cuda:0", torch.bfloat16, int(os.environ.get("C3D_BENCH_B", "32")), 3
    dt = ops.dt_code(DT)
    rt = lambda *s: torch.randn(*s, device=DEV).to(DT)  # noqa: E731
    only = sys.argv[1:]
    for st, H, Cin, Ci, Co in [(1, 128, 24, 54, 24), (2, 64, 48, 108, 48), (3, 32, 96, 216, 96)]:
        M = B * T * H * H
        Cip = ops.cpad(Ci)
        x, a_, b_, c_ = rt(M, Cin), rt(M, Cip), rt(M, Cip), rt(M, Co)
        wa, wc = torch.randn(Ci, Cin, device=DEV) * 0.1, torch.randn(Co, Ci, device=DEV) * 0.1
        stats = torch.zeros(16 * 2 * 256, dtype=torch.float64, device=DEV)
        ss, gate = torch.rand(2 * Cip, device=DEV), torch.rand(B * Cip, device=DEV)
        coef3, coefo = torch.rand(3 * Cip, device=DEV), torch.rand(3 * Co, device=DEV)
        nc3 = torch.zeros(B * Cip * 3, dtype=torch.float64, device=DEV)
        rps = T * H * H
        ia = iat = ic = ict = None
        if os.environ.get("C3D_PW_IMG", "1") != "0":   # packed weight images (the product path); 0: every workgroup converts the f32 weights
            mk = lambda N, K: torch.empty(ops.pw_weight_image_bytes(N, K, dt), dtype=torch.uint8, device=DEV)  # noqa: E731
            ia, iat, ic, ict = mk(Ci, Cin), mk(Cin, Ci), mk(Co, Ci), mk(Ci, Co)
            ops.pw_pack_weights([(wa, ia, Ci, Cin, Cin, 1), (wa, iat, Cin, Ci, 1, Cin), (wc, ic, Co, Ci, Ci, 1),
                                 (wc, ict, Ci, Co, 1, Ci)], dt)
        cases = {
            "conv_a fwd   NONE+STATS": lambda: ops.pw_gemm(x, wa, a_, M=M, K=Cin, N=Ci, w_sn=Cin, w_sk=1, dtype=dt,
                                                           epi_mode=ops.EPI_STATS, stats=stats, w_img=ia),
            "conv_c fwd   SWISH+STATS": lambda: ops.pw_gemm(b_, wc, c_, M=M, K=Ci, N=Co, w_sn=Ci, w_sk=1, dtype=dt,
                                                            pro_mode=ops.PRO_BN_SE_SWISH, pro_p=ss, pro_gate=gate,
                                                            rows_per_sample=rps, epi_mode=ops.EPI_STATS, stats=stats, w_img=ic),
            "conv_c bwd-d AFFINE2+SWISH_SE_BWD": lambda: ops.pw_gemm(
                c_, wc, a_, M=M, K=Co, N=Ci, w_sn=1, w_sk=Ci, dtype=dt, x2=c_, pro_mode=ops.PRO_AFFINE2, pro_p=coefo,
                epi_mode=ops.EPI_SWISH_SE_BWD, e1=b_, epi_p=ss, epi_gate=gate, epi_q=ss, stats=nc3, rows_per_sample=rps,
                w_img=ict),
            "conv_a bwd-d AFFINE2+ADD": lambda: ops.pw_gemm(a_, wa, x, M=M, K=Ci, N=Cin, w_sn=1, w_sk=Cin, dtype=dt, x2=b_,
                                                            pro_mode=ops.PRO_AFFINE2, pro_p=coef3, epi_mode=ops.EPI_ADD,
                                                            e1=c_, res_mode=0, w_img=iat),
        }
        if Ci <= 112:   # the fused data + weight gradient variants (res2 / res3; csrc/pw_gemm_wg.hip)
            dwa_, dwc_ = torch.zeros(Ci, Cin, device=DEV), torch.zeros(Co, Ci, device=DEV)
            x3 = rt(M, Cin)
            cases["conv_c bwd-d AFFINE2+SWISH_SE_BWD +dW"] = lambda: ops.pw_gemm(
                c_, wc, a_, M=M, K=Co, N=Ci, w_sn=1, w_sk=Ci, dtype=dt, x2=c_, pro_mode=ops.PRO_AFFINE2, pro_p=coefo,
                epi_mode=ops.EPI_SWISH_SE_BWD, e1=b_, epi_p=ss, epi_gate=gate, epi_q=ss, stats=nc3, rows_per_sample=rps,
                w_img=ict, wg_mode=ops.WG_SWISH, wg_dw=dwc_)
            cases["conv_a bwd-d AFFINE2+ADD +dW"] = lambda: ops.pw_gemm(
                a_, wa, x, M=M, K=Ci, N=Cin, w_sn=1, w_sk=Cin, dtype=dt, x2=b_, pro_mode=ops.PRO_AFFINE2, pro_p=coef3,
                epi_mode=ops.EPI_ADD, e1=c_, res_mode=0, w_img=iat, wg_mode=ops.WG_ROWS, wg_dw=dwa_, wg_x3=x3)
        for name, fn in cases.items():
            if only and not any(o in f"s{st} {name}" for o in only):
                continue
            state["reader"] = "c3d_debug_pw_wg_clock" if name.endswith("+dW") else "c3d_debug_pw_clock"
            for _ in range(3):
                fn()
            read()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 10
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            v, (pmin, pmean, pmax) = read()
            us = e0.elapsed_time(e1) / iters * 1e3
            waves = v[15] / iters
            tot = sum(v[:15]) / v[15]
            print(f"s{st} {name}: {us:.1f} us/launch, {waves:.0f} waves, clk per wave mean {tot:.0f} min {pmin:.0f} "
                  f"max {pmax:.0f} ({pmax / us:.0f} clk/us if the longest wave spans the launch)")
            for i, ph in enumerate(PHASES):
                print(f"    {ph:18s} {v[i] / v[15]:10.0f} clk  {100.0 * v[i] / sum(v[:15]):5.1f} %")


WG_PHASES = ["setup", "wait loads", "convert -> LDS", "issue next tile", "barrier", "mfma", "partials store"]


def main_wgrad():
    os.environ["C3D_LIB"] = CLK_LIB
    import numpy as np
    import torch
    from change3d_amd import _lib, ops
    h = _lib.lib()
    h.c3d_debug_wgrad_clock.restype = C.c_int
    h.c3d_debug_wgrad_clock.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * (8192 * 8))()

    def read():
        torch.cuda.synchronize()
        assert h.c3d_debug_wgrad_clock(buf, 1) == 0
        return np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)

    DEV, DT, B, T = "cuda:0", torch.bfloat16, 32, 3
    dt = ops.dt_code(DT)
    rt = lambda *s: torch.randn(*s, device=DEV).to(DT)  # noqa: E731
    for st, H, Cin, Ci, Co in [(1, 128, 24, 54, 24), (2, 64, 48, 108, 48), (3, 32, 96, 216, 96)]:
        M = B * T * H * H
        Cip = ops.cpad(Ci)
        x, a_, b_, c_ = rt(M, Cin), rt(M, Cip), rt(M, Cip), rt(M, Co)
        ss, gate = torch.rand(2 * Cip, device=DEV), torch.rand(B * Cip, device=DEV)
        coef3, coefo = torch.rand(3 * Cip, device=DEV), torch.rand(3 * Co, device=DEV)
        dwc, dwa = torch.zeros(Co, Ci, device=DEV), torch.zeros(Ci, Cin, device=DEV)
        cases = {
            f"conv_c wgrad N={Co} K={Ci} (P affine2, Q swish)": lambda: ops.pw_wgrad(
                c_, b_, dwc, M=M, K=Ci, N=Co, dw_sn=Ci, dw_sk=1, dtype=dt, p2=c_, p_coef=coefo,
                q_mode=ops.PRO_BN_SE_SWISH, q_ss=ss, q_gate=gate, rows_per_sample=T * H * H),
            f"conv_a wgrad N={Ci} K={Cin} (P affine2)": lambda: ops.pw_wgrad(
                a_, x, dwa, M=M, K=Cin, N=Ci, dw_sn=Cin, dw_sk=1, dtype=dt, p2=b_, p_coef=coef3),
        }
        for name, fn in cases.items():
            for _ in range(3):
                fn()
            read()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 10
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            a = read()
            us = e0.elapsed_time(e1) / iters * 1e3
            a = a[a[:, 7] > 0]
            print(f"s{st} {name}: {us:.1f} us/launch (incl. reduce kernel), {len(a)} wave slots")
            for grp, sel in (("waves 0-3 (stage P)", np.arange(len(a)) % 8 < 4), ("waves 4-7 (stage Q)", np.arange(len(a)) % 8 >= 4)):
                v = a[sel]
                tot = v[:, :7].sum() / v[:, 7].sum()
                print(f"  {grp}: {tot:.0f} clk per wave")
                for i, ph in enumerate(WG_PHASES):
                    print(f"    {ph:18s} {v[:, i].sum() / v[:, 7].sum():10.0f} clk  {100.0 * v[:, i].sum() / v[:, :7].sum():5.1f} %")


DW_PHASES = ["setup (weights, coefficients)", "barrier A (previous taps done)", "wait prefetched rows", "convert -> f32 LDS planes",
             "issue a rows + next tile", "barrier B", "27 taps", "epilogue + store", "final sums flush"]


FB_PHASES = ["setup (weights, coefficients, first loads)", "convert -> f32 LDS planes | ring: a_in", "a_in + issue next tile / class | ring: DMA issue (tile + 2)", "barrier",
             "27 taps (data + weight gradient)", "epilogue + store", "ring: wait for tile + 1's DMA", "BN_a sums + dW flush", "ring: convert in place (tile + 1)"]


def main_fb():
    """csrc/dw_bwd_fused.hip (stride 1 and 2) at the three BCD stage shapes."""
    os.environ["C3D_LIB"] = CLK_LIB
    import numpy as np
    import torch
    from change3d_amd import _lib, ops
    h = _lib.lib()
    h.c3d_debug_fb_clock.restype = C.c_int
    h.c3d_debug_fb_clock.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * (16384 * 10))()

    def read():
        torch.cuda.synchronize()
        assert h.c3d_debug_fb_clock(buf, 1) == 0
        return np.frombuffer(buf, dtype=np.uint64).reshape(16384, 10).astype(np.float64)

    DEV, DT, B, T = "cuda:0", torch.bfloat16, 32, 3
    dt = ops.dt_code(DT)
    rt = lambda *s: torch.randn(*s, device=DEV).to(DT)  # noqa: E731
    ring = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--ring=")]
    if ring:
        ops.set_option(ops.OPT_DW_RING, ring[0])
        print(f"C3D_OPT_DW_RING = {ring[0]}")
    for st, H, Ci, stride in [(1, 128, 54, 1), (2, 64, 108, 1), (3, 32, 216, 1), (1, 256, 54, 2)]:
        Cip = ops.cpad(Ci)
        Ho = H // stride
        a_, t2 = rt(B, T, H, H, Cip), rt(B, T, H, H, Cip)
        b_, t1 = rt(B, T, Ho, Ho, Cip), rt(B, T, Ho, Ho, Cip)
        w = torch.randn(Ci, 27, device=DEV) * 0.1
        ss = torch.rand(2 * Cip, device=DEV)
        dsums = torch.zeros(2 * Ci, dtype=torch.float64, device=DEV)
        dw = torch.zeros(Ci, 27, device=DEV)
        cA, cC, cB = torch.rand(Cip, device=DEV), torch.rand(Cip, device=DEV), torch.rand(B * Cip, device=DEV)
        fn = lambda: ops.dw_bwd_fused(t1, b_, cA, cB, cC, w, a_, ss, ss, t2, dsums, dw, B, T, H, H, Ci, dt, stride)  # noqa: E731
        for _ in range(3):
            fn()
        read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        a = read()
        us = e0.elapsed_time(e1) / iters * 1e3
        a = a[a[:, 9] > 0]
        per = a[:, :9].sum(1) / a[:, 9]
        print(f"s{st} dw bwd fused C={Ci} {H}x{H} stride {stride}: {us:.1f} us/launch, {a[:, 9].sum() / iters:.0f} waves/launch, "
              f"clk per wave mean {per.mean():.0f} min {per.min():.0f} max {per.max():.0f}")
        for i, ph in enumerate(FB_PHASES):
            print(f"    {ph:44s} {a[:, i].sum() / a[:, 9].sum():10.0f} clk  {100.0 * a[:, i].sum() / a[:, :9].sum():5.1f} %")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--fb" in sys.argv:
        main_fb()
    elif "--wgrad" in sys.argv:
        main_wgrad()
    else:
        main()
