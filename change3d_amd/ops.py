"""Thin torch-tensor wrappers over the C ABI (include/change3d_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
computation below is a hand-written gfx950 kernel in libchange3d_hip.so.  All functions
launch on `torch.cuda.current_stream()` and never synchronise.
"""
import ctypes as C

import os

import torch

from . import _lib as L
from ._lib import (OPT_CONVT_MFMA, OPT_DW_FWD_HV, OPT_DW_RING, OPT_FOLD_SE, OPT_FUSE_WGRAD, OPT_MASK_IN_DGRAD, OPT_PW_CDG, OPT_PW_CFWD, OPT_PW_WGRAD_V2, OPT_SIDE_STREAM, OPT_STEM_MFMA, STAGE_NO_WEIGHT_IMAGES, STAGE_SEPARATE_FINALIZE,  # noqa: F401
                   STAGE_SEPARATE_RESIDUAL, STAGE_SEPARATE_WGRAD)
from ._lib import (DT_BF16, DT_F32, EPI_ADD, EPI_STATS, EPI_STORE, EPI_SWISH_SE_BWD, PRO_AFFINE2,  # noqa: F401
                   PRO_BN_SE_SWISH, PRO_NONE, ROWS_DENSE, ROWS_FRAME, ROWS_S2SHIFT, ROWS_STRIDE2, SC_BN,
                   SC_IDENTITY, SC_NONE, SC_RAW, STAT_STRIPES)


def cpad(c):
    return (c + 7) // 8 * 8


def dt_code(dtype):
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"unsupported activation dtype {dtype} (float32 | bfloat16)")


TRACE = None      # debug: list of (kernel name, [(shape, dtype, checksum), ...]) when tracing is on
_TRACE_ARGS = []


def trace_begin():
    """Debug aid (tools/determinism.py): after every launch, checksum every tensor the wrapper
    passed to it.  Two runs from identical state must give identical traces; the first differing
    entry names the kernel whose output is not reproducible."""
    global TRACE
    TRACE = []
    _TRACE_ARGS.clear()


def trace_end():
    global TRACE
    t, TRACE = TRACE, None
    _TRACE_ARGS.clear()
    return t


def _p(t):
    if t is None:
        return None
    if TRACE is not None:
        _TRACE_ARGS.append(t)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t, what="input"):
    if not t.is_cuda:
        raise L.Change3DHipError(
            f"{what} is on {t.device}: the Change3D hot path runs only as HIP kernels on an MI355X "
            f"(no CPU fallback). Use the oracle in oracle/ for CPU reference results.")


PROFILE = None  # dict: kernel name -> list of (start_event, end_event, algorithmic_bytes) when enabled


def profile_begin(serial=True, detail=False):
    """Per-kernel HIP-event profile of everything launched until `profile_end()`: the wrappers below time their own
    launches, the C++ stage driver times the launches it enqueues itself (`c3d_prof_begin`).  serial: weight
    gradients run inline on the launch stream, so an event pair brackets exactly one kernel."""
    global PROFILE, PROFILE_DETAIL, _prof_side_was, SIDE_STREAM
    PROFILE = {}
    PROFILE_DETAIL = detail
    _prof_side_was = SIDE_STREAM
    if serial:
        SIDE_STREAM = False
    L.check(L.lib().c3d_prof_begin((1 if serial else 0) | (2 if detail else 0)), "c3d_prof_begin")


def profile_end():
    """Returns {name: dict(launches, ms_total, bytes_total)} and disables profiling."""
    global PROFILE, PROFILE_DETAIL, SIDE_STREAM
    prof, PROFILE = PROFILE, None
    PROFILE_DETAIL = False
    SIDE_STREAM = _prof_side_was
    rows = (L.ProfRow * 512)()
    n = C.c_int32(0)
    L.check(L.lib().c3d_prof_end(rows, 512, C.byref(n)), "c3d_prof_end")   # synchronises the device
    out = {}
    for name, recs in (prof or {}).items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
        out[name] = dict(launches=len(recs), ms_total=ms, bytes_total=float(sum(b for _, _, b in recs)))
    for r in rows[:n.value]:
        d = out.setdefault(r.name.decode(), dict(launches=0, ms_total=0.0, bytes_total=0.0))
        d["launches"] += r.launches
        d["ms_total"] += r.ms_total
        d["bytes_total"] += r.bytes_total
    return out


_prof_side_was = True
PROFILE_DETAIL = False  # profile keys carry the GEMM shape/mode (bench.py --kernel-table)


# ---------------------------------------------------------------------------- weights version
# Bumped by everything in this package that writes parameters or BatchNorm buffers IN PLACE without going through
# nn.Module hooks (FusedAdam.launch, broadcast_module_state, ParamArena placement): X3DResStage compares it with the
# version its folded-BatchNorm weights were built from, so an eval forward never runs on stale folded weights.
_weights_version = [0]


def weights_version():
    return _weights_version[0]


def bump_weights_version():
    _weights_version[0] += 1


# ---------------------------------------------------------------------------- side stream
# Weight gradients (and the stem's input gradient) are leaves of the backward graph.  Launched on a
# side stream they fill the chip while the data-gradient chain sits in single-workgroup coefficient
# kernels and in kernel tails (eager launch only: a HIP-graph replay runs its branches one after
# another).  Everything is joined back at the end of the autograd pass (engine callback), so callers
# see ordinary stream semantics: after backward() returns, p.grad is safe to read on the current stream.
SIDE_STREAM = os.environ.get("C3D_WGRAD_SIDE", "1") != "0"
_side_streams = {}
_side_pending = []      # (seq, done_event, tensors kept alive until the event has been waited for)
_side_state = {"seq": 0}


def set_option(option, value):
    """Run-time options of the library (include/change3d_hip.h: C3D_OPT_*)."""
    L.check(L.lib().c3d_set_option(int(option), int(value)), "c3d_set_option")


def side_run(fn, *tensors):
    """Run `fn` (kernel launches) on the side stream after everything issued so far on the current
    stream.  Only inside an autograd backward pass (that is where the join callback can be queued: one
    per call, joining is idempotent); anywhere else `fn` runs inline."""
    in_backward = False
    if SIDE_STREAM:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(side_join)
            in_backward = True
        except RuntimeError:
            pass
    if not in_backward:
        fn()
        return
    dev = torch.cuda.current_device()
    st = _side_streams.get(dev)
    if st is None:
        st = _side_streams[dev] = torch.cuda.Stream(dev)
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(st):
        st.wait_event(ev)
        fn()
        done = torch.cuda.Event()
        done.record()
    _side_state["seq"] += 1
    _side_pending.append((_side_state["seq"], done, [t for t in tensors if t is not None]))


def side_mark():
    return _side_state["seq"]


def side_join(upto=None):
    """Make the current stream wait for the side work issued up to mark `upto` (default: all of it --
    including the C++ stage driver's own side stream) and release the tensors it was reading."""
    last = None
    while _side_pending and (upto is None or _side_pending[0][0] <= upto):
        last = _side_pending.pop(0)
    if last is not None:
        torch.cuda.current_stream().wait_event(last[1])
    if upto is None and _driver_keep:
        rc = L.lib().c3d_side_join(_stream())
        if rc != 0:
            raise L.Change3DHipError(f"c3d_side_join failed with code {rc}")
        _driver_keep.clear()


_driver_keep = []   # workspaces the stage driver's side stream may still be reading (freed at the next full join)


def keep_until_join(*tensors):
    """Keep `tensors` alive until the next full `side_join()`; inside an autograd backward pass that join is
    queued as an end-of-pass callback, so after backward() returns p.grad is safe to read."""
    _driver_keep.extend(t for t in tensors if t is not None)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(side_join)
    except RuntimeError:   # not inside backward(): join now
        side_join()


def _detail(name, a):
    if PROFILE is None or not PROFILE_DETAIL:
        return name
    if name == "c3d_pw_gemm":
        return f"{name}[M={a.M} K={a.K} N={a.N} pro={a.pro_mode} epi={a.epi_mode} rows={a.row_mode}]"
    return f"{name}[M={a.M} K={a.K} N={a.N} q={a.q_mode} rows={a.row_mode}]"


def _launch(name, nbytes, fn, *args):
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        prof.setdefault(name, []).append((e0, e1, nbytes))
    else:
        rc = fn(*args)
    if rc != 0:
        raise L.Change3DHipError(f"{name} failed with code {rc}")
    if TRACE is not None:
        TRACE.append((name, [(tuple(t.shape), str(t.dtype), float(t.detach().double().abs().sum().item()))
                             for t in _TRACE_ARGS]))
        _TRACE_ARGS.clear()


def _es(dtype):
    return 4 if dtype == DT_F32 else 2


def grad_of(p):
    """Accumulation target for a parameter gradient (kernels do `+=`)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# ----------------------------------------------------------------------------- pointwise GEMM
def fin_consume(sums, bn, count, ss, mr, batch=0):
    """c3d_bn_fin for the consumer-side finalisation (c3d_dw333_fwd_fin / c3d_block_out_fwd_fin): `sums` are the
    completed f64 [16][2][C] statistics of an earlier launch; batch > 0: the per-sample layout [batch][Cp][2] of
    c3d_dw333_fwd (BatchNorm_b of a block without SqueezeExcitation, consumed by c3d_pw_gemm's BN_SE_SWISH prologue)."""
    f = L.BnFin()
    f.batch = int(batch)
    f.sums = _p(sums)
    f.gamma, f.beta = _p(bn.weight), _p(bn.bias)
    f.running_mean, f.running_var, f.nbt = _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked)
    f.ss, f.mr, f.count = _p(ss), _p(mr), float(count)
    f.momentum, f.eps, f.training = float(bn.momentum), float(bn.eps), 1
    return f


WG_NONE, WG_SWISH, WG_ROWS, WG_MASKSUM = 0, 1, 2, 3


def pw_gemm(x, w, y, *, M, K, N, w_sn, w_sk, dtype, x2=None, pro_mode=PRO_NONE, pro_p=None,
            pro_gate=None, epi_mode=EPI_STORE, e1=None, epi_p=None, epi_gate=None, epi_q=None, stats=None,
            rows_per_sample=0, row_mode=ROWS_DENSE, rpg=0, gstride=0, H=0, W=0, res_mode=0,
            x_ptr=None, e1_ptr=None, fin=None, bias=None, pro_out=None, w_img=None, wg_mode=WG_NONE, wg_x3=None, wg_dw=None, wg_mask_out=0,
            add_c=None, add_mr=None, add_sums=None):
    a = L.PwArgs()
    a.w_img = _p(w_img)
    if wg_mode == WG_MASKSUM:   # no weight gradient: the previous block's c3d_block_out_bwd in this epilogue (c3d_pw_args.add_sums)
        a.wg_mode, a.wg_x3 = wg_mode, _p(wg_x3)
    elif wg_mode != WG_NONE:   # weight gradient fused into this data-gradient launch (include/change3d_hip.h c3d_pw_args.wg_mode)
        a.wg_mode, a.wg_x3, a.wg_dw, a.wg_mask_out = wg_mode, _p(wg_x3), _p(wg_dw), int(wg_mask_out)
        a.wg_ws = _ws(wg_dw.device, L.lib().c3d_pw_gemm_wg_ws_floats(K, N)).data_ptr()
    a.add_c, a.add_mr, a.add_sums = _p(add_c), _p(add_mr), _p(add_sums)
    if fin is not None:
        a.fin = fin
    a.bias = _p(bias)
    a.pro_out = _p(pro_out)
    a.x = x_ptr if x_ptr is not None else _p(x)
    a.x2 = _p(x2)
    a.y = _p(y)
    a.e1 = e1_ptr if e1_ptr is not None else _p(e1)
    a.w = _p(w)
    a.pro_p, a.pro_gate, a.epi_p, a.epi_gate, a.stats = _p(pro_p), _p(pro_gate), _p(epi_p), _p(epi_gate), _p(stats)
    a.epi_q = _p(epi_q)
    a.M, a.gstride, a.rows_per_sample = M, gstride, rows_per_sample
    a.K, a.Kp, a.N, a.Np = K, cpad(K), N, cpad(N)
    a.w_sn, a.w_sk = w_sn, w_sk
    a.row_mode, a.rpg, a.H, a.W = row_mode, rpg, H, W
    a.pro_mode, a.epi_mode, a.res_mode, a.dtype = pro_mode, epi_mode, res_mode, dtype
    _launch(_detail("c3d_pw_gemm", a), a.M * (a.Kp * (2 if x2 is not None else 1) + a.Np * (2 if (e1 is not None or e1_ptr is not None) else 1)) * _es(dtype), L.lib().c3d_pw_gemm, C.byref(a), _stream())


def pw_weight_image_bytes(N, K, dtype):
    """Bytes of the LDS image of an (N, K) pointwise weight for `c3d_pw_args.w_img` (0: shape not taken)."""
    return int(L.lib().c3d_pw_weight_image_bytes(cpad(N), cpad(K), dtype))


def pw_pack_weights(items, dtype):
    """items: (w, img, N, K, w_sn, w_sk) tuples; one launch per 64 images (`c3d_pw_pack_weights`)."""
    descs = (L.PwPackDesc * len(items))()
    for d, (w, img, N, K, sn, sk) in zip(descs, items):
        d.w, d.img, d.N, d.Np, d.K, d.Kp, d.w_sn, d.w_sk = _p(w), _p(img), N, cpad(N), K, cpad(K), sn, sk
    _launch("c3d_pw_pack_weights", 0, L.lib().c3d_pw_pack_weights, descs, len(items), dtype, _stream())


_wgrad_ws = {}


def _ws(device, n_floats):
    key = (device.index, _stream())   # one workspace per stream: weight gradients may run on a side stream
    buf = _wgrad_ws.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(int(n_floats), dtype=torch.float32, device=device)
        _wgrad_ws[key] = buf
    return buf


def pw_wgrad(p, q, dw, *, M, K, N, dw_sn, dw_sk, dtype, p2=None, p_coef=None, q_mode=PRO_NONE, q_ss=None,
             q_gate=None, rows_per_sample=0, row_mode=ROWS_DENSE, rpg=0, gstride=0, H=0, W=0, dy=0, dx=0,
             q_ptr=None, dw_ptr=None, taps=0, dw_tap_stride=0, chain_ws=None):
    """chain_ws: a caller-owned workspace tensor (c3d_pw_wgrad_ws_floats(N, K) floats) -> a CHAINED launch (c3d_pw_wgrad_args.chain):
    its partials stay pending in chain_ws until the next chained launch or pw_wgrad_flush(); alternate two workspaces."""
    a = L.PwWgradArgs()
    a.p, a.p2 = _p(p), _p(p2)
    a.q = q_ptr if q_ptr is not None else _p(q)
    a.dw = dw_ptr if dw_ptr is not None else _p(dw)
    ws = chain_ws if chain_ws is not None else _ws(p.device, L.lib().c3d_pw_wgrad_ws_floats(N, K))
    a.ws = ws.data_ptr()
    a.chain = 1 if chain_ws is not None else 0
    a.p_coef, a.q_ss, a.q_gate = _p(p_coef), _p(q_ss), _p(q_gate)
    a.M, a.gstride, a.rows_per_sample = M, gstride, rows_per_sample
    a.K, a.Kp, a.N, a.Np = K, cpad(K), N, cpad(N)
    a.dw_sn, a.dw_sk = dw_sn, dw_sk
    a.row_mode, a.rpg, a.H, a.W, a.dy, a.dx = row_mode, rpg, H, W, dy, dx
    a.q_mode, a.dtype = q_mode, dtype
    a.taps, a.dw_tap_stride = taps, dw_tap_stride
    _launch(_detail("c3d_pw_wgrad", a), a.M * (a.Np * (2 if p2 is not None else 1) + a.Kp) * _es(dtype), L.lib().c3d_pw_wgrad, C.byref(a), _stream())


def pw_wgrad_flush():
    """Reduce the pending partials of the last chained pw_wgrad launch (include/change3d_hip.h: c3d_pw_wgrad_flush)."""
    L.check(L.lib().c3d_pw_wgrad_flush(_stream()), "c3d_pw_wgrad_flush")


# ------------------------------------------------------------------------------- BN / SE
def bn_finalize(sums, count, bn, C_, ss, mr, training, stripes=1):
    _launch("c3d_bn_finalize", 0, L.lib().c3d_bn_finalize, _p(sums), stripes, float(count), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
                                    _p(bn.running_var), _p(bn.num_batches_tracked) if training else None,
                                    float(bn.momentum), float(bn.eps), C_, cpad(C_), 1 if training else 0,
                                    _p(ss), _p(mr), _stream())


def bn_se_finalize(nc, B, cnt, bn, se, C_, ss, mr, gate, hid, training):
    if se is not None:
        w1, b1, w2, b2 = se.block[0].weight, se.block[0].bias, se.block[2].weight, se.block[2].bias
        Cr = w1.shape[0]
    else:
        w1 = b1 = w2 = b2 = None
        Cr = 0
    _launch("c3d_bn_se_finalize", 0, L.lib().c3d_bn_se_finalize, _p(nc), B, float(cnt), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
                                       _p(bn.running_var), _p(bn.num_batches_tracked) if training else None,
                                       float(bn.momentum), float(bn.eps), C_, cpad(C_), 1 if training else 0,
                                       _p(w1), _p(b1), _p(w2), _p(b2), Cr, _p(ss), _p(mr), _p(gate), _p(hid),
                                       _stream())


def bn_bwd_coef(dsums, count, bn, mr, C_, coef, stripes=1):
    _launch("c3d_bn_bwd_coef", 0, L.lib().c3d_bn_bwd_coef, _p(dsums), stripes, float(count), _p(bn.weight), _p(mr), C_, cpad(C_), _p(coef),
                                    _p(grad_of(bn.weight)), _p(grad_of(bn.bias)), _stream())


def se_bn_bwd_coef(nc3, ncf, B, cnt, bn, mr, ss, se, gate, hid, C_, coefA, coefC, coefB):
    if se is not None:
        c1, c2 = se.block[0], se.block[2]
        args = (_p(c1.weight), _p(c2.weight), _p(gate), _p(hid), c1.weight.shape[0])
        gargs = (_p(grad_of(c1.weight)), _p(grad_of(c1.bias)), _p(grad_of(c2.weight)), _p(grad_of(c2.bias)))
    else:
        args = (None, None, None, None, 0)
        gargs = (None, None, None, None)
    _launch("c3d_se_bn_bwd_coef", 0, L.lib().c3d_se_bn_bwd_coef, _p(nc3), _p(ncf), B, float(cnt), _p(bn.weight), _p(mr), _p(ss), C_, cpad(C_),
                                       *args, _p(coefA), _p(coefC), _p(coefB), _p(grad_of(bn.weight)),
                                       _p(grad_of(bn.bias)), *gargs, _stream())


# ------------------------------------------------------------------------------ depthwise
def dw_fwd(x, ss, w, y, nc, B, T, H, W, C_, stride, dtype):
    _launch("c3d_dw333_fwd", (x.numel() + y.numel()) * _es(dtype), L.lib().c3d_dw333_fwd, _p(x), _p(ss), _p(w), _p(y), _p(nc), B, T, H, W, C_, cpad(C_), stride, dtype,
                                  _stream())


def dw_fwd_fin(x, fin, w, y, nc, B, T, H, W, C_, stride, dtype):
    _launch("c3d_dw333_fwd", (x.numel() + y.numel()) * _es(dtype), L.lib().c3d_dw333_fwd_fin, _p(x), C.byref(fin), _p(w), _p(y), _p(nc), B, T, H, W, C_, cpad(C_), stride,
                                  dtype, _stream())


def dw_bwd_fused(t1, b, cA, cB, cC, w, a, ss_a, mr_a, t2, dsums, dw, B, T, H, W, C_, dtype, stride=1):
    """Depthwise backward: data gradient, BatchNorm_a-backward sums and weight gradient in one pass."""
    _launch("c3d_dw333_bwd_fused", (t1.numel() + b.numel() + a.numel() + t2.numel()) * _es(dtype), L.lib().c3d_dw333_bwd_fused,
            _p(t1), _p(b), _p(cA), _p(cB), _p(cC), _p(w), _p(a), _p(ss_a), _p(mr_a), _p(t2), _p(dsums), _p(dw),
            B, T, H, W, C_, cpad(C_), stride, dtype, _stream())


def fin_b_bwd(nc3, batch, bn, count, mr):
    """c3d_bn_fin for c3d_dw333_bwd_fused_fin: BatchNorm_b backward coefficients of a block without SqueezeExcitation from
    the per-sample sums nc3 [batch][Cp][3]; d gamma / d beta accumulate into bn.weight.grad / bn.bias.grad."""
    f = L.BnFin()
    f.sums, f.batch, f.gamma, f.mr, f.count = _p(nc3), int(batch), _p(bn.weight), _p(mr), float(count)
    f.running_mean, f.running_var = _p(grad_of(bn.weight)), _p(grad_of(bn.bias))
    return f


def dw_bwd_fused_fin(t1, b, fin_b, w, a, ss_a, mr_a, t2, dsums, dw, B, T, H, W, C_, dtype, stride=1):
    _launch("c3d_dw333_bwd_fused", (t1.numel() + b.numel() + a.numel() + t2.numel()) * _es(dtype), L.lib().c3d_dw333_bwd_fused_fin,
            _p(t1), _p(b), C.byref(fin_b), _p(w), _p(a), _p(ss_a), _p(mr_a), _p(t2), _p(dsums), _p(dw),
            B, T, H, W, C_, cpad(C_), stride, dtype, _stream())


# ----------------------------------------------------------------------------- elementwise
def block_out_fwd(c, ss_c, shortcut, ss_1, mode, y, M, Cp, dtype):
    _launch("c3d_block_out_fwd", M * Cp * (3 if shortcut is not None else 2) * _es(dtype), L.lib().c3d_block_out_fwd, _p(c), _p(ss_c), _p(shortcut), _p(ss_1), mode, _p(y), M, Cp, dtype,
                                      _stream())


def block_out_fwd_fin(c, fin_c, shortcut, fin_1, mode, y, M, C_, dtype):
    _launch("c3d_block_out_fwd", M * cpad(C_) * (3 if shortcut is not None else 2) * _es(dtype), L.lib().c3d_block_out_fwd_fin, _p(c), C.byref(fin_c), _p(shortcut),
                                      C.byref(fin_1) if fin_1 is not None else None, mode, _p(y), M, C_, cpad(C_), dtype, _stream())


def block_out_bwd(dy, y, c, s_bn, g, mr_c, mr_1, dsums_c, dsums_1, M, C_, dtype):
    _launch("c3d_block_out_bwd", M * cpad(C_) * (5 if s_bn is not None else 4) * _es(dtype), L.lib().c3d_block_out_bwd, _p(dy), _p(y), _p(c), _p(s_bn), _p(g), _p(mr_c), _p(mr_1), _p(dsums_c), _p(dsums_1), M, C_,
                                      cpad(C_), dtype, _stream())


def block_out_bwd_fin(dy, y, c, s_bn, g, mr_c, mr_1, dsums_c, dsums_1, M, C_, dtype, fin_c, fin_1):
    _launch("c3d_block_out_bwd", M * cpad(C_) * (5 if s_bn is not None else 4) * _es(dtype), L.lib().c3d_block_out_bwd_fin, _p(dy), _p(y), _p(c), _p(s_bn), _p(g), _p(mr_c), _p(mr_1), _p(dsums_c), _p(dsums_1), M, C_,
            cpad(C_), dtype, C.byref(fin_c), C.byref(fin_1) if fin_1 is not None else None, _stream())


def frame_absdiff(y, d, B, T, HW, Cp, t_pre, t_post, dtype):
    _launch("c3d_frame_absdiff", B * HW * Cp * 3 * _es(dtype), L.lib().c3d_frame_absdiff, _p(y), _p(d), B, T, HW, Cp, t_pre, t_post, dtype, _stream())


def enhance_apply(y, e, out, B, T, HW, Cp, t_mid, dtype):
    _launch("c3d_enhance_apply", B * HW * Cp * (2 * T + 1) * _es(dtype), L.lib().c3d_enhance_apply, _p(y), _p(e), _p(out), B, T, HW, Cp, t_mid, dtype, _stream())


def enhance_bwd_mask(dout, e, de, B, T, HW, Cp, t_mid, dtype):
    _launch("c3d_enhance_bwd_mask", B * HW * Cp * 3 * _es(dtype), L.lib().c3d_enhance_bwd_mask, _p(dout), _p(e), _p(de), B, T, HW, Cp, t_mid, dtype, _stream())


def enhance_bwd_apply(dout, y, dd, dy, B, T, HW, Cp, t_pre, t_post, dtype):
    _launch("c3d_enhance_bwd_apply", B * HW * Cp * (2 * T + 3) * _es(dtype), L.lib().c3d_enhance_bwd_apply, _p(dout), _p(y), _p(dd), _p(dy), B, T, HW, Cp, t_pre, t_post, dtype,
                                          _stream())


def frame_scatter(src, dst, B, T, HW, Cp, t_dst, accumulate, dtype):
    """dst[:, t_dst] (+)= src for channels-last src [B][HW][Cp], dst [B][T][HW][Cp]."""
    _launch("c3d_frame_scatter", B * HW * Cp * (3 if accumulate else 2) * _es(dtype), L.lib().c3d_frame_scatter, _p(src), _p(dst), B, T, HW, Cp,
            t_dst, 1 if accumulate else 0, dtype, _stream())


# ------------------------------------------------------------------------------------ stem
def stem_fwd(x, w_t, w_xy, u, sums, B, T, H, W, dtype):
    _launch("c3d_stem_fwd", x.numel() * 4 + u.numel() * _es(dtype), L.lib().c3d_stem_fwd, _p(x), _p(w_t), _p(w_xy), _p(u), _p(sums), B, T, H, W, dtype, _stream())


def stem_bwd_dv(x, w_t, w_xy, g0, u, coef, dv, dw_xy, B, T, H, W, dtype):
    _launch("c3d_stem_bwd_dv", x.numel() * 4 + 3 * u.numel() * _es(dtype), L.lib().c3d_stem_bwd_dv, _p(x), _p(w_t), _p(w_xy), _p(g0), _p(u), _p(coef), _p(dv), _p(dw_xy), B, T, H,
                                    W, dtype, _stream())


def stem_bwd_wx(x, w_t, dv, dw_t, dP, B, T, H, W, t_first, n_frames, per_sample, dtype):
    _launch("c3d_stem_bwd_wx", x.numel() * 4 + dv.numel() * _es(dtype), L.lib().c3d_stem_bwd_wx, _p(x), _p(w_t), _p(dv), _p(dw_t), _p(dP), B, T, H, W, t_first, n_frames,
                                    1 if per_sample else 0, dtype, _stream())


# --------------------------------------------------------------------------------- decoder
def convT_fwd(inp, w, bias, skip_ptr, skip_bstride, out, B, h, wd, C_, dtype):
    _launch("c3d_convT4s2_fwd", (inp.numel() + 2 * out.numel()) * _es(dtype), L.lib().c3d_convT4s2_fwd, _p(inp), _p(w), _p(bias), skip_ptr, skip_bstride, _p(out), B, h, wd, C_,
                                     dtype, _stream())


def convT_bwd_data(dout, w, din, B, h, wd, C_, dtype):
    _launch("c3d_convT4s2_bwd_data", (dout.numel() + din.numel()) * _es(dtype), L.lib().c3d_convT4s2_bwd_data, _p(dout), _p(w), _p(din), B, h, wd, C_, dtype, _stream())


CONVT_MFMA = os.environ.get("C3D_CONVT_MFMA", "1") != "0"


def convT_wgrad(t, dout, dw, B, h, wd, C_, dtype):
    """ConvTranspose2d k4s2p1 weight gradient on MFMA (bf16 storage only)."""
    ws = _ws(t.device, L.lib().c3d_convT4s2_wgrad_ws_floats(B, h, wd, C_))
    _launch("c3d_convT4s2_wgrad", (t.numel() + dout.numel()) * _es(dtype), L.lib().c3d_convT4s2_wgrad, _p(t), _p(dout), _p(dw),
            ws.data_ptr(), B, h, wd, C_, dtype, _stream())


def col_sum(x, out, M, C_, dtype):
    _launch("c3d_col_sum", M * cpad(C_) * _es(dtype), L.lib().c3d_col_sum, _p(x), _p(out), M, C_, cpad(C_), dtype, _stream())


def head_fwd(x, w, out, B, H, W, C_, NC, has_sigmoid, dtype):
    _launch("c3d_head3x3_fwd", x.numel() * _es(dtype) + out.numel() * 4, L.lib().c3d_head3x3_fwd, _p(x), _p(w), _p(out), B, H, W, C_, NC, 1 if has_sigmoid else 0, dtype,
                                    _stream())


def head_bwd(dout, prob, x, w, dx, dw, B, H, W, C_, NC, has_sigmoid, dtype):
    ws = _ws(x.device, max(1, L.lib().c3d_head3x3_bwd_ws_floats(B, H, W, NC)))
    _launch("c3d_head3x3_bwd", 2 * x.numel() * _es(dtype) + 2 * dout.numel() * 4, L.lib().c3d_head3x3_bwd, _p(dout), _p(prob), _p(x), _p(w), _p(dx), _p(dw), ws.data_ptr(), B, H, W, C_, NC,
                                    1 if has_sigmoid else 0, dtype, _stream())


# ------------------------------------------------------------------------------ train shell
def bce_dice_fwd(prob, target, sums4, loss):
    _launch("c3d_bce_dice_fwd", prob.numel() * 8, L.lib().c3d_bce_dice_fwd, _p(prob), _p(target), prob.numel(), _p(sums4), _p(loss), _stream())


def bce_dice_bwd(prob, target, sums4, dloss, dprob):
    _launch("c3d_bce_dice_bwd", prob.numel() * 12, L.lib().c3d_bce_dice_bwd, _p(prob), _p(target), _p(sums4), _p(dloss), prob.numel(), _p(dprob),
                                     _stream())


def _nchw_view(x):
    """(B, NC, HW, batch stride, channel stride) of an f32 NCHW view whose pixels are contiguous."""
    B, NC, H, W = x.shape
    assert x.dtype == torch.float32 and (W == 1 or x.stride(3) == 1) and (H == 1 or x.stride(2) == W)
    return B, NC, H * W, x.stride(0), x.stride(1)


def ce2d_fwd(logits, target, ignore_index, sums2, loss):
    B, NC, HW, bs, cs = _nchw_view(logits)
    _launch("c3d_ce2d_fwd", B * HW * (NC * 4 + 8), L.lib().c3d_ce2d_fwd, _p(logits), _p(target), B, NC, HW, bs, cs,
            ignore_index, _p(sums2), _p(loss), _stream())


def ce2d_bwd(logits, target, ignore_index, sums2, dloss, dlogits):
    B, NC, HW, bs, cs = _nchw_view(logits)
    _launch("c3d_ce2d_bwd", B * HW * (NC * 8 + 8), L.lib().c3d_ce2d_bwd, _p(logits), _p(target), _p(sums2), _p(dloss),
            B, NC, HW, bs, cs, ignore_index, _p(dlogits), _stream())


def cossim_fwd(x1, x2, label_change, sums1, loss):
    B, NC, HW, bs1, cs1 = _nchw_view(x1)
    _, _, _, bs2, cs2 = _nchw_view(x2)
    _launch("c3d_cossim_fwd", B * HW * (NC * 8 + 8), L.lib().c3d_cossim_fwd, _p(x1), _p(x2), _p(label_change), B, NC, HW,
            bs1, cs1, bs2, cs2, _p(sums1), _p(loss), _stream())


def cossim_bwd(x1, x2, label_change, dloss, dx1, dx2):
    B, NC, HW, bs1, cs1 = _nchw_view(x1)
    _, _, _, bs2, cs2 = _nchw_view(x2)
    _launch("c3d_cossim_bwd", B * HW * (NC * 16 + 8), L.lib().c3d_cossim_bwd, _p(x1), _p(x2), _p(label_change), _p(dloss),
            B, NC, HW, bs1, cs1, bs2, cs2, _p(dx1), _p(dx2), _stream())


def adam_step(param, grad, exp_avg, exp_avg_sq, n, hp_dev, lr, bc1, bc2_sqrt, beta1, beta2, eps, wd):
    _launch("c3d_adam_step", n * 28, L.lib().c3d_adam_step, _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(hp_dev), lr, bc1, bc2_sqrt,
                                  beta1, beta2, eps, wd, _stream())


def confusion2(prob, target, cm4):
    _launch("c3d_confusion2", prob.numel() * 8, L.lib().c3d_confusion2, _p(prob), _p(target), prob.numel(), _p(cm4), _stream())


def hist2d(a, b, n, hist):
    """hist (uint64-as-int64 [n*n + 1], device) += joint histogram of the int64 label maps a (rows) and b (columns)."""
    _launch("c3d_hist2d", a.numel() * 16, L.lib().c3d_hist2d, _p(a), _p(b), a.numel(), n, _p(hist), _stream())


# ------------------------------------------------------------------------------ stage driver
class StageBinding:
    """ctypes descriptor (include/change3d_hip.h: c3d_stage_desc) of one residual stage, bound to the tensors of
    the reference-shaped module tree (`X3DResStage.res_blocks[j].branch2.conv_a.weight`, ...).  Pointers are
    re-read on every call (parameters move when a ParamArena is built or the module is moved; gradients move
    when they are reset): ~600 attribute reads per step, no struct write unless something changed."""

    def __init__(self, stage):
        blocks = list(stage.res_blocks)
        self.n = len(blocks)
        self.blocks = (L.BlockDesc * self.n)()
        self.desc = L.StageDesc()
        self.desc.n_blocks = self.n
        self.desc.blocks = C.cast(self.blocks, C.POINTER(L.BlockDesc))
        self.params, self.buffers = [], []      # (struct, field, grad_struct, grad_field, parameter) / (struct, field, buffer)
        self._grad_refs = None
        for bd, blk in zip(self.blocks, blocks):
            b2 = blk.branch2
            se = b2.norm_b[1] if blk.use_se else None
            bd.cin, bd.cinner, bd.cout, bd.stride = blk.cin, blk.cinner, blk.cout, blk.stride
            bd.se_width = se.block[0].weight.shape[0] if se is not None else 0
            bd.has_sc_conv = 1 if blk.branch1_conv is not None else 0
            bd.has_sc_bn = 1 if blk.branch1_norm is not None else 0
            self.params += [(bd, "w_a", bd, "dw_a", b2.conv_a), (bd, "w_b", bd, "dw_b", b2.conv_b),
                            (bd, "w_c", bd, "dw_c", b2.conv_c)]
            if blk.branch1_conv is not None:
                self.params.append((bd, "w_sc", bd, "dw_sc", blk.branch1_conv))
            bns = [(bd.bn_a, b2.norm_a), (bd.bn_b, b2.norm_b[0]), (bd.bn_c, b2.norm_c)]
            if blk.branch1_norm is not None:
                bns.append((bd.bn_sc, blk.branch1_norm))
            for st, bn in bns:
                self.params += [(st, "gamma", st, "dgamma", (bn, "weight")), (st, "beta", st, "dbeta", (bn, "bias"))]
                self.buffers += [(st, "running_mean", bn, "running_mean"), (st, "running_var", bn, "running_var"),
                                 (st, "num_batches_tracked", bn, "num_batches_tracked")]
            if se is not None:
                self.params += [(bd, "se_w1", bd, "dse_w1", (se.block[0], "weight")), (bd, "se_b1", bd, "dse_b1", (se.block[0], "bias")),
                                (bd, "se_w2", bd, "dse_w2", (se.block[2], "weight")), (bd, "se_b2", bd, "dse_b2", (se.block[2], "bias"))]
        # resolve to the Parameter / buffer OBJECTS once (nn.Module.__getattr__ costs ~1 us per lookup, 600 lookups
        # per stage pass): Parameters keep their identity across .to() / load_state_dict() / arena placement;
        # buffers are REPLACED by Module._apply, which is why X3DResStage._apply drops this binding
        self.params = [(s, f, gs, gf, getattr(*(m if isinstance(m, tuple) else (m, "weight")))) for s, f, gs, gf, m in self.params]
        self.buffers = [(s, f, getattr(mod, attr)) for s, f, mod, attr in self.buffers]
        self._ptr_cache = {}
        self._last = {False: None, True: None}   # pointer signatures of the forward / backward bindings

    def _set(self, st, field, ptr):
        key = (id(st), field)
        if self._ptr_cache.get(key) != ptr:
            setattr(st, field, ptr)
            self._ptr_cache[key] = ptr

    def refresh(self, B, T, H, W, dtype, training, momentum, eps, with_grads):
        d = self.desc
        d.B, d.T, d.H, d.W, d.dtype, d.training = B, T, H, W, dtype, 1 if training else 0
        d.momentum, d.eps = momentum, eps
        # pointer signature of everything bound (data, gradients, buffers): ~0.1 us per tensor; the ctypes structs are
        # rewritten only when it changed (first call, new arena, gradients re-created, module moved)
        if with_grads:
            grads = [p.grad if p.grad is not None else grad_of(p) for _, _, _, _, p in self.params]
            sig = tuple([p.data_ptr() for _, _, _, _, p in self.params] + [g.data_ptr() for g in grads] +
                        [b.data_ptr() for _, _, b in self.buffers])
        else:
            grads = None
            sig = tuple([p.data_ptr() for _, _, _, _, p in self.params] + [b.data_ptr() for _, _, b in self.buffers])
        if sig != self._last[with_grads]:
            for i, (st, f, gs, gf, p) in enumerate(self.params):
                self._set(st, f, p.data_ptr())
                if grads is not None:
                    self._set(gs, gf, grads[i].data_ptr())
            for st, f, b in self.buffers:
                self._set(st, f, b.data_ptr())
            self._last[with_grads] = sig
        self._grad_refs = grads   # keep the bound gradient tensors alive while kernels may write them
        return d

    def sizes(self):
        out = [C.c_int64() for _ in range(4)]
        rc = L.lib().c3d_stage_ws_bytes(C.byref(self.desc), *[C.byref(v) for v in out])
        if rc == -2:   # C3D_E_UNSUPPORTED: the one geometry limit of the stage API (include/change3d_hip.h)
            d = self.desc
            raise L.Change3DHipError(
                f"stage geometry B={d.B} T={d.T} {d.H}x{d.W} is too large: an activation tensor of a stage with <= 224 channels "
                f"must stay under 2 GiB (32-bit offsets in the pointwise kernels); reduce the per-GPU batch "
                f"(bf16 at 256x256, T=3: B <= 96; f32: B <= 48)")
        if rc != 0:
            raise L.Change3DHipError(f"c3d_stage_ws_bytes failed with code {rc}")
        return tuple(int(v.value) for v in out)   # ws_fwd, ws_bwd, y, dx bytes

    def saved(self, blk, name):
        off, n = C.c_int64(), C.c_int64()
        rc = L.lib().c3d_stage_saved(C.byref(self.desc), blk, name.encode(), C.byref(off), C.byref(n))
        return None if rc != 0 else (int(off.value), int(n.value))


def stage_fold_sizes(binding):
    a, b = C.c_int64(), C.c_int64()
    rc = L.lib().c3d_stage_fold_bytes(C.byref(binding.desc), C.byref(a), C.byref(b))
    if rc != 0:
        raise L.Change3DHipError(f"c3d_stage_fold_bytes failed with code {rc}")
    return int(a.value), int(b.value)


def stage_fold_bn(binding, fold):
    rc = L.lib().c3d_stage_fold_bn(C.byref(binding.desc), fold.data_ptr(), _stream())
    if rc != 0:
        raise L.Change3DHipError(f"c3d_stage_fold_bn failed with code {rc}")


def stage_fwd_folded(binding, fold, x, ws, y):
    rc = L.lib().c3d_stage_fwd_folded(C.byref(binding.desc), fold.data_ptr(), x.data_ptr(), ws.data_ptr(), y.data_ptr(),
                                      _stream())
    if rc != 0:
        raise L.Change3DHipError(f"c3d_stage_fwd_folded failed with code {rc}")


def stage_fwd(binding, x, ws, y):
    rc = L.lib().c3d_stage_fwd(C.byref(binding.desc), x.data_ptr(), ws.data_ptr(), y.data_ptr(), _stream())
    if rc != 0:
        raise L.Change3DHipError(f"c3d_stage_fwd failed with code {rc}")


def stage_bwd(binding, x, y, dy, ws, wb, dx):
    rc = L.lib().c3d_stage_bwd(C.byref(binding.desc), x.data_ptr(), y.data_ptr(), dy.data_ptr(), ws.data_ptr(),
                               wb.data_ptr(), dx.data_ptr(), _stream())
    if rc != 0:
        raise L.Change3DHipError(f"c3d_stage_bwd failed with code {rc}")


# ------------------------------------------------------------------------------ input pipeline
def bcd_preprocess(image6, label, flags, mean6, std6, pre, post, label_out, B, H, W):
    _launch("c3d_bcd_preprocess", B * H * W * (7 + 28), L.lib().c3d_bcd_preprocess, _p(image6), _p(label), _p(flags),
            _p(mean6), _p(std6), _p(pre), _p(post), _p(label_out), B, H, W, _stream())


def scd_label_preprocess(label3, flags, out, B, H, W):
    _launch("c3d_scd_label_preprocess", B * H * W * (3 + 24), L.lib().c3d_scd_label_preprocess, _p(label3), _p(flags), _p(out),
            B, H, W, _stream())


def cc_preprocess(img, swap, lut, pre, post, B, H, W):
    _launch("c3d_cc_preprocess", B * H * W * (6 + 24), L.lib().c3d_cc_preprocess, _p(img), _p(swap), _p(lut), _p(pre), _p(post),
            B, H, W, _stream())


def build_clip(pre, post, frames, clip, B, K, H, W):
    _launch("c3d_build_clip", clip.numel() * 8, L.lib().c3d_build_clip, _p(pre), _p(post), _p(frames), _p(clip), B, K, H, W,
            _stream())


# ------------------------------------------------------------------------------ caption decoder (change captioning)
def cap_embed_fwd(tokens, emb, pe, out, B, Lq, D, V, p, seed, dtype):
    _launch("c3d_cap_embed_fwd", out.numel() * _es(dtype), L.lib().c3d_cap_embed_fwd, _p(tokens), _p(emb), _p(pe), _p(out), B, Lq, D, V,
            float(p), int(seed), dtype, _stream())


def cap_embed_bwd(tokens, dout, demb, B, Lq, D, V, p, seed, dtype):
    _launch("c3d_cap_embed_bwd", dout.numel() * _es(dtype), L.lib().c3d_cap_embed_bwd, _p(tokens), _p(dout), _p(demb), B, Lq, D, V,
            float(p), int(seed), dtype, _stream())


def cap_dropout(x, y, rows, D, p, seed, dtype):
    _launch("c3d_cap_dropout", 2 * x.numel() * _es(dtype), L.lib().c3d_cap_dropout, _p(x), _p(y), rows, D, float(p), int(seed), dtype, _stream())


def cap_layernorm_fwd(x, a, ln, y, mr, rows, D, dtype):
    _launch("c3d_cap_layernorm_fwd", 3 * x.numel() * _es(dtype), L.lib().c3d_cap_layernorm_fwd, _p(x), _p(a), _p(ln.weight), _p(ln.bias), _p(y),
            _p(mr), rows, D, float(ln.eps), dtype, _stream())


def cap_layernorm_bwd(x, a, dy, ln, mr, dx, rows, D, dtype):
    _launch("c3d_cap_layernorm_bwd", 4 * x.numel() * _es(dtype), L.lib().c3d_cap_layernorm_bwd, _p(x), _p(a), _p(dy), _p(ln.weight), _p(mr), _p(dx),
            _p(grad_of(ln.weight)), _p(grad_of(ln.bias)), rows, D, dtype, _stream())


def cap_attn_fwd(q, k, v, ldq, ldk, ldv, o, ldo, P, B, H, Lq, Lk, hd, scale, causal, p, seed, dtype, q_off=0, k_off=0, v_off=0):
    es = _es(dtype)
    _launch("c3d_cap_attn_fwd", (q.numel() + k.numel() + o.numel()) * es, L.lib().c3d_cap_attn_fwd, q.data_ptr() + q_off * es,
            k.data_ptr() + k_off * es, v.data_ptr() + v_off * es, ldq, ldk, ldv, _p(o), ldo, _p(P), B, H, Lq, Lk, hd, float(scale),
            1 if causal else 0, float(p), int(seed), dtype, _stream())


def cap_attn_bwd(q, k, v, ldq, ldk, ldv, dout, ldo, P, dq, dk, dv, lddq, lddk, lddv, B, H, Lq, Lk, hd, scale, p, seed, dtype,
                 q_off=0, k_off=0, v_off=0, dq_off=0, dk_off=0, dv_off=0):
    es = _es(dtype)
    _launch("c3d_cap_attn_bwd", (q.numel() + k.numel() + dout.numel()) * 2 * es, L.lib().c3d_cap_attn_bwd, q.data_ptr() + q_off * es,
            k.data_ptr() + k_off * es, v.data_ptr() + v_off * es, ldq, ldk, ldv, _p(dout), ldo, _p(P), dq.data_ptr() + dq_off * es,
            dk.data_ptr() + dk_off * es, dv.data_ptr() + dv_off * es, lddq, lddk, lddv, B, H, Lq, Lk, hd, float(scale), float(p),
            int(seed), dtype, _stream())


def cap_ce_fwd(logits, caps, declen, acc2, lse, loss, B, Lq, V, ignore_index, dtype):
    _launch("c3d_cap_ce_fwd", logits.numel() * _es(dtype), L.lib().c3d_cap_ce_fwd, _p(logits), _p(caps), _p(declen), _p(acc2), _p(lse), _p(loss),
            B, Lq, V, ignore_index, dtype, _stream())


def cap_ce_bwd(logits, caps, declen, acc2, lse, dloss, dlogits, B, Lq, V, ignore_index, dtype):
    _launch("c3d_cap_ce_bwd", 2 * logits.numel() * _es(dtype), L.lib().c3d_cap_ce_bwd, _p(logits), _p(caps), _p(declen), _p(acc2), _p(lse), _p(dloss),
            _p(dlogits), B, Lq, V, ignore_index, dtype, _stream())


def clamp_(g, limit):
    _launch("c3d_clamp_", g.numel() * 8, L.lib().c3d_clamp_, _p(g), g.numel(), float(limit), _stream())


def linear_fwd(x, weight, bias, y, M, K, N, dtype):
    """y[M][Np] = x[M][Kp] @ weight[N][K]^T + bias (nn.Linear / in_proj slices)."""
    pw_gemm(x, weight, y, M=M, K=K, N=N, w_sn=weight.stride(0), w_sk=1, dtype=dtype, bias=bias)


def linear_bwd(x, weight, dy, dx, gw, gb, M, K, N, dtype, accumulate_dx=None):
    """dx = dy @ weight (written, or added to `accumulate_dx`), gw += dy^T x, gb += colsum(dy)."""
    if dx is not None:
        pw_gemm(dy, weight, dx, M=M, K=N, N=K, w_sn=1, w_sk=weight.stride(0), dtype=dtype,
                epi_mode=EPI_ADD if accumulate_dx is not None else EPI_STORE, e1=accumulate_dx, res_mode=0)
    pw_wgrad(dy, x, gw, M=M, K=K, N=N, dw_sn=gw.stride(0), dw_sk=1, dtype=dtype)
    if gb is not None:
        col_sum(dy, gb, M, N, dtype)
