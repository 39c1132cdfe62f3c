"""ctypes binding of libchange3d_hip.so (the C ABI declared in include/change3d_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There is NO
fallback: if the shared object is missing or a symbol is absent, import-time use raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C3D_LIB") or os.path.join(_HERE, "lib", "libchange3d_hip.so")  # C3D_LIB: instrumented builds (tools/)

DT_F32, DT_BF16 = 0, 1
PRO_NONE, PRO_BN_SE_SWISH, PRO_AFFINE2 = 0, 1, 2
EPI_STORE, EPI_STATS, EPI_SWISH_SE_BWD, EPI_ADD = 0, 1, 2, 3
ROWS_DENSE, ROWS_FRAME, ROWS_STRIDE2, ROWS_S2SHIFT = 0, 1, 2, 3
SC_NONE, SC_IDENTITY, SC_BN, SC_RAW = 0, 1, 2, 3
STAT_STRIPES = 16
OPT_SIDE_STREAM, OPT_STEM_MFMA, OPT_CONVT_MFMA, OPT_FUSE_WGRAD, OPT_FOLD_SE, OPT_MASK_IN_DGRAD, OPT_DW_RING, OPT_PW_WGRAD_V2, OPT_DW_FWD_HV, OPT_PW_CFWD, OPT_PW_CDG = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
STAGE_SEPARATE_FINALIZE, STAGE_NO_WEIGHT_IMAGES, STAGE_SEPARATE_RESIDUAL, STAGE_SEPARATE_WGRAD = 1, 2, 4, 8

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class BnFin(C.Structure):
    _fields_ = [("ticket", vp), ("gamma", vp), ("beta", vp), ("running_mean", vp), ("running_var", vp), ("nbt", vp),
                ("ss", vp), ("mr", vp), ("count", f64), ("momentum", f32), ("eps", f32), ("training", i32),
                ("batch", i32), ("sums", vp)]


class PwArgs(C.Structure):
    _fields_ = [("x", vp), ("x2", vp), ("y", vp), ("e1", vp), ("w", vp), ("pro_p", vp), ("pro_gate", vp),
                ("epi_p", vp), ("epi_gate", vp), ("epi_q", vp), ("stats", vp),
                ("M", i64), ("gstride", i64), ("rows_per_sample", i64),
                ("K", i32), ("Kp", i32), ("N", i32), ("Np", i32), ("w_sn", i32), ("w_sk", i32),
                ("row_mode", i32), ("rpg", i32), ("H", i32), ("W", i32),
                ("pro_mode", i32), ("epi_mode", i32), ("res_mode", i32), ("dtype", i32), ("fin", BnFin), ("bias", vp), ("pro_out", vp), ("w_img", vp),
                ("wg_x3", vp), ("wg_dw", vp), ("wg_ws", vp), ("wg_mode", i32), ("wg_mask_out", i32),
                ("se_w1", vp), ("se_b1", vp), ("se_w2", vp), ("se_b2", vp), ("se_hid", vp), ("se_cr", i32), ("se_reserved", i32),
                ("add_c", vp), ("add_mr", vp), ("add_sums", vp)]


class PwPackDesc(C.Structure):
    _fields_ = [("w", vp), ("img", vp), ("N", i32), ("Np", i32), ("K", i32), ("Kp", i32), ("w_sn", i32), ("w_sk", i32)]


class PwWgradArgs(C.Structure):
    _fields_ = [("p", vp), ("p2", vp), ("q", vp), ("dw", vp), ("ws", vp), ("p_coef", vp), ("q_ss", vp),
                ("q_gate", vp),
                ("M", i64), ("gstride", i64), ("rows_per_sample", i64),
                ("K", i32), ("Kp", i32), ("N", i32), ("Np", i32), ("dw_sn", i32), ("dw_sk", i32),
                ("row_mode", i32), ("rpg", i32), ("H", i32), ("W", i32), ("dy", i32), ("dx", i32),
                ("q_mode", i32), ("dtype", i32), ("taps", i32), ("dw_tap_stride", i32), ("p_fin", BnFin),
                ("chain", i32), ("reserved_", i32)]


class BnPtrs(C.Structure):
    _fields_ = [("gamma", vp), ("beta", vp), ("running_mean", vp), ("running_var", vp), ("num_batches_tracked", vp),
                ("dgamma", vp), ("dbeta", vp)]


class BlockDesc(C.Structure):
    _fields_ = [("cin", i32), ("cinner", i32), ("cout", i32), ("stride", i32), ("se_width", i32),
                ("has_sc_conv", i32), ("has_sc_bn", i32), ("reserved", i32),
                ("w_a", vp), ("w_b", vp), ("w_c", vp), ("w_sc", vp),
                ("dw_a", vp), ("dw_b", vp), ("dw_c", vp), ("dw_sc", vp),
                ("bn_a", BnPtrs), ("bn_b", BnPtrs), ("bn_c", BnPtrs), ("bn_sc", BnPtrs),
                ("se_w1", vp), ("se_b1", vp), ("se_w2", vp), ("se_b2", vp),
                ("dse_w1", vp), ("dse_b1", vp), ("dse_w2", vp), ("dse_b2", vp)]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", i32), ("reserved", i32), ("ms_total", f32), ("reserved2", f32),
                ("bytes_total", f64)]


class StageDesc(C.Structure):
    _fields_ = [("n_blocks", i32), ("B", i32), ("T", i32), ("H", i32), ("W", i32), ("dtype", i32),
                ("training", i32), ("flags", i32), ("momentum", f32), ("eps", f32),
                ("blocks", C.POINTER(BlockDesc))]


# name -> (restype, argtypes); every function declared in include/change3d_hip.h
SIGNATURES = {
    "c3d_abi_version": (i32, []),
    "c3d_build_info": (C.c_char_p, []),
    "c3d_device_cus": (i32, []),
    "c3d_pw_gemm": (i32, [C.POINTER(PwArgs), vp]),
    "c3d_pw_weight_image_bytes": (i64, [i32, i32, i32]),
    "c3d_pw_pack_weights": (i32, [C.POINTER(PwPackDesc), i32, i32, vp]),
    "c3d_pw_wgrad_ws_floats": (i64, [i32, i32]),
    "c3d_pw_gemm_wg_ws_floats": (i64, [i32, i32]),
    "c3d_pw_wgrad": (i32, [C.POINTER(PwWgradArgs), vp]),
    "c3d_pw_wgrad_flush": (i32, [vp]),
    "c3d_bn_finalize": (i32, [vp, i32, f64, vp, vp, vp, vp, vp, f32, f32, i32, i32, i32, vp, vp, vp]),
    "c3d_bn_se_finalize": (i32, [vp, i32, f64, vp, vp, vp, vp, vp, f32, f32, i32, i32, i32, vp, vp, vp, vp,
                                 i32, vp, vp, vp, vp, vp]),
    "c3d_bn_bwd_coef": (i32, [vp, i32, f64, vp, vp, i32, i32, vp, vp, vp, vp]),
    "c3d_se_bn_bwd_coef": (i32, [vp, vp, i32, f64, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                                 vp, vp, vp, vp, vp, vp]),
    "c3d_dw333_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_dw333_fwd_fin": (i32, [vp, C.POINTER(BnFin), vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_dw333_bwd_fused": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_dw333_bwd_fused_fin": (i32, [vp, vp, C.POINTER(BnFin), vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_block_out_fwd": (i32, [vp, vp, vp, vp, i32, vp, i64, i32, i32, vp]),
    "c3d_block_out_fwd_fin": (i32, [vp, C.POINTER(BnFin), vp, C.POINTER(BnFin), i32, vp, i64, i32, i32, i32, vp]),
    "c3d_block_out_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "c3d_block_out_bwd_fin": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, C.POINTER(BnFin),
                                    C.POINTER(BnFin), vp]),
    "c3d_frame_absdiff": (i32, [vp, vp, i32, i32, i64, i32, i32, i32, i32, vp]),
    "c3d_enhance_apply": (i32, [vp, vp, vp, i32, i32, i64, i32, i32, i32, vp]),
    "c3d_enhance_bwd_mask": (i32, [vp, vp, vp, i32, i32, i64, i32, i32, i32, vp]),
    "c3d_enhance_bwd_apply": (i32, [vp, vp, vp, vp, i32, i32, i64, i32, i32, i32, i32, vp]),
    "c3d_frame_scatter": (i32, [vp, vp, i32, i32, i64, i32, i32, i32, i32, vp]),
    "c3d_stem_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "c3d_stem_bwd_dv": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "c3d_stem_bwd_wx": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_convT4s2_fwd": (i32, [vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, vp]),
    "c3d_convT4s2_bwd_data": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "c3d_convT4s2_wgrad_ws_floats": (i64, [i32, i32, i32, i32]),
    "c3d_convT4s2_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "c3d_col_sum": (i32, [vp, vp, i64, i32, i32, i32, vp]),
    "c3d_head3x3_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_head3x3_bwd_ws_floats": (i64, [i32, i32, i32, i32]),
    "c3d_head3x3_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "c3d_bce_dice_fwd": (i32, [vp, vp, i64, vp, vp, vp]),
    "c3d_bce_dice_bwd": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "c3d_ce2d_fwd": (i32, [vp, vp, i64, i32, i64, i64, i64, i64, vp, vp, vp]),
    "c3d_ce2d_bwd": (i32, [vp, vp, vp, vp, i64, i32, i64, i64, i64, i64, vp, vp]),
    "c3d_cossim_fwd": (i32, [vp, vp, vp, i64, i32, i64, i64, i64, i64, i64, vp, vp, vp]),
    "c3d_cossim_bwd": (i32, [vp, vp, vp, vp, i64, i32, i64, i64, i64, i64, i64, vp, vp, vp]),
    "c3d_adam_step": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, f32, vp]),
    "c3d_confusion2": (i32, [vp, vp, i64, vp, vp]),
    "c3d_hist2d": (i32, [vp, vp, i64, i32, vp, vp]),
    "c3d_build_clip": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "c3d_bcd_preprocess": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "c3d_scd_label_preprocess": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "c3d_cc_preprocess": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "c3d_cap_embed_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, i32, vp]),
    "c3d_cap_embed_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, i32, vp]),
    "c3d_cap_dropout": (i32, [vp, vp, i64, i32, f32, C.c_uint64, i32, vp]),
    "c3d_cap_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]),
    "c3d_cap_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "c3d_cap_attn_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, f32, i32, f32, C.c_uint64, i32, vp]),
    "c3d_cap_attn_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32,
                               C.c_uint64, i32, vp]),
    "c3d_cap_ce_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i32, vp]),
    "c3d_cap_ce_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i32, vp]),
    "c3d_clamp_": (i32, [vp, i64, f32, vp]),
    "c3d_stage_ws_bytes": (i32, [C.POINTER(StageDesc), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "c3d_stage_fwd": (i32, [C.POINTER(StageDesc), vp, vp, vp, vp]),
    "c3d_stage_bwd": (i32, [C.POINTER(StageDesc), vp, vp, vp, vp, vp, vp, vp]),
    "c3d_side_join": (i32, [vp]),
    "c3d_set_option": (i32, [i32, i32]),
    "c3d_prof_begin": (i32, [i32]),
    "c3d_prof_end": (i32, [C.POINTER(ProfRow), i32, C.POINTER(i32)]),
    "c3d_stage_fold_bytes": (i32, [C.POINTER(StageDesc), C.POINTER(i64), C.POINTER(i64)]),
    "c3d_stage_fold_bn": (i32, [C.POINTER(StageDesc), vp, vp]),
    "c3d_stage_fwd_folded": (i32, [C.POINTER(StageDesc), vp, vp, vp, vp, vp]),
    "c3d_stage_saved": (i32, [C.POINTER(StageDesc), i32, C.c_char_p, C.POINTER(i64), C.POINTER(i64)]),
}

_lib = None


class Change3DHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library.  Raises loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise Change3DHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` at the repo root. "
                f"There is no CPU/PyTorch fallback for the Change3D hot path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise Change3DHipError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        # host-side convenience (the library itself never reads the environment): C3D_WGRAD_SIDE=0 keeps the weight
        # gradients on the caller's stream, e.g. for one-kernel-at-a-time rocprofv3 traces of unmodified commands
        if os.environ.get("C3D_WGRAD_SIDE", "1") == "0":
            handle.c3d_set_option(OPT_SIDE_STREAM, 0)
        _lib = handle
    return _lib


def check_exports():
    """Used by build() and the CPU test-suite: every declared symbol must be exported."""
    handle = lib()
    assert handle.c3d_abi_version() == 1
    return sorted(SIGNATURES)


def check(rc, what):
    if rc != 0:
        raise Change3DHipError(f"{what} failed with code {rc}")


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources and the C header: what a counter summary under profiles/ was taken at
    (tools/summarize_rocprof.py stores it, bench.py compares it -- the GPU box has no .git to ask for a commit hash)."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, "csrc", "*.hip")) + glob.glob(os.path.join(root, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(root), "include", "change3d_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
