// Launch hints passed between translation units of the library (not part of the C ABI).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "../../include/change3d_hip.h"

// Set by the stage driver around launches it places on its side stream.  Single-round kernels (one long-running
// workgroup per CU for the whole launch) read it to leave CUs free: a side kernel that occupies every CU keeps EVERY
// kernel of the data-gradient chain that becomes ready meanwhile -- including its 1-8 workgroup coefficient kernels --
// waiting for its whole duration (measured on MI355X, B=32 bf16: depthwise weight gradient on 128 instead of 256
// workgroups: step 35.6 -> 34.1 ms, and the 40 us stall around c3d_se_bn_bwd_coef disappears).
// The pointwise weight gradient takes 7/8 of the CUs in the same situation (round 2: 256 -> 192 workgroups: 32.43 -> 31.69 ms; round 5: 192 -> 160: 23.18 -> 22.86 ms,
// then -- with c3d_block_out_bwd folded into the conv_a data gradient: nothing small left to run beside a narrow grid -- 160 -> 224: 22.70 -> 22.29 ms;
// profiles/r02_side_stream_width_final.json, csrc/pw_wgrad.hip).
extern thread_local int c3d_side_launch;

// c3d_set_option (stage_driver.hip): kernel-family selectors with a parity test between the two implementations
extern int c3d_option_stem_mfma;    // 2: + c3d_stem_bwd_wx of bf16 storage on the bf16 matrix cores, 1: stem on the f32 matrix cores
                                    // (stem_mfma.hip), 0: scalar-FMA kernels (stem.hip)
extern int c3d_option_convt_mfma;   // 1: bf16 ConvTranspose2d on the matrix cores (convt_mfma.hip), 0: decoder.hip's
extern int c3d_option_dw_ring;      // C3D_OPT_DW_RING (include/change3d_hip.h): LDS-DMA ring variant of the bf16 stride-1 depthwise backward
extern int c3d_option_dw_fwd_hv;     // C3D_OPT_DW_FWD_HV: stride-1 three-frame depthwise forward on half-vector lanes (bit 0 bf16, bit 1 f32 storage)
extern int c3d_option_pw_cfwd;       // C3D_OPT_PW_CFWD: conv_c forward of the training path on csrc/pw_cfwd.hip
extern int c3d_option_pw_cdg;        // C3D_OPT_PW_CDG: conv_a (bit 0) / conv_c (bit 1) data + weight gradient on csrc/pw_cdgrad.hip
int c3d_detail_pw_cdg_a(const c3d_pw_args* args, void* stream);
bool c3d_detail_pw_cdg_a_supported(int Kp, int Np, int64_t M);
int c3d_detail_pw_cdg_c(const c3d_pw_args* args, void* stream);
bool c3d_detail_pw_cdg_c_supported(int Kp, int Np, int64_t M, int64_t rows_per_sample);
// Set by the stage driver around a c3d_pw_gemm call: the cooperative kernel then leaves its weight-gradient partials in
// wg_ws WITHOUT launching the reducer and reports their count in c3d_cdg_parts (0: another kernel ran, which reduced on its
// own) -- the driver launches the reducer on its side stream (c3d_detail_pw_wgrad_reduce).
extern thread_local int c3d_cdg_defer_reduce, c3d_cdg_parts;
int c3d_detail_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int parts, int sn, int sk, hipStream_t stream);   // pw_wgrad.hip
extern int c3d_option_pw_wgrad_v2;  // C3D_OPT_PW_WGRAD_V2: 1 = c3d_pw_wgrad of bf16 dense rows on csrc/pw_wgrad_v2.hip, 0 = pw_wgrad.hip's kernel
void c3d_detail_pw_wgrad_v2_drop();  // forget pending partials of a chained c3d_pw_wgrad launch WITHOUT reducing them (start of a stage pass)
