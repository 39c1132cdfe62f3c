// MFMA versions of the stem kernels (stem_mfma.hip), dispatched from the c3d_stem_* entries in stem.hip.
// Internal to the library (not part of the C ABI).  Return C3D_E_UNSUPPORTED when a shape has no instantiation.
#pragma once
#include <hip/hip_runtime.h>

bool c3d_stem_mfma_enabled();   // c3d_set_option(C3D_OPT_STEM_MFMA, 0) selects the scalar-FMA kernels
int c3d_stem_fwd_mfma(const float* x, const float* w_t, const float* w_xy, void* u, double* sums, int B, int T, int H,
                      int W, int dtype, hipStream_t s);
int c3d_stem_bwd_dv_mfma(const float* x, const float* w_t, const float* w_xy, const void* g0, const void* u,
                         const float* coef, void* dv, float* dw_xy, int B, int T, int H, int W, int dtype, hipStream_t s);
int c3d_stem_bwd_wx_mfma(const float* x, const float* w_t, const void* dv, float* dw_t, float* dP, int B, int T, int H,
                         int W, int t_first, int n_frames, int per_sample, int dtype, hipStream_t s);
