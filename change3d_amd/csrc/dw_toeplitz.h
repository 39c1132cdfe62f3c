// Toeplitz-MFMA depthwise forward (dw_toeplitz.hip), dispatched from c3d_dw333_fwd in dw_conv.hip.  Internal.
#pragma once
#include <hip/hip_runtime.h>

bool c3d_dw_toeplitz_enabled();   // C3D_DW_TZ=1
int c3d_dw333_fwd_toeplitz(const void* x, const float* ss, const float* w, void* y, double* nc, int B, int T, int H, int W,
                           int C, int Cp, hipStream_t s);
