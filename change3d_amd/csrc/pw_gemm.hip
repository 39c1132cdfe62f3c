// Pointwise-convolution row GEMM: C entry point and the bf16 (throughput) instantiations.  The kernel lives in
// pw_gemm_impl.h; the f32 (parity) instantiations are compiled in pw_gemm_f32.hip.
#include "pw_gemm_impl.h"

// f32 instantiations (pw_gemm_f32.hip); library-internal
__attribute__((visibility("hidden"))) int c3d_detail_pw_gemm_f32(const c3d_pw_args* args, void* stream);

extern "C" int c3d_device_cus(void) { return device_cus(); }

#ifdef C3D_PW_CLOCK
extern "C" int c3d_debug_pw_clock(unsigned long long* out, int reset) {   // out[CLK_WAVES][16]
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_pw_clk), sizeof(unsigned long long) * CLK_WAVES * 16);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(c3d_pw_clk));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * CLK_WAVES * 16);
  }
  return (int)e;
}
#endif

int c3d_detail_pw_gemm_wide(const c3d_pw_args* args, void* stream);   // pw_wide.hip

extern "C" int c3d_pw_gemm(const c3d_pw_args* args, void* stream) {
  if (!args || !args->x || !args->y || !args->w) return C3D_E_BADARG;
  const c3d_pw_args& a = *args;
  if (a.M <= 0 || (a.Kp & 7) || (a.Np & 7) || a.K > a.Kp || a.N > a.Np) return C3D_E_BADARG;
  const bool wide = a.Kp > 224 || a.Np > 224 || a.bias != nullptr;
  if (wide && (a.Kp > 1024 || a.Np > 1024 || a.fin.ticket || a.fin.sums || a.pro_out)) return C3D_E_UNSUPPORTED;
  if (a.pro_out && (a.pro_mode != C3D_PRO_AFFINE2 || a.row_mode != C3D_ROWS_DENSE)) return C3D_E_BADARG;
  if (a.pro_mode == C3D_PRO_AFFINE2 && !a.x2) return C3D_E_BADARG;
  if (a.pro_mode != C3D_PRO_NONE && !a.pro_p && !(a.pro_mode == C3D_PRO_AFFINE2 && a.fin.sums)) return C3D_E_BADARG;
  if (a.fin.sums && a.pro_mode == C3D_PRO_AFFINE2 && !a.fin.training && (!a.fin.gamma || !a.fin.mr)) return C3D_E_BADARG;
  if (a.fin.sums && a.pro_mode == C3D_PRO_AFFINE2 && a.fin.training && (!a.fin.gamma || !a.fin.beta || !a.fin.ss)) return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_STATS && !a.stats) return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_SWISH_SE_BWD &&
      (!a.stats || !a.e1 || !a.epi_p || !a.epi_q || a.rows_per_sample <= 0 || (!wide && (a.rows_per_sample & 15))))
    return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_ADD && !a.e1) return C3D_E_BADARG;
  if (a.pro_mode == C3D_PRO_BN_SE_SWISH && a.pro_gate && a.rows_per_sample <= 0) return C3D_E_BADARG;
  if (a.pro_mode == C3D_PRO_BN_SE_SWISH && a.pro_gate && !wide && (a.rows_per_sample & 15)) return C3D_E_BADARG;
  if (a.M >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  if (wide) return c3d_detail_pw_gemm_wide(args, stream);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = C3D_E_BADARG;
  if (a.dtype == C3D_DT_F32) rc = c3d_detail_pw_gemm_f32(args, stream);
  else if (a.dtype == C3D_DT_BF16) rc = dispatch_mode<bf16_t>(a, s);
  // shapes the wave-private-tile kernel cannot hold in LDS (f32 storage with K*N near 224 x 224): block-tiled kernel
  if (rc == C3D_E_UNSUPPORTED && !a.fin.ticket && !a.fin.sums) rc = c3d_detail_pw_gemm_wide(args, stream);
  return rc;
}

