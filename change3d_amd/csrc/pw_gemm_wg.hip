// Pointwise-convolution data gradient WITH the weight gradient fused in (c3d_pw_args.wg_mode): the bf16 8-wave dense-row
// instantiations of pw_gemm_impl.h's kernel, in their own translation unit (they compile beside pw_gemm.hip).
// Replaces the (c3d_pw_gemm, c3d_pw_wgrad) launch pair for conv_a / conv_c of the res2 / res3 blocks (reference
// model/x3d.py:173-175,214-216: one convolution_backward produces both gradients).
#include <cstring>
#include "pw_gemm_impl.h"

namespace {

template <int EPI, int WG>
int dispatch_wg(const c3d_pw_args& a, hipStream_t s) {
  const int nt = (a.Np + 15) / 16;
  if (nt <= 2) return launch_pw_d<bf16_t, 2, C3D_PRO_AFFINE2, EPI, 8, true, WG>(a, s);
  // N = 48 (conv_a of res3) on THREE output tiles: the first 48 rows of the NT = 4 weight image; 8 KB less LDS (weights,
  // result tile) is what lets this variant hold the second result-tile buffer of the weight-gradient pairs
  if (nt == 3 && WG == C3D_WG_ROWS) return launch_pw_d<bf16_t, 3, C3D_PRO_AFFINE2, EPI, 8, true, WG>(a, s);
  if (nt <= 4) return launch_pw_d<bf16_t, 4, C3D_PRO_AFFINE2, EPI, 8, true, WG>(a, s);
  return launch_pw_d<bf16_t, 7, C3D_PRO_AFFINE2, EPI, 8, true, WG>(a, s);
}

}  // namespace

__attribute__((visibility("hidden"))) int c3d_detail_pw_gemm_wg(const c3d_pw_args* args, void* stream) {
  const c3d_pw_args& a = *args;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype != C3D_DT_BF16 || a.pro_mode != C3D_PRO_AFFINE2 || a.row_mode != C3D_ROWS_DENSE || a.Np > 112) return C3D_E_UNSUPPORTED;
  // no weight gradient, no K limit: the conv_a data gradient of the wider stages (res4: K = 216) with the previous block's
  // c3d_block_out_bwd in its epilogue
  // (instantiated for the 7-tile bucket only: res2 / res3 take C3D_WG_ROWS with the weight gradient fused as well)
  if (a.wg_mode == C3D_WG_MASKSUM) {
    const int nt = (a.Np + 15) / 16;
    if (a.epi_mode != C3D_EPI_ADD || nt <= 4) return C3D_E_UNSUPPORTED;
    return launch_pw_d<bf16_t, 7, C3D_PRO_AFFINE2, C3D_EPI_ADD, 8, true, C3D_WG_MASKSUM>(a, s);
  }
  if (a.Kp > 16 * WG_NTP_MAX) return C3D_E_UNSUPPORTED;
  if (a.wg_mode == C3D_WG_SWISH && a.Kp > 16 * WG_NTP_MAX_SWISH) return C3D_E_UNSUPPORTED;
  if (a.wg_mode == C3D_WG_SWISH && a.epi_mode == C3D_EPI_SWISH_SE_BWD) return dispatch_wg<C3D_EPI_SWISH_SE_BWD, C3D_WG_SWISH>(a, s);
  if (a.wg_mode == C3D_WG_ROWS && a.epi_mode == C3D_EPI_ADD) return dispatch_wg<C3D_EPI_ADD, C3D_WG_ROWS>(a, s);
  return C3D_E_UNSUPPORTED;
}

// Host-side plan check for the stage driver (csrc/stage_driver.hip fuse_wgrad): the same gates and the same LDS plan as the
// launch path above, so that a shape whose f64 accumulator image does not fit beside the weights and the wave regions (e.g.
// Kp = 112 with Np >= 80) keeps its c3d_pw_wgrad launch instead of failing c3d_stage_bwd with C3D_E_UNSUPPORTED.
namespace {
template <int EPI, int WG>
bool plan_wg(const c3d_pw_args& a) {
  PwLaunch L;
  size_t lds = 0;
  const int nt = (a.Np + 15) / 16;
  if (nt <= 2) return plan_pw<bf16_t, 2, C3D_PRO_AFFINE2, EPI, 8, WG>(a, L, lds);
  if (nt == 3 && WG == C3D_WG_ROWS) return plan_pw<bf16_t, 3, C3D_PRO_AFFINE2, EPI, 8, WG>(a, L, lds);
  if (nt <= 4) return plan_pw<bf16_t, 4, C3D_PRO_AFFINE2, EPI, 8, WG>(a, L, lds);
  return plan_pw<bf16_t, 7, C3D_PRO_AFFINE2, EPI, 8, WG>(a, L, lds);
}
}  // namespace

__attribute__((visibility("hidden"))) bool c3d_detail_pw_gemm_wg_supported(int Kp, int Np, int wg_mode) {
  if (Kp <= 0 || Np <= 0 || (Kp & 7) || (Np & 7) || Kp > 16 * WG_NTP_MAX || Np > 112) return false;
  if (wg_mode == C3D_WG_SWISH && Kp > 16 * WG_NTP_MAX_SWISH) return false;
  c3d_pw_args a;
  std::memset(&a, 0, sizeof(a));
  a.M = 1 << 20; a.K = a.Kp = Kp; a.N = a.Np = Np; a.dtype = C3D_DT_BF16; a.pro_mode = C3D_PRO_AFFINE2; a.wg_mode = wg_mode;
  if (wg_mode == C3D_WG_SWISH) { a.epi_mode = C3D_EPI_SWISH_SE_BWD; return plan_wg<C3D_EPI_SWISH_SE_BWD, C3D_WG_SWISH>(a); }
  if (wg_mode == C3D_WG_ROWS) { a.epi_mode = C3D_EPI_ADD; return plan_wg<C3D_EPI_ADD, C3D_WG_ROWS>(a); }
  return false;
}

// C3D_WG_MASKSUM (stage driver: fold c3d_block_out_bwd of the previous block into this conv_a data gradient): same gates
// and LDS plan as the launch path
__attribute__((visibility("hidden"))) bool c3d_detail_pw_gemm_masksum_supported(int Kp, int Np) {
  if (Kp <= 0 || Np <= 0 || (Kp & 7) || (Np & 7) || Kp > 224 || Np > 112 || (Np + 15) / 16 <= 4) return false;
  c3d_pw_args a;
  std::memset(&a, 0, sizeof(a));
  a.M = 1 << 20; a.K = a.Kp = Kp; a.N = a.Np = Np; a.dtype = C3D_DT_BF16; a.pro_mode = C3D_PRO_AFFINE2; a.wg_mode = C3D_WG_MASKSUM;
  a.epi_mode = C3D_EPI_ADD;
  PwLaunch L;
  size_t lds = 0;
  return plan_pw<bf16_t, 7, C3D_PRO_AFFINE2, C3D_EPI_ADD, 8, C3D_WG_MASKSUM>(a, L, lds);
}

#ifdef C3D_PW_CLOCK
extern "C" int c3d_debug_pw_wg_clock(unsigned long long* out, int reset) {   // out[CLK_WAVES][16]: the fused variants' phase clocks
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
