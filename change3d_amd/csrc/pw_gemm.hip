// Pointwise-convolution row GEMM: C entry point and the bf16 (throughput) instantiations.  The kernel lives in
// pw_gemm_impl.h; the f32 (parity) instantiations are compiled in pw_gemm_f32.hip.
#include <cstring>
#include "pw_gemm_impl.h"
#include "launch_hints.h"

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
__attribute__((visibility("hidden"))) int c3d_detail_pw_gemm_wg(const c3d_pw_args* args, void* stream);   // pw_gemm_wg.hip
__attribute__((visibility("hidden"))) int c3d_detail_pw_cfwd(const c3d_pw_args* args, void* stream);      // pw_cfwd.hip

// ------------------------------------------------------------------------------------------ weight images
namespace {

// (NT, KL) of the narrow kernel for a padded shape: dispatch_nt's buckets and pw_gemm_kernel's KL
bool pw_img_geom(int Np, int Kp, int dtype, int& NT, int& KL, int& esz) {
  if (Kp <= 0 || Np <= 0 || Kp > 224 || Np > 224 || (Kp & 7) || (Np & 7)) return false;
  const int nt = (Np + 15) / 16;
  NT = nt <= 2 ? 2 : nt <= 4 ? 4 : nt <= 7 ? 7 : 14;
  if (dtype == C3D_DT_BF16) {
    KL = (Kp + Mma<bf16_t>::KSTEP - 1) / Mma<bf16_t>::KSTEP * Mma<bf16_t>::KSTEP + Mma<bf16_t>::KPAD; esz = 2;
  } else if (dtype == C3D_DT_F32) {
    KL = (Kp + Mma<float>::KSTEP - 1) / Mma<float>::KSTEP * Mma<float>::KSTEP + Mma<float>::KPAD; esz = 4;
  } else {
    return false;
  }
  return true;
}

struct PackOne { const float* w; void* img; int N, K, sn, sk, NT, KL; };
struct PackArgs { PackOne d[C3D_PW_PACK_MAX]; };

// one thread = 4 consecutive k of one image row (8-byte store bf16, 16-byte store f32); blockIdx.y = image
template <typename T>
__global__ __launch_bounds__(256) void pw_pack_kernel(const PackArgs P) {
  typedef Mma<T> MM;
  const PackOne& d = P.d[blockIdx.y];
  const int kq = d.KL >> 2, total = d.NT * 16 * kq;
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < total; i += (int)(gridDim.x * blockDim.x)) {
    const int n = i / kq, k0 = (i - n * kq) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      v[e] = (n < d.N && k < d.K) ? d.w[(size_t)n * d.sn + (size_t)k * d.sk] : 0.f;
    }
    typename MM::lds_t* dst = reinterpret_cast<typename MM::lds_t*>(d.img) + MM::widx(n, k0, d.KL, d.NT * 16);   // (k0 % 4 == 0: one chunk)
    if (sizeof(typename MM::lds_t) == 2)
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

extern "C" int64_t c3d_pw_weight_image_bytes(int32_t Np, int32_t Kp, int32_t dtype) {
  int NT, KL, esz;
  if (!pw_img_geom(Np, Kp, dtype, NT, KL, esz)) return 0;
  return (int64_t)NT * 16 * KL * esz;
}

extern "C" int c3d_pw_pack_weights(const c3d_pw_pack_desc* descs, int32_t n, int32_t dtype, void* stream) {
  if (n < 0 || (n > 0 && !descs)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n; base += C3D_PW_PACK_MAX) {
    const int m = n - base < C3D_PW_PACK_MAX ? n - base : C3D_PW_PACK_MAX;
    PackArgs P;
    std::memset(&P, 0, sizeof(P));
    int most = 0;
    for (int i = 0; i < m; ++i) {
      const c3d_pw_pack_desc& d = descs[base + i];
      int NT, KL, esz;
      if (!d.w || !d.img || ((uintptr_t)d.img & 15) || d.N <= 0 || d.K <= 0 || d.N > d.Np || d.K > d.Kp) return C3D_E_BADARG;
      if (!pw_img_geom(d.Np, d.Kp, dtype, NT, KL, esz)) return C3D_E_UNSUPPORTED;
      P.d[i] = PackOne{d.w, d.img, d.N, d.K, d.w_sn, d.w_sk, NT, KL};
      const int groups = NT * 16 * (KL >> 2);
      if (groups > most) most = groups;
    }
    const dim3 grid((unsigned)((most + 255) / 256), (unsigned)m);
    if (dtype == C3D_DT_BF16) pw_pack_kernel<bf16_t><<<grid, 256, 0, s>>>(P);
    else pw_pack_kernel<float><<<grid, 256, 0, s>>>(P);
    C3D_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int64_t c3d_pw_gemm_wg_ws_floats(int32_t K, int32_t N) { return (int64_t)PW_WG_MAX_PARTS * K * N; }

extern "C" int c3d_pw_gemm(const c3d_pw_args* args, void* stream) {
  if (!args || !args->x || !args->y || !args->w) return C3D_E_BADARG;
  const c3d_pw_args& a = *args;
  if (a.M <= 0 || (a.Kp & 7) || (a.Np & 7) || a.K > a.Kp || a.N > a.Np) return C3D_E_BADARG;
  const bool wide = a.Kp > 224 || a.Np > 224 || a.bias != nullptr;
  // (the block-tiled kernel consumes BatchNorm-BACKWARD sums only: forward statistics tickets stay with the narrow kernel)
  if (wide && (a.Kp > 1024 || a.Np > 1024 || a.fin.ticket || a.pro_out ||
               (a.fin.sums && (a.pro_mode != C3D_PRO_AFFINE2 || a.fin.training))))
    return C3D_E_UNSUPPORTED;
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
  if (a.w_img && ((uintptr_t)a.w_img & 15)) return C3D_E_BADARG;
  if (a.wg_mode != C3D_WG_NONE) {
    if (a.wg_mode != C3D_WG_SWISH && a.wg_mode != C3D_WG_ROWS && a.wg_mode != C3D_WG_MASKSUM) return C3D_E_BADARG;
    if (a.wg_mode == C3D_WG_MASKSUM) {   // no weight gradient: the previous block's c3d_block_out_bwd in this epilogue
      if (!a.wg_x3 || !a.add_sums || a.epi_mode != C3D_EPI_ADD || a.res_mode != 0) return C3D_E_BADARG;
    } else if (!a.wg_dw || !a.wg_ws || (a.wg_mode == C3D_WG_ROWS && !a.wg_x3)) return C3D_E_BADARG;
    if (wide || a.dtype != C3D_DT_BF16) return C3D_E_UNSUPPORTED;
  }
  if (a.add_sums) {
    if (!a.add_c || !a.add_mr || ((uintptr_t)a.add_mr & 15)) return C3D_E_BADARG;
    if (a.wg_mode != C3D_WG_MASKSUM && !(a.wg_mode == C3D_WG_ROWS && a.wg_mask_out && a.epi_mode == C3D_EPI_ADD && a.res_mode == 0))
      return C3D_E_BADARG;   // the sums are over the masked output: only where the mask is applied
  }
  if (a.se_w1) {
    if (a.pro_mode != C3D_PRO_BN_SE_SWISH || !a.fin.sums || a.fin.batch <= 0 || !a.pro_gate || !a.se_b1 || !a.se_w2 || !a.se_b2 ||
        !a.se_hid || a.se_cr <= 0 || a.rows_per_sample <= 0 || !a.fin.ss)
      return C3D_E_BADARG;
    if (wide) return C3D_E_UNSUPPORTED;
  }
  if (wide) return c3d_detail_pw_gemm_wide(args, stream);
  // the wave-private-tile kernels address rows with 32-bit byte offsets into bounds-checked buffer resources (offset 2^31 =
  // "nowhere"): every tensor of the call must stay under 2 GiB
  if ((int64_t)a.M * (a.Kp > a.Np ? a.Kp : a.Np) * (a.dtype == C3D_DT_F32 ? 4 : 2) >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  if (a.wg_mode == C3D_WG_ROWS && (c3d_option_pw_cdg & 1)) {
    // conv_a data gradient + weight gradient: the workgroup-cooperative kernel (csrc/pw_cdgrad.hip) where it applies
    const int rcd = c3d_detail_pw_cdg_a(args, stream);
    if (rcd != C3D_E_UNSUPPORTED) return rcd;
  }
  if (a.wg_mode == C3D_WG_SWISH && (c3d_option_pw_cdg & 2)) {
    const int rcd = c3d_detail_pw_cdg_c(args, stream);   // conv_c data gradient + weight gradient, cooperative (csrc/pw_cdgrad.hip)
    if (rcd != C3D_E_UNSUPPORTED) return rcd;
  }
  if (a.wg_mode != C3D_WG_NONE) return c3d_detail_pw_gemm_wg(args, stream);
  if (a.epi_mode == C3D_EPI_STATS && ((a.pro_mode == C3D_PRO_BN_SE_SWISH && (c3d_option_pw_cfwd & 1)) ||
                                      (a.pro_mode == C3D_PRO_AFFINE2 && a.pro_out && (c3d_option_pw_cfwd & 2)))) {
    // conv_c / conv_a forward of the training path: the workgroup-cooperative kernel (csrc/pw_cfwd.hip) where it applies
    const int rcf = c3d_detail_pw_cfwd(args, stream);
    if (rcf != C3D_E_UNSUPPORTED) return rcf;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = C3D_E_BADARG;
  if (a.dtype == C3D_DT_F32) rc = c3d_detail_pw_gemm_f32(args, stream);
  else if (a.dtype == C3D_DT_BF16) rc = dispatch_mode<bf16_t>(a, s);
  if (rc == PW_E_SE_FALLBACK) {
    // a workgroup would span more samples than the in-kernel SE gate holds (tiny inputs): the separate finalize launch, then
    // the same GEMM reading scale / shift / gate from memory
    rc = c3d_bn_se_finalize(a.fin.sums, a.fin.batch, (double)a.rows_per_sample, a.fin.gamma, a.fin.beta, a.fin.running_mean,
                            a.fin.running_var, a.fin.training ? a.fin.nbt : nullptr, a.fin.momentum, a.fin.eps, a.K, a.Kp,
                            a.fin.training ? 1 : 0, a.se_w1, a.se_b1, a.se_w2, a.se_b2, a.se_cr, a.fin.ss, a.fin.mr,
                            const_cast<float*>(a.pro_gate), a.se_hid, stream);
    if (rc != 0) return rc;
    c3d_pw_args b = a;
    std::memset(&b.fin, 0, sizeof(b.fin));
    b.se_w1 = nullptr;
    b.pro_p = a.fin.ss;
    return c3d_pw_gemm(&b, stream);
  }
  // shapes the wave-private-tile kernel cannot hold in LDS (f32 storage with K*N near 224 x 224): block-tiled kernel
  if (rc == C3D_E_UNSUPPORTED && !a.fin.ticket && a.wg_mode == C3D_WG_NONE &&
      (!a.fin.sums || (a.pro_mode == C3D_PRO_AFFINE2 && !a.fin.training)))
    rc = c3d_detail_pw_gemm_wide(args, stream);
  return rc;
}

