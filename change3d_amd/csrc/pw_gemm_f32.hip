// f32 (parity path) instantiations of the pointwise GEMM: exact f32 FMA chains on v_mfma_f32_16x16x4_f32, generic row
// addressing.  Split from pw_gemm.hip so that the two halves compile in parallel.
#include "pw_gemm_impl.h"

__attribute__((visibility("hidden"))) int c3d_detail_pw_gemm_f32(const c3d_pw_args* args, void* stream) {
  return dispatch_mode<float>(*args, reinterpret_cast<hipStream_t>(stream));
}
