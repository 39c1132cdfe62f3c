// Shared pieces of the pointwise-convolution kernels (pw_gemm.hip, pw_wgrad.hip).
#pragma once
#include "common.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16_t lds_t;
  static constexpr int KSTEP = 32;
  static constexpr int KPAD = 8;
  typedef uint4 frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int row, int ks, int kl, int lane) {
    return *reinterpret_cast<const uint4*>(base + row * kl + ks * 32 + (lane >> 4) * 8);
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<bf16_t>::store(p, f); }
  static __device__ __forceinline__ lds_t cvt(float f) { return f32_to_bf16(f); }
};
template <> struct Mma<float> {
  typedef float lds_t;
  static constexpr int KSTEP = 4;
  static constexpr int KPAD = 4;
  typedef float frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int row, int ks, int kl, int lane) {
    return base[row * kl + ks * 4 + (lane >> 4)];
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<float>::store(p, f); }
  static __device__ __forceinline__ lds_t cvt(float f) { return f; }
};


inline int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace
