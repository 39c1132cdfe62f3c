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
  // WEIGHT image of the narrow kernel: 8-element k-chunk major, [Kpad / 8][NROWS][8] (NROWS = NT * 16 image rows).  A
  // ds_read_b128 is served in four groups of 16 lanes, each half of one 16-lane row group and half of the next
  // ({0-3, 12-15, 20-27}, ...: /opt/skills/guides/MI355X_MICROARCH.md, LDS table): an A fragment (lane = image row
  // lane % 16, k-chunk lane / 16) of a row-major [n][KL] image makes those halves share bank quads for every KL (8 LDS
  // cycles per read instead of 4); with the chunk index a multiple of 256 B away, the bank quad is the row alone.
  static __host__ __device__ __forceinline__ int widx(int n, int k, int kl, int nrows) { return ((k >> 3) * nrows + n) * 8 + (k & 7); }
  static __device__ __forceinline__ frag_t loadw(const lds_t* base, int n, int ks, int kl, int nrows, int lane) {
    return *reinterpret_cast<const uint4*>(base + ((ks * 4 + (lane >> 4)) * nrows + n) * 8);
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
  static __host__ __device__ __forceinline__ int widx(int n, int k, int kl, int nrows) { return n * kl + k; }   // row major
  static __device__ __forceinline__ frag_t loadw(const lds_t* base, int n, int ks, int kl, int nrows, int lane) {
    return load(base, n, ks, kl, lane);
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
