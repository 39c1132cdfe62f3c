// Shared device helpers for the Change3D gfx950 kernels.
//
// Layout convention for every activation tensor handled here: channels-last,
// [B][T][H][W][Cp] with Cp = round_up(C, 8) so that each pixel's channel row is a whole
// number of 8-element vectors (16 B in bf16, 32 B in f32).  Pad channels hold zeros.
// Storage type T is float (parity path) or bf16 (throughput path); all arithmetic is f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define C3D_WAVE 64

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

enum { C3D_F32 = 0, C3D_BF16 = 1 };

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round-to-nearest-even in hardware (gfx950 v_cvt_pk_bf16_f32: one instruction per
// PAIR; the integer-arithmetic rounding it replaces cost ~12 VALU per pair in every epilogue)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// ---- 8-element vector load / store (global or LDS), converting to/from f32 ---------------
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
  // raw form: the load is issued now, converted where it is consumed (register prefetch across a loop iteration)
  struct raw_t { float4 a, b; };
  static __device__ __forceinline__ raw_t load_raw(const float* p) {
    raw_t r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r;
  }
  static __device__ __forceinline__ void cvt_raw(const raw_t& r, float (&f)[8]) {
    f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w; f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w;
  }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = v;
  }
  typedef uint4 raw_t;
  static __device__ __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt_raw(const raw_t& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};

// Round a value the way it will be stored (so statistics describe what consumers read).
template <typename T> __device__ __forceinline__ float round_as(float f);
template <> __device__ __forceinline__ float round_as<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float f) { return bf16_to_f32(f32_to_bf16(f)); }

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float f);
template <> __device__ __forceinline__ void st1<float>(float* p, float f) { *p = f; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float f) { *p = f32_to_bf16(f); }

// sigmoid: accurate expf on the f32 (parity) path, hardware v_exp on the bf16 (throughput) path
template <typename T> __device__ __forceinline__ float sigmoid_t(float x);
template <> __device__ __forceinline__ float sigmoid_t<float>(float x) { return 1.0f / (1.0f + expf(-x)); }
template <> __device__ __forceinline__ float sigmoid_t<bf16_t>(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));  // v_exp + v_rcp (1 ulp): ample for bf16 storage
}

// ---- wave / block reductions -----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

// Host-side helpers -----------------------------------------------------------------------
// Tuning knobs (walk lengths, grid caps, experiment switches): the environment is consulted only in the instrumented
// build (hipcc -DC3D_TUNING: `python __graft_entry__.py --tuning` -> lib/libchange3d_hip_tune.so, loaded through
// C3D_LIB).  The product library compiles every knob to its measured-best default and never reads the environment;
// what a caller may legitimately choose at run time is an explicit C-ABI option (c3d_set_option, c3d_stage_desc.flags).
#include <cstdlib>
inline const char* c3d_env(const char* name) {
#ifdef C3D_TUNING
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
inline int c3d_knob(const char* name, int dflt) {
  const char* s = c3d_env(name);
  return s ? atoi(s) : dflt;
}
#define C3D_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return (int)e_;                    \
  } while (0)
