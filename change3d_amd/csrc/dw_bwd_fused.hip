// Depthwise 3x3x3 Conv3d backward, stride 1: DATA gradient and WEIGHT gradient from ONE staged tile
// (conv_b of the X3D bottleneck, reference model/x3d.py:184-193; autograd's convolution_backward computes both
// from the same two operands).
//
// Round 2 ran two kernels per block: c3d_dw333_bwd_data staged db = A*t1 + B[n] + C*b (+halo) and read the `a`
// rows for the ReLU mask; c3d_dw333_wgrad (side stream) read t1, b and a AGAIN (10.2 GB of the 98 GB a B=32 step
// moved, 4.1 ms of kernel time) and spent most of its VALU time re-doing the same conversions.  With the tile
// of db (+halo) in LDS and the centre pixel's relu(bn_a(a)) in registers both gradients are the same LDS reads:
//
//     d a_in[i]  = sum_k db[i + 1 - k] * w[k]            (per dimension, zero padding)
//     d w[k]     = sum_i db[i + 1 - k] * a_in[i]
//
// so the weight gradient costs one more FMA per tap and pixel and no memory traffic at all.
//
// Mapping: workgroup = 8x8 pixels x 32 channels, 512 threads; thread = (pixel, 4 channels, all T frames):
// 27 x 4 weight-gradient partial sums live in registers for the whole tile walk (a thread with 8 channels would
// need 216), reduced once per workgroup through LDS (dump [tap][thread], 64-term sums in fixed order) and added to
// dw with f32 atomics.  The staged tile is double buffered: one barrier per tile.
#include "pw_common.h"   // device_cus()
#include "dw_common.h"
#include "bn_fin.h"
#include "launch_hints.h"
#include "../../include/change3d_hip.h"
#include <cstdlib>
#include <type_traits>

#ifdef C3D_PW_CLOCK
// Debug build only (tools/pw_phase_clock.py --fb): per-phase shader-clock sums, one slot per (workgroup, wave).
constexpr int FCLK_WAVES = 16384;
__device__ unsigned long long c3d_fb_clk[FCLK_WAVES][10];
#define FCLK_DECL unsigned long long fclk_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long fclk_last_ = __builtin_amdgcn_s_memtime();
#define FCLK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fclk_[i] += t_ - fclk_last_; fclk_last_ = t_; }
#define FCLK_FLUSH if ((threadIdx.x & 63) == 0) { const int w_ = (blockIdx.x * 8 + (threadIdx.x >> 6)) % FCLK_WAVES; for (int i_ = 0; i_ < 9; ++i_) c3d_fb_clk[w_][i_] += fclk_[i_]; c3d_fb_clk[w_][9] += 1ull; }
#else
#define FCLK_DECL
#define FCLK(i)
#define FCLK_FLUSH
#endif

namespace {

// C3D_FB_ROWS (build-time experiment, tools/experiments/): tile rows = waves per workgroup.  8 = the shipped geometry (8 x 8
// pixels, 512 threads, one workgroup per CU); 4 = 4 x 8 pixels, 256 threads, two workgroups per CU out of phase.
#ifndef C3D_FB_ROWS
#define C3D_FB_ROWS 8
#endif
constexpr int FB_TH = C3D_FB_ROWS, FB_TW = 8, FB_DH = FB_TH + 2, FB_DW = FB_TW + 2, FB_NTHR = FB_TH * FB_TW * 8;
#if C3D_FB_ROWS == 8
#define FB_KATTR __launch_bounds__(FB_NTHR)
#else
#define FB_KATTR __launch_bounds__(FB_NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))   // 256 registers per wave: two workgroups per CU
#endif

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct Raw8<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};
// 4-channel half vectors: what a thread owns of a pixel
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
  typedef uint2 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
  }
};
template <> struct Raw4<float> {
  typedef float4 type;
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};

__device__ __forceinline__ void lds4(const float* p, float (&f)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
}

__device__ __forceinline__ void lds8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// empty asm that "modifies" the accumulators: their FMAs cannot be sunk below it, later LDS reads not hoisted above
template <int TT> __device__ __forceinline__ void pin_acc(f32x2_t (&a)[TT][2]);
template <> __device__ __forceinline__ void pin_acc<3>(f32x2_t (&a)[3][2]) {
  asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]) : : "memory");
}
template <> __device__ __forceinline__ void pin_acc<5>(f32x2_t (&a)[5][2]) {
  asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
               "+v"(a[3][1]), "+v"(a[4][0]), "+v"(a[4][1]) : : "memory");
}
// (the weight-gradient partial sums of one (ky, kx) step: temporal taps kt = 0, 1, 2)
__device__ __forceinline__ void pin_dw(f32x2_t (&a0)[2], f32x2_t (&a1)[2], f32x2_t (&a2)[2]) {
  asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a1[0]), "+v"(a1[1]), "+v"(a2[0]), "+v"(a2[1]));
}

// Stride 2 (first block of a stage).  The staged db tile is the same 10 x 10 output-resolution tile; it now covers 16 x 16
// INPUT pixels: a thread owns the 2x2 quad (2py + cy, 2px + cx) and handles its four pixels one parity class (cy, cx) after
// the other.  An input pixel receives only the taps of matching parity -- (iy + 1 - ky) must be even -- so the taps of a
// class, and the dW accumulators they update, are compile-time constants: 1 + 2 + 2 + 4 = 9 (ky, kx) steps per quad, as
// many as ONE stride-1 pixel, against one staging pass and one barrier.  (A first version with one 8x8 INPUT tile per pass
// and one class per wave was correct and no faster than the separate kernels: 616 us against 356 + 264 us per block --
// staging, barrier and flush were paid per 64 input pixels.)
template <int TT, int CY, int CX>
__device__ __forceinline__ void fb_taps_s2(const float4* tp, const float* wlp, f32x2_t (&acc)[TT][2], f32x2_t (&dwa)[27][2],
                                           const f32x2_t (&ain)[TT][2]) {
#pragma unroll
  for (int ky = CY ? 0 : 1; ky < 3; ky += 2) {
#pragma unroll
    for (int kx = CX ? 0 : 1; kx < 3; kx += 2) {
      const int lyo = (CY + 1 - ky) / 2 + 1, lxo = (CX + 1 - kx) / 2 + 1;   // db row / column relative to (py, px)
      f32x2_t wk[3][2];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const float4 wv = *reinterpret_cast<const float4*>(wlp + (kt * 9 + ky * 3 + kx) * 32);
        wk[kt][0] = f32x2_t{wv.x, wv.y}; wk[kt][1] = f32x2_t{wv.z, wv.w};
      }
#pragma unroll
      for (int to = 0; to < TT; ++to) {
        const float4 hv = tp[((to * FB_DH + lyo) * FB_DW + lxo) * DW_CV];
        const f32x2_t v0 = {hv.x, hv.y}, v1 = {hv.z, hv.w};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          const int ti = to + kt - 1;
          if (ti >= 0 && ti < TT) {
            const int k = kt * 9 + ky * 3 + kx;
            acc[ti][0] = __builtin_elementwise_fma(v0, wk[kt][0], acc[ti][0]);
            acc[ti][1] = __builtin_elementwise_fma(v1, wk[kt][1], acc[ti][1]);
            dwa[k][0] = __builtin_elementwise_fma(v0, ain[ti][0], dwa[k][0]);
            dwa[k][1] = __builtin_elementwise_fma(v1, ain[ti][1], dwa[k][1]);
          }
        }
      }
      pin_acc<TT>(acc);
      pin_dw(dwa[ky * 3 + kx], dwa[9 + ky * 3 + kx], dwa[18 + ky * 3 + kx]);
      if (CX == 0) break;   // (kx = 1 only)
    }
    if (CY == 0) break;     // (ky = 1 only)
  }
}

// Stride-1 tap walk in plain (compiler-scheduled) form: one fragment slot, nine (ky, kx) steps.  Five frames (SCD) run
// this: the hand-pipelined walk below needs a second fragment slot that does not fit beside 20 + 20 + 108 accumulators.
template <int TT>
__device__ __forceinline__ void fb_taps_s1(const float4* tp, const float* wlp, f32x2_t (&acc)[TT][2], f32x2_t (&dwa)[27][2],
                                           const f32x2_t (&ain)[TT][2]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      f32x2_t wk[3][2];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const float4 wv = *reinterpret_cast<const float4*>(wlp + (kt * 9 + ky * 3 + kx) * 32);
        wk[kt][0] = f32x2_t{wv.x, wv.y}; wk[kt][1] = f32x2_t{wv.z, wv.w};
      }
#pragma unroll
      for (int to = 0; to < TT; ++to) {
        const float4 hv = tp[((to * FB_DH + (2 - ky)) * FB_DW + (2 - kx)) * DW_CV];
        const f32x2_t v0 = {hv.x, hv.y}, v1 = {hv.z, hv.w};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          const int ti = to + kt - 1;
          if (ti >= 0 && ti < TT) {
            const int k = kt * 9 + ky * 3 + kx;
            acc[ti][0] = __builtin_elementwise_fma(v0, wk[kt][0], acc[ti][0]);
            acc[ti][1] = __builtin_elementwise_fma(v1, wk[kt][1], acc[ti][1]);
            dwa[k][0] = __builtin_elementwise_fma(v0, ain[ti][0], dwa[k][0]);
            dwa[k][1] = __builtin_elementwise_fma(v1, ain[ti][1], dwa[k][1]);
          }
        }
      }
      pin_acc<TT>(acc);
      pin_dw(dwa[ky * 3 + kx], dwa[9 + ky * 3 + kx], dwa[18 + ky * 3 + kx]);
    }
  }
}

// ---- end of a workgroup's walk (shared by the register-prefetch and the LDS-DMA ring kernels): BN_a-backward sums (lanes of
// equal (half, vector) inside a wave, then the eight waves) and the weight gradient (dump [tap][thread] per channel-of-four,
// 64-pixel sums in fixed order, f32 atomics).  `scratch` = the (dead) tile buffers: >= 27 * (FB_NTHR + 48) floats.
__device__ __forceinline__ void fb_flush(void* scratch, const float (&S1)[4], const float (&S2)[4], const f32x2_t (&dwa)[27][2],
                                         double* __restrict__ dsums, float* __restrict__ dw, const DwGeom& g, const int c0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  double* red64 = reinterpret_cast<double*>(scratch);   // [8 waves][2 halves x DW_CV][8]
  __syncthreads();
  double D1[4], D2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    D1[j] = (double)S1[j]; D2[j] = (double)S2[j];
#pragma unroll
    for (int o = 2 * DW_CV; o < 64; o <<= 1) {   // the 8 pixels of the wave (lane bits 3-5)
      D1[j] += __shfl_xor(D1[j], o, 64);
      D2[j] += __shfl_xor(D2[j], o, 64);
    }
  }
  if (lane < 2 * DW_CV) {                        // lane = half * 4 + vector
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red64[(wave * 2 * DW_CV + lane) * 8 + j] = D1[j];
      red64[(wave * 2 * DW_CV + lane) * 8 + 4 + j] = D2[j];
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int v = tid >> 4, r = tid & 15;
    const int hh = r >> 3, which = (r >> 2) & 1, j = r & 3;
    double sacc = 0.0;
    for (int wv = 0; wv < FB_NTHR / 64; ++wv) sacc += red64[(wv * 2 * DW_CV + hh * DW_CV + v) * 8 + which * 4 + j];
    const int c = c0 + v * 8 + hh * 4 + j;
    if (c < g.C) atomicAdd(dsums + (size_t)which * g.C + c, sacc);
  }
  // ---- weight gradient
  if (dw == nullptr) return;
  // Bank-padded rows: thread t = [pixel][half][vector] dumps at t + 8 * (pixel half); the 32 lanes of a ds_read_b32 group
  // below differ in (tap bit, pixel half, channel half, vector) = 16 * (LD = 560) + 8 * (264) + 4 + 1 floats mod 32: 32 banks
  // (the plain [tap][thread] rows put them on 4: 16 LDS cycles per read instead of 2, 12 k of the flush's 17 k clocks,
  // tools/lds_bank_model.py).  The 64-pixel sums keep their order.
  constexpr int FB_DUMP_LD = FB_NTHR + 48, FB_DUMP_PT = FB_NTHR / 2 + 8;
  float* dump = reinterpret_cast<float*>(scratch);      // 27 * 560 floats = 60 KB
  const int dslot = tid + (FB_DUMP_PT - FB_NTHR / 2) * (tid / (FB_NTHR / 2));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 27; ++k) dump[k * FB_DUMP_LD + dslot] = dwa[k][j >> 1][j & 1];
    __syncthreads();
    for (int t2 = tid; t2 < 27 * 8 * 2; t2 += FB_NTHR) {   // (one trip with 512 threads)
      const int o = t2 >> 1, part = t2 & 1;
      const int tap = o >> 3, v = o & 3, hh = (o >> 2) & 1;
      const float* src = dump + tap * FB_DUMP_LD + part * FB_DUMP_PT + hh * DW_CV + v;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < FB_NTHR / 16; ++k) s += src[k * 2 * DW_CV];
      s += __shfl_xor(s, 1, 64);
      const int c = c0 + v * 8 + hh * 4 + j;
      if (part == 0 && c < g.C) atomicAdd(dw + (size_t)c * 27 + tap, s);
    }
  }
}

template <typename T, int TT, int S>
__global__ FB_KATTR void dw_bwd_fused_kernel(
    const T* __restrict__ t1, const T* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const float* __restrict__ w,
    const T* __restrict__ a, const float* __restrict__ ss_a, const float* __restrict__ mr_a, T* __restrict__ t2,
    double* __restrict__ dsums, float* __restrict__ dw, const DwGeom g, const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef Raw8<T> R8;
  typedef Raw4<T> R4;
  constexpr int NI = TT * FB_DH * FB_DW * DW_CV;          // staged 8-channel vectors per tile
  constexpr int SL = (NI + FB_NTHR - 1) / FB_NTHR;        // prefetch slots per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);             // [27][32]
  float* cf = wl + 27 * 32;                               // [7][32]: cA, cB(sample), cC, sa, sb, ma, ra
  float4* tile = reinterpret_cast<float4*>(cf + 7 * 32);  // [2 buffers][2 half-vector planes][NI]

  const int tid = threadIdx.x;
  FCLK_DECL
  // thread = [pixel][channel half][channel vector]: the 8 lanes of a pixel own its 32 channels, so the `a` rows they load and
  // the `t2` rows they store are 64 contiguous bytes per pixel (half-vector loads of one channel half per wave -- h = tid >> 8
  // until late round 4 -- were 8-byte pieces 16 B apart: twice the cache-line segments per instruction, and the stride-2
  // classes four times).  LDS: a 16-lane read group covers 4 pixels x 4 vectors of ONE half each -- the two half planes are
  // a multiple of 16 float4 apart, so the bank quad is the (pixel, vector) alone: conflict-free as before.
  const int h = (tid >> 2) & 1;                           // channel half of the vector
  const int cv = tid & (DW_CV - 1);
  const int pix = tid >> 3;
  const int px = pix & (FB_TW - 1), py = pix >> 3;

  const int tiles_x = (g.W + FB_TW * S - 1) / (FB_TW * S), tiles_y = (g.H + FB_TH * S - 1) / (FB_TH * S);   // tile: 8S x 8S input pixels
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int b = co.group / gx, tg = co.group % gx;
  const int c0 = co.chunk * DW_CV * 8;
  const int cb8 = c0 + cv * 8;                            // the 8-channel vector this thread stages
  const int cb4 = cb8 + h * 4;                            // the 4 channels this thread owns
  const bool c_ok = cb8 < g.Cp;                           // (Cp is a multiple of 8: both halves exist or neither)

  for (int i = tid; i < 27 * 32; i += FB_NTHR) {
    const int tap = i >> 5, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  if (fin.sums) {
    // BatchNorm_b backward coefficients of a block without SqueezeExcitation, rebuilt from the per-sample sums of the
    // conv_c data gradient's epilogue (csrc/bn_fin.h: no c3d_se_bn_bwd_coef launch in front of this kernel); the first
    // workgroup of each channel chunk accumulates d gamma / d beta
    if (tid < 128) {
      float cA, cB, cC;
      c3dfin::bn_b_bwd_coef_nc(fin, g.C, g.Cp, c0 + (tid >> 2), tid & 3, co.group == 0, cA, cB, cC);
      if ((tid & 3) == 0) { cf[tid >> 2] = cA; cf[32 + (tid >> 2)] = cB; cf[64 + (tid >> 2)] = cC; }
    }
  }
  for (int i = tid + (fin.sums ? 3 * 32 : 0); i < 7 * 32; i += FB_NTHR) {
    const int k = i >> 5, c = c0 + (i & 31);
    float v = 0.f;
    if (c < g.Cp) {
      v = k == 0 ? coefA[c] : k == 1 ? coefB[(size_t)b * g.Cp + c] : k == 2 ? coefC[c] : k == 3 ? ss_a[c]
        : k == 4 ? ss_a[g.Cp + c] : k == 5 ? mr_a[c] : mr_a[g.Cp + c];
    }
    cf[i] = v;
  }

  // BN_a-backward sums (sum t2, sum t2*ahat): f32 per thread over its walk (tiles_per_wg x T terms), f64 from the
  // cross-thread reduction on (the sums of the two terms nearly cancel on some channels: d gamma needs the digits)
  float S1[4], S2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { S1[j] = 0.f; S2[j] = 0.f; }
  f32x2_t dwa[27][2];   // weight-gradient partial sums of this thread's 4 channels, as register pairs (v_pk_fma_f32)
#pragma unroll
  for (int k = 0; k < 27; ++k) { dwa[k][0] = f32x2_t{0.f, 0.f}; dwa[k][1] = f32x2_t{0.f, 0.f}; }

  // per-slot staging descriptors, computed once (offset relative to the tile origin; (row, column) in the tile)
  typename R8::type r1[SL], r2[SL];
  typename R4::type ar[TT];
  unsigned vmask = 0;
  int rel[SL], yx[SL];
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    const int i_ = tid + sl * FB_NTHR;
    const int p_ = i_ / DW_CV;
    const int ix_ = p_ % FB_DW, q_ = p_ / FB_DW;
    const int iy_ = q_ % FB_DH, t_ = q_ / FB_DH;
    const bool use_ = i_ < NI && c_ok && t_ < g.T;
    rel[sl] = ((t_ * g.Ho + iy_) * g.Wo + ix_) * g.Cp + cb8;
    yx[sl] = use_ ? (iy_ | (ix_ << 16)) : 0x7fff7fff;
  }
  const int orel = (S * py * g.W + S * px) * g.Cp + cb4, ofr = g.H * g.W * g.Cp;   // lane offset in a tile (class (0,0) pixel), frame stride
  // raw (t1, b) rows of the db tile of tile TL -> r1 / r2
#define FB_ISSUE_RAW(TL)                                                                          \
  {                                                                                               \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                         \
    const int dy0_ = ty_ * FB_TH - 1, dx0_ = tx_ * FB_TW - 1;                                     \
    const int64_t tb_ = ((((int64_t)b * g.T) * g.Ho + dy0_) * g.Wo + dx0_) * g.Cp;  /* wave-uniform */ \
    const T* t1b_ = t1 + tb_;                                                                     \
    const T* bbb_ = bb + tb_;                                                                     \
    vmask = 0;                                                                                    \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                           \
      const unsigned gy_ = (unsigned)(dy0_ + (yx[sl] & 0xffff));                                  \
      const unsigned gx_ = (unsigned)(dx0_ + (yx[sl] >> 16));                                     \
      if (gy_ < (unsigned)g.Ho && gx_ < (unsigned)g.Wo) {                                         \
        r1[sl] = R8::load(t1b_ + rel[sl]);                                                        \
        r2[sl] = R8::load(bbb_ + rel[sl]);                                                        \
        vmask |= 1u << sl;                                                                        \
      }                                                                                           \
    }                                                                                             \
  }
  // this thread's `a` rows (all frames) of its class-(CY, CX) pixel in tile TL -> ar
#define FB_ISSUE_A(TL, CY, CX)                                                                    \
  {                                                                                               \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                         \
    if (c_ok && ty_ * FB_TH * S + S * py + (CY) < g.H && tx_ * FB_TW * S + S * px + (CX) < g.W) { \
      const T* ab_ = a + ((((int64_t)b * g.T) * g.H + ty_ * FB_TH * S + (CY)) * g.W + tx_ * FB_TW * S + (CX)) * g.Cp; \
      _Pragma("unroll") for (int t = 0; t < TT; ++t)                                              \
        if (t < g.T) ar[t] = R4::load(ab_ + (orel + t * ofr));                                    \
    }                                                                                             \
  }
#define FB_ISSUE(TL) { FB_ISSUE_RAW(TL) FB_ISSUE_A(TL, 0, 0) }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) FB_ISSUE(tl0)
  __syncthreads();   // wl / cf staged
  FCLK(0)

  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int y0 = ty * FB_TH * S, x0 = tx * FB_TW * S;   // input-resolution origin of the tile
    float4* tb = tile + (size_t)((tl - tl0) & 1) * 2 * NI;
    // ---- stage db = A*t1 + B[n] + C*b (zero outside the image) as two f32 half-vector planes
    {
      float cA[8], cB[8], cC[8];
      lds8(cf + 0 * 32 + cv * 8, cA);
      lds8(cf + 1 * 32 + cv * 8, cB);
      lds8(cf + 2 * 32 + cv * 8, cC);
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int i = tid + sl * FB_NTHR;
        if (i < NI) {
          float f[8];
          if ((vmask >> sl) & 1u) {
            float f2[8];
            R8::cvt(r1[sl], f);
            R8::cvt(r2[sl], f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
          tb[i] = make_float4(f[0], f[1], f[2], f[3]);            // i = p * DW_CV + cv
          tb[NI + i] = make_float4(f[4], f[5], f[6], f[7]);
        }
      }
    }
    FCLK(1)
    // ---- one pixel of this thread (class (CY, CX) of its quad; stride 1: the only one): a_in = relu(bn_a(a)) for the
    // weight gradient and the mask
    typename R4::type arc[TT];
    f32x2_t ain[TT][2];
    f32x2_t acc[TT][2];
    bool p_ok;
    float sa[4], sb[4];
    lds4(cf + 3 * 32 + cv * 8 + h * 4, sa);
    lds4(cf + 4 * 32 + cv * 8 + h * 4, sb);
#define FB_PRE(CY, CX)                                                                                            \
  {                                                                                                               \
    p_ok = c_ok && y0 + S * py + (CY) < g.H && x0 + S * px + (CX) < g.W;                                          \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                             \
      if (p_ok && t < g.T) {                                                                                      \
        arc[t] = ar[t];                                                                                           \
        float av[4];                                                                                              \
        R4::cvt(arc[t], av);                                                                                      \
        ain[t][0] = f32x2_t{fmaxf(fmaf(av[0], sa[0], sb[0]), 0.f), fmaxf(fmaf(av[1], sa[1], sb[1]), 0.f)};        \
        ain[t][1] = f32x2_t{fmaxf(fmaf(av[2], sa[2], sb[2]), 0.f), fmaxf(fmaf(av[3], sa[3], sb[3]), 0.f)};        \
      } else {                                                                                                    \
        ain[t][0] = f32x2_t{0.f, 0.f}; ain[t][1] = f32x2_t{0.f, 0.f};                                             \
      }                                                                                                           \
      acc[t][0] = f32x2_t{0.f, 0.f}; acc[t][1] = f32x2_t{0.f, 0.f};                                               \
    }                                                                                                             \
  }
    // ---- mask, store t2, BN_a-backward sums
#define FB_EPI(CY, CX)                                                                                            \
  if (p_ok) {                                                                                                     \
    float ma[4], ra[4];                                                                                           \
    lds4(cf + 5 * 32 + cv * 8 + h * 4, ma);                                                                       \
    lds4(cf + 6 * 32 + cv * 8 + h * 4, ra);                                                                       \
    T* ob = t2 + ((((int64_t)b * g.T) * g.H + y0 + (CY)) * g.W + x0 + (CX)) * g.Cp;                               \
    _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                             \
      if (t < g.T) {                                                                                              \
        float av[4], o[4];                                                                                        \
        R4::cvt(arc[t], av);                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
          const float d = round_as<T>(ain[t][j >> 1][j & 1] > 0.f ? acc[t][j >> 1][j & 1] : 0.f);                 \
          o[j] = d;                                                                                               \
          S1[j] += d; S2[j] += d * ((av[j] - ma[j]) * ra[j]);                                                     \
        }                                                                                                         \
        R4::store(ob + (orel + t * ofr), o);                                                                      \
      }                                                                                                           \
    }                                                                                                             \
  }
    FB_PRE(0, 0)
    if constexpr (S == 2) {
      // the next class's `a` rows go out before the next tile's raw rows: vmcnt retires in order, and class (1,0)
      // is needed three tap steps from now, the raw rows a whole tile from now
      FB_ISSUE_A(tl, 1, 0)
      if (tl + 1 < tl1) FB_ISSUE_RAW(tl + 1)
      FCLK(2)
      __syncthreads();
      FCLK(3)
      const float4* tp = tb + (size_t)h * NI + (py * FB_DW + px) * DW_CV + cv;
      const float* wlp = wl + cv * 8 + h * 4;
      fb_taps_s2<TT, 0, 0>(tp, wlp, acc, dwa, ain);
      FCLK(4)
      FB_EPI(0, 0)
      FCLK(5)
      FB_PRE(1, 0)
      FB_ISSUE_A(tl, 0, 1)
      FCLK(2)
      fb_taps_s2<TT, 1, 0>(tp, wlp, acc, dwa, ain);
      FCLK(4)
      FB_EPI(1, 0)
      FCLK(5)
      FB_PRE(0, 1)
      FB_ISSUE_A(tl, 1, 1)
      FCLK(2)
      fb_taps_s2<TT, 0, 1>(tp, wlp, acc, dwa, ain);
      FCLK(4)
      FB_EPI(0, 1)
      FCLK(5)
      FB_PRE(1, 1)
      if (tl + 1 < tl1) FB_ISSUE_A(tl + 1, 0, 0)
      FCLK(2)
      fb_taps_s2<TT, 1, 1>(tp, wlp, acc, dwa, ain);
      FCLK(4)
      FB_EPI(1, 1)
      FCLK(5)
      continue;
    }
    // (Issuing the next tile's loads one per tap step instead of in a bunch here was measured in round 3: the stalled
    // issue time just moves into the tap walk, +8 % per launch -- the stall is the memory system's back-pressure at
    // ~10 B/clk/CU, not a queue that drains while the address path idles; profiles/r03_dw_bwd_fused_phase_clock.txt.)
    if (tl + 1 < tl1) FB_ISSUE(tl + 1)
    FCLK(2)
    __syncthreads();
    FCLK(3)

    // ---- 27 taps: one LDS read feeds the data gradient (x weight) and the weight gradient (x a_in)
    // lowest-address tap (ky = kx = 2) as base: every other tap is a non-negative immediate offset of the ds_read
    const float4* tp = tb + (size_t)h * NI + (py * FB_DW + px) * DW_CV + cv;
    // software pipeline over the nine (ky, kx) steps: the LDS reads of step s+1 are issued before the FMAs of step s
    // (with two waves per SIMD the ~130-clock LDS latency of a step was exposed nine times per tile)
    float4 wq[2][3], hq[2][TT];
#define FB_LOAD(S, SLOT)                                                                                          \
  {                                                                                                               \
    constexpr int ky_ = (S) / 3, kx_ = (S) % 3;                                                                   \
    _Pragma("unroll") for (int kt = 0; kt < 3; ++kt)                                                             \
      wq[SLOT][kt] = *reinterpret_cast<const float4*>(wl + (kt * 9 + (S)) * 32 + cv * 8 + h * 4);                \
    _Pragma("unroll") for (int to = 0; to < TT; ++to)                                                            \
      hq[SLOT][to] = tp[((to * FB_DH + (2 - ky_)) * FB_DW + (2 - kx_)) * DW_CV];                                 \
  }
#define FB_STEP(S, SLOT)                                                                                          \
  {                                                                                                               \
    _Pragma("unroll") for (int to = 0; to < TT; ++to) {                                                          \
      const f32x2_t v0 = {hq[SLOT][to].x, hq[SLOT][to].y}, v1 = {hq[SLOT][to].z, hq[SLOT][to].w};                 \
      _Pragma("unroll") for (int kt = 0; kt < 3; ++kt) {                                                         \
        const int ti = to + kt - 1;   /* in[ti] <-> out[to] through temporal tap kt */                            \
        if (ti >= 0 && ti < TT) {                                                                                 \
          const f32x2_t w0 = {wq[SLOT][kt].x, wq[SLOT][kt].y}, w1 = {wq[SLOT][kt].z, wq[SLOT][kt].w};             \
          acc[ti][0] = __builtin_elementwise_fma(v0, w0, acc[ti][0]);                                             \
          acc[ti][1] = __builtin_elementwise_fma(v1, w1, acc[ti][1]);                                             \
          dwa[kt * 9 + (S)][0] = __builtin_elementwise_fma(v0, ain[ti][0], dwa[kt * 9 + (S)][0]);                 \
          dwa[kt * 9 + (S)][1] = __builtin_elementwise_fma(v1, ain[ti][1], dwa[kt * 9 + (S)][1]);                 \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    /* pin the schedule: the data-gradient accumulators are only consumed under `p_ok` below and the weight-     \
       gradient sums at the end of the walk, so the compiler sinks their FMAs and keeps the LDS reads of ALL     \
       nine steps live until then (216 registers, spills) */                                                      \
    pin_acc<TT>(acc);                                                                                             \
    pin_dw(dwa[(S)], dwa[9 + (S)], dwa[18 + (S)]);                                                                \
  }
    if constexpr (TT <= 3) {
      FB_LOAD(0, 0)
      FB_LOAD(1, 1) FB_STEP(0, 0)
      FB_LOAD(2, 0) FB_STEP(1, 1)
      FB_LOAD(3, 1) FB_STEP(2, 0)
      FB_LOAD(4, 0) FB_STEP(3, 1)
      FB_LOAD(5, 1) FB_STEP(4, 0)
      FB_LOAD(6, 0) FB_STEP(5, 1)
      FB_LOAD(7, 1) FB_STEP(6, 0)
      FB_LOAD(8, 0) FB_STEP(7, 1)
      FB_STEP(8, 0)
    } else {
      fb_taps_s1<TT>(tp, wl + cv * 8 + h * 4, acc, dwa, ain);   // five frames (SCD)
    }
#undef FB_LOAD
#undef FB_STEP
    FCLK(4)
    FB_EPI(0, 0)
    FCLK(5)
  }
#undef FB_PRE
#undef FB_EPI
#undef FB_ISSUE
#undef FB_ISSUE_RAW
#undef FB_ISSUE_A

  fb_flush(tile, S1, S2, dwa, dsums, dw, g, c0);
  FCLK(7)
  FCLK_FLUSH
}

// =====================================================================================================================
// LDS-DMA ring variant (bf16 storage, stride 1, T <= 3): the raw t1 / b rows of the db tile and the workgroup's `a` rows
// arrive by global_load_lds_dwordx4 (1 KiB per wave instruction, no destination registers) TWO tiles ahead of the tap walk,
// the requests carried across the tile barrier.  The register-prefetch kernel above holds ONE tile ahead in 24 + 6 VGPRs,
// issues it in a burst and stalls 30 % of a wave's time in that burst; its exec-masked global loads also make the
// compiler fall back to vmcnt(0) around them.  Here:
//   * a ring slot = [t1 plane][b plane][a rows]: a raw bf16 8-channel vector of t1 plus one of b are 32 bytes, exactly the
//     two f32 half-vector planes of db = A t1 + B[n] + C b -- the staging pass converts a slot IN PLACE (thread i reads
//     plane0[i], plane1[i] and writes db[0..3] -> plane0[i], db[4..7] -> plane1[i]); three slots = 150 KB;
//   * every wave converts exactly the pieces it requested (lane-linear DMA image = the staging index i = tid + 512 sl) and
//     reads only its own pixels' `a` rows: a wave's counted vmcnt is the only ordering the landed data needs; the tile
//     barrier (conversions visible to the tap walk) stays the one barrier per tile;
//   * the DMA is issued from inline asm, invisible to the compiler's s_waitcnt bookkeeping (a compiler-visible LDS-DMA makes
//     every later LDS access of the wave wait for vmcnt(0)); the waits on it are explicit and counted;
//   * lanes without a source (halo outside the image, channel padding, the tail of the last piece) request a safe address
//     instead of being masked off: every DMA instruction is always issued, so the counts are exact.
// Arithmetic, summation order, tap walk and flush are the register-prefetch kernel's: bit-identical t2 / dw / sums.
__device__ __forceinline__ void fb_glds16(const void* gsrc, const uint32_t lds_dst) {   // lds_dst: wave-uniform byte address
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void fb_wait_vm(const int n) {   // wave-uniform n: s_waitcnt vmcnt(n)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
typedef __attribute__((address_space(3))) unsigned char* fb_lds_ptr_t;

template <int TT>
struct RingPlan {
  static constexpr int NI = TT * FB_DH * FB_DW * DW_CV;       // staged 8-channel vectors per plane
  static constexpr int NP = (NI + 63) / 64;                   // 1 KiB pieces per plane
  static constexpr int NPL = NP * 64;                         // plane stride (vectors)
  static constexpr int SL = (NP + 7) / 8;                     // pieces per wave and plane (at most)
  // three frames: the `a` rows ride in the slot too and the ring is three slots deep (two tiles ahead); five frames (SCD):
  // a slot is 64 KB, two fit -- one tile ahead, like the register prefetch, but in no registers -- and the `a` rows stay
  // register loads
  static constexpr bool A_LDS = TT <= 3;
  static constexpr int NA = A_LDS ? (TT * 8 * DW_CV + 63) / 64 : 0;   // `a` pieces per wave: TT frames x 8 pixels x 4 vectors
  static constexpr int A_WAVE_BYTES = TT * 8 * DW_CV * 16;
  static constexpr int A_BYTES = A_LDS ? 8 * A_WAVE_BYTES : 0;
  static constexpr int SLOT_BYTES = 2 * NPL * 16 + A_BYTES;
  static constexpr int R = A_LDS ? 3 : 2;
  static constexpr int AHEAD = R - 1;
  static constexpr int HEAD_BYTES = (27 * 32 + 7 * 32) * 4;
  static constexpr int LDS_BYTES = HEAD_BYTES + R * SLOT_BYTES;
};

// SPREAD: the requests of tile + AHEAD are issued one per (ky, kx) step of the tap walk instead of in a burst after the barrier
template <int TT, bool SPREAD>
__global__ __launch_bounds__(FB_NTHR) void dw_bwd_ring_kernel(
    const bf16_t* __restrict__ t1, const bf16_t* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const float* __restrict__ w,
    const bf16_t* __restrict__ a, const float* __restrict__ ss_a, const float* __restrict__ mr_a, bf16_t* __restrict__ t2,
    double* __restrict__ dsums, float* __restrict__ dw, const DwGeom g, const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef bf16_t T;
  typedef Raw4<T> R4;
  typedef RingPlan<TT> P;
  constexpr int NI = P::NI, NP = P::NP, NPL = P::NPL, SL = P::SL, NA = P::NA, AHEAD = P::AHEAD;
  constexpr bool A_LDS = P::A_LDS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);             // [27][32]
  float* cf = wl + 27 * 32;                               // [7][32]: cA, cB(sample), cC, sa, sb, ma, ra
  unsigned char* ring = smem + P::HEAD_BYTES;             // [R slots]{[2 planes][NPL] 16 B, [8 waves][TT][8 pixels][4 vectors] 16 B}
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(fb_lds_ptr_t)ring;

  const int tid = threadIdx.x;
  FCLK_DECL
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = (tid >> 2) & 1;                           // thread = [pixel][channel half][channel vector], as above
  const int cv = tid & (DW_CV - 1);
  const int pix = tid >> 3;
  const int px = pix & (FB_TW - 1), py = pix >> 3;        // py == wave

  const int tiles_x = (g.W + FB_TW - 1) / FB_TW, tiles_y = (g.H + FB_TH - 1) / FB_TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int b = co.group / gx, tg = co.group % gx;
  const int c0 = co.chunk * DW_CV * 8;
  const int cb8 = c0 + cv * 8;
  const int cb4 = cb8 + h * 4;
  const bool c_ok = cb8 < g.Cp;

  for (int i = tid; i < 27 * 32; i += FB_NTHR) {
    const int tap = i >> 5, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  if (fin.sums) {
    if (tid < 128) {
      float cA, cB, cC;
      c3dfin::bn_b_bwd_coef_nc(fin, g.C, g.Cp, c0 + (tid >> 2), tid & 3, co.group == 0, cA, cB, cC);
      if ((tid & 3) == 0) { cf[tid >> 2] = cA; cf[32 + (tid >> 2)] = cB; cf[64 + (tid >> 2)] = cC; }
    }
  }
  for (int i = tid + (fin.sums ? 3 * 32 : 0); i < 7 * 32; i += FB_NTHR) {
    const int k = i >> 5, c = c0 + (i & 31);
    float v = 0.f;
    if (c < g.Cp) {
      v = k == 0 ? coefA[c] : k == 1 ? coefB[(size_t)b * g.Cp + c] : k == 2 ? coefC[c] : k == 3 ? ss_a[c]
        : k == 4 ? ss_a[g.Cp + c] : k == 5 ? mr_a[c] : mr_a[g.Cp + c];
    }
    cf[i] = v;
  }

  float S1[4], S2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { S1[j] = 0.f; S2[j] = 0.f; }
  f32x2_t dwa[27][2];
#pragma unroll
  for (int k = 0; k < 27; ++k) { dwa[k][0] = f32x2_t{0.f, 0.f}; dwa[k][1] = f32x2_t{0.f, 0.f}; }

  // staging descriptors of this thread's vectors i = tid + 512 sl = 64 (wave + 8 sl) + lane: element offset relative to the
  // tile's db origin, (row, column) in the tile (0x7fff7fff: no source -> zero)
  int rel[SL], yx[SL];
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    const int i_ = tid + sl * FB_NTHR;
    const int p_ = i_ / DW_CV;
    const int ix_ = p_ % FB_DW, q_ = p_ / FB_DW;
    const int iy_ = q_ % FB_DH, t_ = q_ / FB_DH;
    const bool use_ = i_ < NI && c_ok && t_ < g.T;
    rel[sl] = ((t_ * g.Ho + iy_) * g.Wo + ix_) * g.Cp + cb8;
    yx[sl] = use_ ? (iy_ | (ix_ << 16)) : 0x7fff7fff;
  }
  const int rel_safe = (g.Wo + 1) * g.Cp + c0;            // the tile's first interior pixel, frame 0, first vector of the chunk
  // `a` pieces of this wave (A_LDS): lane -> (frame, pixel of the wave's tile row, vector)
  int arel[NA > 0 ? NA : 1];
  unsigned aok = 0;                                        // bit j: piece j of this lane has a source apart from the tile-edge test
  int apx[NA > 0 ? NA : 1];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int idx = j * 64 + lane;
    const int vec = idx & (DW_CV - 1), combo = idx >> 2;
    const int p_ = combo & 7, t_ = combo >> 3;
    arel[j] = ((t_ * g.H + wave) * g.W + p_) * g.Cp + c0 + vec * 8;
    apx[j] = p_;
    if (t_ < TT && t_ < g.T && c0 + vec * 8 < g.Cp) aok |= 1u << j;
  }
  typename R4::type ar[A_LDS ? 1 : TT];                    // (!A_LDS) this thread's `a` rows of the next tile
  const int orel = (py * g.W + px) * g.Cp + cb4, ofr = g.H * g.W * g.Cp;
  const int n_raw = (NP - 1 - wave) / 8 + 1;               // pieces per plane of this wave (wave-uniform)
  const int n_dma = 2 * n_raw + NA;                        // DMA instructions per tile of this wave

  // ---- requests of a tile: wave-uniform bases (RB_BASES), then one instruction per index Q (raw pieces 2 sl + plane, then
  //      the `a` pieces); RB_ISSUE = all of them in a burst
  int64_t ri_tb = 0, ri_ab = 0;
  uint32_t ri_sb = 0;
  int ri_dy0 = 0, ri_dx0 = 0;
  bool ri_row_ok = false, ri_on = false;
#define RB_BASES(TL, SIDX)                                                                        \
  {                                                                                               \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                         \
    ri_dy0 = ty_ * FB_TH - 1; ri_dx0 = tx_ * FB_TW - 1;                                           \
    ri_tb = ((((int64_t)b * g.T) * g.Ho + ri_dy0) * g.Wo + ri_dx0) * g.Cp;  /* wave-uniform */    \
    ri_sb = ring_lds + (uint32_t)(SIDX) * P::SLOT_BYTES;                                          \
    ri_ab = ((((int64_t)b * g.T) * g.H + ty_ * FB_TH) * g.W + tx_ * FB_TW) * g.Cp;                \
    ri_row_ok = ty_ * FB_TH + wave < g.H;                                                         \
  }
#define RB_ISSUE_Q(Q)                                                                             \
  if (ri_on) {                                                                                    \
    if constexpr ((Q) < 2 * SL) {                                                                 \
      constexpr int sl_ = (Q) >> 1;                                                               \
      if (wave + 8 * sl_ < NP) {                                                                  \
        const unsigned gy_ = (unsigned)(ri_dy0 + (yx[sl_] & 0xffff));                             \
        const unsigned gx_ = (unsigned)(ri_dx0 + (yx[sl_] >> 16));                                \
        const int off_ = (gy_ < (unsigned)g.Ho && gx_ < (unsigned)g.Wo) ? rel[sl_] : rel_safe;    \
        fb_glds16((((Q) & 1) ? bb : t1) + ri_tb + off_,                                           \
                  ri_sb + (uint32_t)(((Q) & 1) * NPL * 16) + (uint32_t)(wave + 8 * sl_) * 1024u); \
      }                                                                                           \
    } else if constexpr ((Q) < 2 * SL + NA) {                                                     \
      constexpr int j_ = (Q) - 2 * SL;                                                            \
      if (j_ * 64 + lane < TT * 8 * DW_CV) {                                                      \
        const int ax0_ = ri_dx0 + 1;                                                              \
        const int off_ = (ri_row_ok && ((aok >> j_) & 1u) && ax0_ + apx[j_] < g.W) ? arel[j_] : c0; \
        fb_glds16(a + ri_ab + off_, ri_sb + (uint32_t)(2 * NPL * 16) + (uint32_t)wave * (uint32_t)P::A_WAVE_BYTES + (uint32_t)j_ * 1024u); \
      }                                                                                           \
    }                                                                                             \
  }
#define RB_ISSUE_ALL                                                                              \
  { RB_ISSUE_Q(0) RB_ISSUE_Q(1) RB_ISSUE_Q(2) RB_ISSUE_Q(3) RB_ISSUE_Q(4) RB_ISSUE_Q(5) RB_ISSUE_Q(6) RB_ISSUE_Q(7) RB_ISSUE_Q(8) RB_ISSUE_Q(9) }
  static_assert(2 * SL + NA <= 10, "RB_ISSUE_ALL covers ten request indices");
  // (!A_LDS) this thread's `a` rows of tile TL -> ar (compiler-visible loads: its own vmcnt bookkeeping covers them; the
  // hidden DMA requests in the queue only make its waits conservative)
#define RB_ISSUE_A(TL)                                                                            \
  if constexpr (!A_LDS) {                                                                         \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                         \
    if (c_ok && ty_ * FB_TH + py < g.H && tx_ * FB_TW + px < g.W) {                               \
      const T* ab_ = a + ((((int64_t)b * g.T) * g.H + ty_ * FB_TH) * g.W + tx_ * FB_TW) * g.Cp;   \
      _Pragma("unroll") for (int t = 0; t < TT; ++t)                                              \
        if (t < g.T) ar[t] = R4::load(ab_ + (orel + t * ofr));                                    \
    }                                                                                             \
  }
  // in-place conversion of this thread's vectors of slot SIDX (tile TL): db = A t1 + B[n] + C b, zero outside the image
#define RB_CONVERT(TL, SIDX)                                                                      \
  {                                                                                               \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                         \
    const int dy0_ = ty_ * FB_TH - 1, dx0_ = tx_ * FB_TW - 1;                                     \
    uint4* p0_ = reinterpret_cast<uint4*>(ring + (size_t)(SIDX) * P::SLOT_BYTES);                 \
    float cA[8], cB[8], cC[8];                                                                    \
    lds8(cf + 0 * 32 + cv * 8, cA);                                                               \
    lds8(cf + 1 * 32 + cv * 8, cB);                                                               \
    lds8(cf + 2 * 32 + cv * 8, cC);                                                               \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                           \
      if (wave + 8 * sl < NP) {                                                                   \
        const int i = tid + sl * FB_NTHR;                                                         \
        const unsigned gy_ = (unsigned)(dy0_ + (yx[sl] & 0xffff));                                \
        const unsigned gx_ = (unsigned)(dx0_ + (yx[sl] >> 16));                                   \
        float f[8];                                                                               \
        if (gy_ < (unsigned)g.Ho && gx_ < (unsigned)g.Wo) {                                       \
          float f2[8];                                                                            \
          Raw8<T>::cvt(p0_[i], f);                                                                \
          Raw8<T>::cvt(p0_[NPL + i], f2);                                                         \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j])); \
        } else {                                                                                  \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) f[j] = 0.f;                               \
        }                                                                                         \
        reinterpret_cast<float4*>(p0_)[i] = make_float4(f[0], f[1], f[2], f[3]);                  \
        reinterpret_cast<float4*>(p0_)[NPL + i] = make_float4(f[4], f[5], f[6], f[7]);            \
      }                                                                                           \
    }                                                                                             \
  }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  ri_on = true;
#pragma unroll
  for (int d = 0; d < AHEAD; ++d)
    if (tl0 + d < tl1) { RB_BASES(tl0 + d, d) RB_ISSUE_ALL }
  if (tl0 < tl1) RB_ISSUE_A(tl0)
  __syncthreads();   // wl / cf staged
  if (tl0 < tl1) {
    const int younger = tl1 - tl0 - 1 < AHEAD - 1 ? tl1 - tl0 - 1 : AHEAD - 1;   // tiles requested after tile tl0
    fb_wait_vm(A_LDS ? younger * n_dma : 0);
    RB_CONVERT(tl0, 0)
  }
  FCLK(0)

  int slot = 0;   // ring slot of tile tl
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int y0 = ty * FB_TH, x0 = tx * FB_TW;
    const int slot1 = slot + 1 == P::R ? 0 : slot + 1;
    const int slotn = slot + AHEAD >= P::R ? slot + AHEAD - P::R : slot + AHEAD;   // the slot of tile tl + AHEAD = of tile tl - 1
    // this thread's `a` rows and BatchNorm_a parameters: the LDS reads go out before the barrier / the request burst
    const float4* tb = reinterpret_cast<const float4*>(ring + (size_t)slot * P::SLOT_BYTES);
    typename R4::type arc[TT];
    f32x2_t ain[TT][2];
    f32x2_t acc[TT][2];
    float sa[4], sb[4];
    const bool p_ok = c_ok && y0 + py < g.H && x0 + px < g.W;
    if constexpr (A_LDS) {
      // [wave][frame][pixel][vector] 16-byte vectors, this thread's channel half (its own wave's requests: landed since the
      // counted wait in front of the conversion of this slot)
      const uint2* ab = reinterpret_cast<const uint2*>(ring + (size_t)slot * P::SLOT_BYTES + 2 * NPL * 16 + wave * P::A_WAVE_BYTES) + (px * DW_CV + cv) * 2 + h;
#pragma unroll
      for (int t = 0; t < TT; ++t) arc[t] = ab[t * 8 * DW_CV * 2];
    } else {
#pragma unroll
      for (int t = 0; t < TT; ++t) arc[t] = ar[t];
    }
    lds4(cf + 3 * 32 + cv * 8 + h * 4, sa);
    lds4(cf + 4 * 32 + cv * 8 + h * 4, sb);
    __syncthreads();   // conversions of tile tl visible; every wave is past the tap walk of tile tl - 1 (slotn)
    FCLK(3)
    ri_on = tl + AHEAD < tl1;
    if (ri_on) RB_BASES(tl + AHEAD, slotn)
    if constexpr (!SPREAD) RB_ISSUE_ALL
    if (tl + 1 < tl1) RB_ISSUE_A(tl + 1)
    FCLK(2)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (p_ok && t < g.T) {
        float av[4];
        R4::cvt(arc[t], av);
        ain[t][0] = f32x2_t{fmaxf(fmaf(av[0], sa[0], sb[0]), 0.f), fmaxf(fmaf(av[1], sa[1], sb[1]), 0.f)};
        ain[t][1] = f32x2_t{fmaxf(fmaf(av[2], sa[2], sb[2]), 0.f), fmaxf(fmaf(av[3], sa[3], sb[3]), 0.f)};
      } else {
        ain[t][0] = f32x2_t{0.f, 0.f}; ain[t][1] = f32x2_t{0.f, 0.f};
      }
      acc[t][0] = f32x2_t{0.f, 0.f}; acc[t][1] = f32x2_t{0.f, 0.f};
    }
    FCLK(1)
    // ---- 27 taps (the register-prefetch kernel's walk: two fragment slots, LDS reads of step s+1 before the FMAs of step s)
    const float4* tp = tb + (size_t)h * NPL + (py * FB_DW + px) * DW_CV + cv;
    if constexpr (TT <= 3) {
      float4 wq[2][3], hq[2][TT];
#define FB_LOAD(S, SLOT)                                                                                          \
  {                                                                                                               \
    constexpr int ky_ = (S) / 3, kx_ = (S) % 3;                                                                   \
    _Pragma("unroll") for (int kt = 0; kt < 3; ++kt)                                                             \
      wq[SLOT][kt] = *reinterpret_cast<const float4*>(wl + (kt * 9 + (S)) * 32 + cv * 8 + h * 4);                \
    _Pragma("unroll") for (int to = 0; to < TT; ++to)                                                            \
      hq[SLOT][to] = tp[((to * FB_DH + (2 - ky_)) * FB_DW + (2 - kx_)) * DW_CV];                                 \
  }
#define FB_STEP(S, SLOT)                                                                                          \
  {                                                                                                               \
    _Pragma("unroll") for (int to = 0; to < TT; ++to) {                                                          \
      const f32x2_t v0 = {hq[SLOT][to].x, hq[SLOT][to].y}, v1 = {hq[SLOT][to].z, hq[SLOT][to].w};                 \
      _Pragma("unroll") for (int kt = 0; kt < 3; ++kt) {                                                         \
        const int ti = to + kt - 1;                                                                               \
        if (ti >= 0 && ti < TT) {                                                                                 \
          const f32x2_t w0 = {wq[SLOT][kt].x, wq[SLOT][kt].y}, w1 = {wq[SLOT][kt].z, wq[SLOT][kt].w};             \
          acc[ti][0] = __builtin_elementwise_fma(v0, w0, acc[ti][0]);                                             \
          acc[ti][1] = __builtin_elementwise_fma(v1, w1, acc[ti][1]);                                             \
          dwa[kt * 9 + (S)][0] = __builtin_elementwise_fma(v0, ain[ti][0], dwa[kt * 9 + (S)][0]);                 \
          dwa[kt * 9 + (S)][1] = __builtin_elementwise_fma(v1, ain[ti][1], dwa[kt * 9 + (S)][1]);                 \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    pin_acc<TT>(acc);                                                                                             \
    pin_dw(dwa[(S)], dwa[9 + (S)], dwa[18 + (S)]);                                                                \
    if constexpr (SPREAD) { RB_ISSUE_Q(S) if constexpr ((S) == 8) { RB_ISSUE_Q(9) } }                             \
  }
      FB_LOAD(0, 0)
      FB_LOAD(1, 1) FB_STEP(0, 0)
      FB_LOAD(2, 0) FB_STEP(1, 1)
      FB_LOAD(3, 1) FB_STEP(2, 0)
      FB_LOAD(4, 0) FB_STEP(3, 1)
      FB_LOAD(5, 1) FB_STEP(4, 0)
      FB_LOAD(6, 0) FB_STEP(5, 1)
      FB_LOAD(7, 1) FB_STEP(6, 0)
      FB_LOAD(8, 0) FB_STEP(7, 1)
      FB_STEP(8, 0)
#undef FB_LOAD
#undef FB_STEP
    } else {
      fb_taps_s1<TT>(tp, wl + cv * 8 + h * 4, acc, dwa, ain);   // five frames (SCD): plain walk, plane stride NPL in `tp`
      if constexpr (SPREAD) RB_ISSUE_ALL
    }
    FCLK(4)
    // ---- the next tile's requests have landed once at most the requests of the tiles after it are outstanding
    if (tl + 1 < tl1) {
      const int younger = tl1 - tl - 2 < AHEAD - 1 ? tl1 - tl - 2 : AHEAD - 1;   // tiles tl + 2 .. tl + AHEAD that exist
      fb_wait_vm(A_LDS ? younger * n_dma : 0);
    }
    FCLK(6)
    // ---- mask, store t2, BN_a-backward sums
    if (p_ok) {
      float ma[4], ra[4];
      lds4(cf + 5 * 32 + cv * 8 + h * 4, ma);
      lds4(cf + 6 * 32 + cv * 8 + h * 4, ra);
      T* ob = t2 + ((((int64_t)b * g.T) * g.H + y0) * g.W + x0) * g.Cp;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < g.T) {
          float av[4], o[4];
          R4::cvt(arc[t], av);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = round_as<T>(ain[t][j >> 1][j & 1] > 0.f ? acc[t][j >> 1][j & 1] : 0.f);
            o[j] = d;
            S1[j] += d; S2[j] += d * ((av[j] - ma[j]) * ra[j]);
          }
          R4::store(ob + (orel + t * ofr), o);
        }
      }
    }
    FCLK(5)
    if (tl + 1 < tl1) RB_CONVERT(tl + 1, slot1)
    FCLK(8)
    slot = slot1;
  }
#undef RB_BASES
#undef RB_ISSUE_Q
#undef RB_ISSUE_ALL
#undef RB_ISSUE_A
#undef RB_CONVERT

  fb_flush(ring, S1, S2, dwa, dsums, dw, g, c0);
  FCLK(7)
  FCLK_FLUSH
}

template <int TT, bool SPREAD>
int launch_ring_t(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const float* w,
                  const void* a, const float* ss_a, const float* mr_a, void* t2, double* dsums, float* dw,
                  const DwGeom& g, hipStream_t stream, const c3d_bn_fin& fin) {
  typedef RingPlan<TT> P;
  static_assert(P::LDS_BYTES <= 160 * 1024, "ring does not fit");
  static_assert((size_t)P::R * P::SLOT_BYTES >= (size_t)27 * (FB_NTHR + 48) * sizeof(float), "dump region");
  static_assert((P::AHEAD - 1) * (2 * P::SL + P::NA) <= 12, "fb_wait_vm covers counts up to 12");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_ring_kernel<TT, SPREAD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.W + FB_TW - 1) / FB_TW) * ((g.H + FB_TH - 1) / FB_TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  static const int env_tpw = c3d_env("C3D_DWBF_TPW") ? atoi(c3d_env("C3D_DWBF_TPW")) : 0;
  static const int env_max = c3d_env("C3D_DWBF_MAX") ? atoi(c3d_env("C3D_DWBF_MAX")) : 64;
  int tpw = env_max;
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 85L * device_cus() / 100) tpw >>= 1;
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  dw_bwd_ring_kernel<TT, SPREAD><<<grid, dim3(FB_NTHR), P::LDS_BYTES, stream>>>(
      reinterpret_cast<const bf16_t*>(t1), reinterpret_cast<const bf16_t*>(bb), cA, cB, cC, w, reinterpret_cast<const bf16_t*>(a),
      ss_a, mr_a, reinterpret_cast<bf16_t*>(t2), dsums, dw, g, tpw, fin);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int TT, int S>
int launch_fused_t(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const float* w,
                   const void* a, const float* ss_a, const float* mr_a, void* t2, double* dsums, float* dw,
                   const DwGeom& g, hipStream_t stream, const c3d_bn_fin& fin) {
  constexpr int NI = TT * FB_DH * FB_DW * DW_CV;
  const size_t lds = (27 * 32 + 7 * 32) * sizeof(float) + (size_t)2 * 2 * NI * sizeof(float4);
  static_assert((size_t)2 * 2 * NI * sizeof(float4) >= (size_t)27 * (FB_NTHR + 48) * sizeof(float), "dump region");
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_fused_kernel<T, TT, S>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.W + FB_TW * S - 1) / (FB_TW * S)) * ((g.H + FB_TH * S - 1) / (FB_TH * S));
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  // one 512-thread workgroup is resident per CU: a walk amortises the weight-gradient flush (~3 us) and pipelines the
  // loads; short enough for ~2 rounds of workgroups (C3D_DWBF_TPW: tuning knob)
  static const int env_tpw = c3d_env("C3D_DWBF_TPW") ? atoi(c3d_env("C3D_DWBF_TPW")) : 0;
  // measured on MI355X (B=32 bf16 step, side stream on): 4 / 8 / 16 / 32 / 64 tiles -> 32.9 / 31.3 / 30.6 / 30.4 / 31.2 ms
  // (12, 24: +0.3..1.2 ms -- ragged last groups)
  // -> the longest walk that still gives (almost) every CU a workgroup: 16 / 32 / 32 tiles for the 32x32 / 64x64 / 128x128 stages
  // (round 5, re-swept at 23.2 ms per step: cap 16 / 32 / 64 / 128 tiles -> 23.42 / 23.18 / 22.95 / 23.15 ms; with the narrower
  // side-stream weight gradient, three interleaved repeats: 22.90 against 23.39 ms for the old pair of defaults)
  static const int env_max = c3d_env("C3D_DWBF_MAX") ? atoi(c3d_env("C3D_DWBF_MAX")) : 64;
  int tpw = env_max / (S * S) * (8 / FB_TH);   // a stride-2 tile is four pixels per thread; (half-height tiles: twice as many)
  constexpr int WGC = 512 / FB_NTHR;           // workgroups resident per CU
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 85L * WGC * device_cus() / 100) tpw >>= 1;
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  dw_bwd_fused_kernel<T, TT, S><<<grid, dim3(FB_NTHR), lds, stream>>>(
      reinterpret_cast<const T*>(t1), reinterpret_cast<const T*>(bb), cA, cB, cC, w, reinterpret_cast<const T*>(a),
      ss_a, mr_a, reinterpret_cast<T*>(t2), dsums, dw, g, tpw, fin);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

namespace {
int dispatch_fused(const void* t1, const void* b, const float* coefA, const float* coefB, const float* coefC, const float* w,
                   const void* a, const float* ss_a, const float* mr_a, void* t2, double* dsums, float* dw, const DwGeom& g,
                   int dtype, hipStream_t s, const c3d_bn_fin& fin) {
#define FB_DISPATCH(TY, S_)                                                                                         \
  return g.T <= 3 ? launch_fused_t<TY, 3, S_>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin)  \
                  : launch_fused_t<TY, 5, S_>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin);
  if (dtype == C3D_DT_F32) {
    if (g.stride == 1) { FB_DISPATCH(float, 1) }
    FB_DISPATCH(float, 2)
  }
  if (dtype == C3D_DT_BF16) {
    // LDS-DMA ring kernels (bit 2: requests spread over the tap walk; bit 3: three-frame maps under 64 x 64 too -- their
    // 16-tile walks pay the two-tile ring fill: res4 of the BCD step 65.9 us with the register prefetch, 68.0 us with the
    // ring; the five-frame register kernel spills, its ring variant wins on every map: SCD 663 -> 681 img/s)
    if (C3D_FB_ROWS == 8 && g.stride == 1 && (c3d_option_dw_ring & 1) && ((c3d_option_dw_ring & 8) || g.T > 3 || (long)g.H * g.W >= 64 * 64)) {
      if (g.T <= 3) {
        if (c3d_option_dw_ring & 4) return launch_ring_t<3, true>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin);
        return launch_ring_t<3, false>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin);
      }
      if (c3d_option_dw_ring & 4) return launch_ring_t<5, true>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin);
      return launch_ring_t<5, false>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, s, fin);
    }
    if (g.stride == 1) { FB_DISPATCH(bf16_t, 1) }
    FB_DISPATCH(bf16_t, 2)
  }
#undef FB_DISPATCH
  return C3D_E_BADARG;
}
}  // namespace

extern "C" int c3d_dw333_bwd_fused(const void* t1, const void* b, const float* coefA, const float* coefB,
                                   const float* coefC, const float* w, const void* a, const float* ss_a,
                                   const float* mr_a, void* t2, double* dsums, float* dw, int32_t B, int32_t T,
                                   int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype, void* stream) {
  if (stride != 1 && stride != 2) return C3D_E_UNSUPPORTED;
  DwGeom g{B, T, H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, C, Cp, stride};
  if (!t1 || !b || !coefA || !coefB || !coefC || !w || !a || !ss_a || !mr_a || !t2 || !dsums || !dw || !geom_ok(g))
    return C3D_E_BADARG;
  return dispatch_fused(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, dw, g, dtype,
                        reinterpret_cast<hipStream_t>(stream), c3d_bn_fin{});
}

extern "C" int c3d_dw333_bwd_fused_fin(const void* t1, const void* b, const c3d_bn_fin* fin_b, const float* w, const void* a,
                                       const float* ss_a, const float* mr_a, void* t2, double* dsums, float* dw, int32_t B,
                                       int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                                       void* stream) {
  if (stride != 1 && stride != 2) return C3D_E_UNSUPPORTED;
  DwGeom g{B, T, H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, C, Cp, stride};
  if (!t1 || !b || !w || !a || !ss_a || !mr_a || !t2 || !dsums || !dw || !geom_ok(g)) return C3D_E_BADARG;
  if (!fin_b || !fin_b->sums || fin_b->batch != B || !fin_b->gamma || !fin_b->mr || !(fin_b->count > 0)) return C3D_E_BADARG;
  return dispatch_fused(t1, b, nullptr, nullptr, nullptr, w, a, ss_a, mr_a, t2, dsums, dw, g, dtype,
                        reinterpret_cast<hipStream_t>(stream), *fin_b);
}

#ifdef C3D_PW_CLOCK
extern "C" int c3d_debug_fb_clock(unsigned long long* out, int reset) {   // out[FCLK_WAVES][10]
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_fb_clk), sizeof(unsigned long long) * FCLK_WAVES * 10);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(c3d_fb_clk));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * FCLK_WAVES * 10);
  }
  return (int)e;
}
#endif
