// Depthwise 3x3x3 Conv3d (groups = C, padding 1, stride (1,s,s), s in {1,2}) on the VALU with
// LDS-tiled inputs; replaces conv_b of the X3D bottleneck (reference model/x3d.py:184-193)
// forward, data-gradient and weight-gradient, with the neighbouring BatchNorm arithmetic fused:
//
//   fwd      : in  = relu(a*scale_a + shift_a) applied once per element while staging the tile
//              (zero padding is applied AFTER the activation, as in the reference graph);
//              out = raw conv output b; epilogue = per-(sample,channel) sum / sum-of-squares
//              (feeds BN_b statistics and the SE squeeze).
//   bwd_data : in  = db = A[c]*t1 + B[n][c] + C[c]*b applied while staging (BN_b/SE backward);
//              out = t2 = dconv * (a*scale_a+shift_a > 0); epilogue = per-channel sum t2, sum t2*ahat
//   wgrad    : dW[c][kt][ky][kx] += sum db[out] * relu(bn(a))[in]
//
// All T frames of a spatial tile are resident in LDS (T = 3 for BCD, 5 for SCD).  A thread
// owns one output pixel and one 8-channel vector for all T frames.
#include "pw_common.h"  // common.h + device_cus()
#include "bn_fin.h"
#include "dw_common.h"
#include "launch_hints.h"
#ifdef C3D_TUNING
#include "dw_toeplitz.h"   // Toeplitz-MFMA forward experiment (slower; instrumented build only)
#endif
#include "../../include/change3d_hip.h"
#include <cstdlib>
#include <cstring>

thread_local int c3d_side_launch = 0;

// Workgroups of the (single-round) depthwise weight-gradient kernels: every CU when the kernel has the GPU to itself,
// half of them when the stage driver runs it beside the data-gradient chain (launch_hints.h).
static long dw_wgrad_target_wgs() {
  static const int env_wgs = c3d_env("C3D_DWWG_WGS") ? atoi(c3d_env("C3D_DWWG_WGS")) : 0;
  static const int env_side = c3d_env("C3D_DWWG_SIDE_WGS") ? atoi(c3d_env("C3D_DWWG_SIDE_WGS")) : 0;
  if (env_wgs > 0) return env_wgs;
  if (c3d_side_launch) return env_side > 0 ? env_side : device_cus() / 2;
  return device_cus();
}


#ifdef C3D_PW_CLOCK
// Debug build only (tools/pw_phase_clock.py --dw): per-phase shader-clock sums of the data-gradient kernel.
constexpr int DCLK_WAVES = 16384;
__device__ unsigned long long c3d_dw_clk[DCLK_WAVES][10];
#define DCLK_DECL unsigned long long dclk_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dclk_last_ = __builtin_amdgcn_s_memtime();
#define DCLK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dclk_[i] += t_ - dclk_last_; dclk_last_ = t_; }
#define DCLK_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define DCLK_FLUSH if ((threadIdx.x & 63) == 0) { const int w_ = (blockIdx.x * 4 + (threadIdx.x >> 6)) % DCLK_WAVES; for (int i_ = 0; i_ < 9; ++i_) c3d_dw_clk[w_][i_] += dclk_[i_]; c3d_dw_clk[w_][9] += 1ull; }
#else
#define DCLK_DECL
#define DCLK(i)
#define DCLK_WAITVM
#define DCLK_FLUSH
#endif

namespace {

template <typename T> struct LdsStore;  // tile element type in LDS
template <> struct LdsStore<float> { typedef float type; };
template <> struct LdsStore<bf16_t> { typedef bf16_t type; };

// raw (unconverted) 8-element vectors: what a register prefetch holds between issue and use
template <typename T> struct RawD;
template <> struct RawD<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RawD<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};


__device__ __forceinline__ void lds_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ----------------------------------------------------------------------------------------------
// Forward.  grid = (tiles_x*tiles_y, channel chunks, B); block = TH*TW*DW_CV threads.
template <typename T, int S, int TH, int TW, int TT>
__global__ __launch_bounds__(TH * TW * DW_CV) void dw_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ w, T* __restrict__ y,
    double* __restrict__ nc, const DwGeom g, const int tiles_per_wg) {
  typedef typename LdsStore<T>::type L;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int NTHR = TH * TW * DW_CV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                         // [27][32]
  float* red = wl + 27 * 32;                                          // [NTHR/64][DW_CV][16]
  L* tile = reinterpret_cast<L*>(red + (NTHR / 64) * DW_CV * 16);     // [T][IH][IW][32]

  const int tid = threadIdx.x;
  const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx_ = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx_ * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx_, tg = co.group % gx_;
  const int c0 = chunk * DW_CV * 8;
  const int cv = tid % DW_CV;
  const int cbase = c0 + cv * 8;
  const bool c_ok = cbase < g.Cp;

  // weights -> LDS as [tap][32 channels] (zero for channels >= C)
  for (int i = tid; i < 27 * 32; i += NTHR) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = c_ok ? ss[cbase + j] : 0.f; sh[j] = c_ok ? ss[g.Cp + cbase + j] : 0.f; }

  // a workgroup walks `tiles_per_wg` tiles (one statistics flush; TT sizes the accumulators)
  const int pix = tid / DW_CV;
  const int px = pix % TW, py = pix / TW;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int tl0 = tg * tiles_per_wg;
  const int tl1 = tl0 + tiles_per_wg < ntiles ? tl0 + tiles_per_wg : ntiles;
  for (int tl = tl0; tl < tl1; ++tl) {
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  __syncthreads();   // previous tile consumed (and the weights are staged)
  // input tile (all T frames) with the BN+ReLU prologue; zero outside the image
  const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
  const int items = g.T * IH * IW * DW_CV;
  for (int i = tid; i < items; i += NTHR) {  // NTHR % DW_CV == 0 -> cv fixed per thread
    const int p = i / DW_CV;
    const int ix = p % IW;
    const int q = p / IW;
    const int iy = q % IH, t = q / IH;
    const int gy = iy0 + iy, gx = ix0 + ix;
    float f[8];
    if (c_ok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
      Vec8<T>::load(x + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * g.Cp + cbase, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
    }
    Vec8<L>::store(tile + (size_t)p * 32 + cv * 8, f);
  }
  __syncthreads();

  const int oy = ty * TH + py, ox = tx * TW + px;
  const bool p_ok = c_ok && oy < g.Ho && ox < g.Wo;

  float acc[TT][8];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;

#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float wk[3][8];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const float4 w0 = *reinterpret_cast<const float4*>(wl + (kt * 9 + ky * 3 + kx) * 32 + cv * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + (kt * 9 + ky * 3 + kx) * 32 + cv * 8 + 4);
        wk[kt][0] = w0.x; wk[kt][1] = w0.y; wk[kt][2] = w0.z; wk[kt][3] = w0.w;
        wk[kt][4] = w1.x; wk[kt][5] = w1.y; wk[kt][6] = w1.z; wk[kt][7] = w1.w;
      }
#pragma unroll
      for (int ti = 0; ti < TT; ++ti) {
        if (ti < g.T) {
          float v[8];
          Vec8<L>::load(tile + ((size_t)(ti * IH + py * S + ky) * IW + px * S + kx) * 32 + cv * 8, v);
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT && to < g.T) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[to][j] = fmaf(v[j], wk[kt][j], acc[to][j]);
            }
          }
        }
      }
    }
  }

  if (p_ok) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < g.T) {
        Vec8<T>::store(y + ((((size_t)b * g.T + t) * g.Ho + oy) * g.Wo + ox) * g.Cp + cbase, acc[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float r = round_as<T>(acc[t][j]);
          s1[j] += r; s2[j] += r * r;
        }
      }
    }
  }
  }  // tile walk
  if (nc == nullptr) return;
  // reduce over the pixels of the workgroup: lanes with equal cv inside a wave, then across waves
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = DW_CV; o < 64; o <<= 1) {
      s1[j] += __shfl_xor(s1[j], o, 64);
      s2[j] += __shfl_xor(s2[j], o, 64);
    }
  }
  if (lane < DW_CV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(wave * DW_CV + lane) * 16 + j] = s1[j];
      red[(wave * DW_CV + lane) * 16 + 8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (tid < DW_CV * 16) {
    const int v = tid / 16, k = tid & 15;
    float s = 0.f;
    for (int wv = 0; wv < NTHR / 64; ++wv) s += red[(wv * DW_CV + v) * 16 + k];
    const int c = c0 + v * 8 + (k & 7);
    if (c < g.C) atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + (k >> 3), (double)s);
  }
}

// ----------------------------------------------------------------------------------------------
// Data gradient.  Output positions are INPUT-resolution pixels; the staged tile is db at
// output resolution (+halo), built on load from (t1, b, coefA, coefB[n], coefC).
//   * a workgroup walks `tiles_per_wg` tiles of one (sample, 32-channel chunk): the next tile's raw
//     (t1, b) rows and this tile's `a` rows (mask + BN_a-backward sums) are in flight while the
//     taps of the current tile run, and the BN_a sums are flushed once per workgroup (the
//     one-tile-per-workgroup version was bound by load latency and by 64 f64 atomics per tile);
//   * the staged db tile is f32 in two half-vector planes [half][t][y][x][cv] of float4: no
//     bf16->f32 conversion per tap (integer VALU ops run at half the f32 FMA rate on gfx950) and a
//     wave's lanes read consecutive 16 B (conflict-free);
//   * per-channel coefficient vectors live in LDS (register budget: 3 workgroups per CU).
template <typename T, int S, int TH, int TW, int TT>
__global__ __launch_bounds__(TH * TW * DW_CV) void dw_bwd_data_kernel(
    const T* __restrict__ t1, const T* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const float* __restrict__ w,
    const T* __restrict__ a, const float* __restrict__ ss_a, const float* __restrict__ mr_a, T* __restrict__ t2,
    double* __restrict__ dsums, const DwGeom g, const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef RawD<T> RW;
  constexpr int DH = (S == 1) ? TH + 2 : TH / 2 + 2;
  constexpr int DW_ = (S == 1) ? TW + 2 : TW / 2 + 2;
  constexpr int NTHR = TH * TW * DW_CV;
  constexpr int NI = TT * DH * DW_ * DW_CV;       // staged vectors per tile
  constexpr int SL = (NI + NTHR - 1) / NTHR;      // prefetch slots per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);     // [27][32]
  float* cf = wl + 27 * 32;                       // [7][32]: cA, cB(sample), cC, sa, sb, ma, ra
  float* red = cf + 7 * 32;                       // [waves][DW_CV][16]
  float4* tile = reinterpret_cast<float4*>(red + (NTHR / 64) * DW_CV * 16);  // [2][TT][DH][DW_][DW_CV]
  constexpr int plane = NI;                       // float4 units per half-vector plane

  const int tid = threadIdx.x;
  DCLK_DECL
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx, tg = co.group % gx;
  const int c0 = chunk * DW_CV * 8;
  const int cv = tid % DW_CV;
  const int cbase = c0 + cv * 8;
  const bool c_ok = cbase < g.Cp;

  for (int i = tid; i < 27 * 32; i += NTHR) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  for (int i = tid; i < 7 * 32; i += NTHR) {
    const int k = i >> 5, c = c0 + (i & 31);
    float v = 0.f;
    if (c < g.Cp) {
      v = k == 0 ? coefA[c] : k == 1 ? coefB[(size_t)b * g.Cp + c] : k == 2 ? coefC[c] : k == 3 ? ss_a[c]
        : k == 4 ? ss_a[g.Cp + c] : k == 5 ? mr_a[c] : mr_a[g.Cp + c];
    }
    cf[i] = v;
  }
  // BN_a-backward sums (sum t2, sum t2*ahat): f32 within a tile, f64 across the walk -- the two terms
  // nearly cancel on some channels and a long f32 chain cost 1-2 digits of d gamma there
  double S1[8], S2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { S1[j] = 0.0; S2[j] = 0.0; }

  typename RW::type r1[SL], r2[SL];
  unsigned vmask = 0;
  // Per-slot staging descriptors, computed once: the kernel is VALU-bound (4 waves per SIMD, in-kernel clocks), and
  // the per-tile decode of (frame, row, column) with its 64-bit offset multiplies was ~70 quarter-rate integer
  // instructions per tile and thread.  rel = offset of the slot's element relative to the tile origin, yx = its
  // (row, column) inside the staged tile (0x7fff: slot not in use).
  int rel[SL], yx[SL];
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    const int i_ = tid + sl * NTHR;
    const int p_ = i_ / DW_CV;
    const int ix_ = p_ % DW_, q_ = p_ / DW_;
    const int iy_ = q_ % DH, t_ = q_ / DH;
    const bool use_ = i_ < NI && c_ok && t_ < g.T;
    rel[sl] = ((t_ * g.Ho + iy_) * g.Wo + ix_) * g.Cp + cbase;
    yx[sl] = use_ ? (iy_ | (ix_ << 16)) : 0x7fff7fff;
  }
#define BD_ISSUE(TL)                                                                             \
  {                                                                                              \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                        \
    const int dy0_ = (S == 1) ? ty_ * TH - 1 : ((ty_ * TH) >> 1) - 1;                            \
    const int dx0_ = (S == 1) ? tx_ * TW - 1 : ((tx_ * TW) >> 1) - 1;                            \
    const int64_t tb_ = ((((int64_t)b * g.T) * g.Ho + dy0_) * g.Wo + dx0_) * g.Cp;  /* wave-uniform */ \
    const T* t1b_ = t1 + tb_;                                                                    \
    const T* bbb_ = bb + tb_;                                                                    \
    vmask = 0;                                                                                   \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                          \
      const unsigned gy_ = (unsigned)(dy0_ + (yx[sl] & 0xffff));                                 \
      const unsigned gx_ = (unsigned)(dx0_ + (yx[sl] >> 16));                                    \
      if (gy_ < (unsigned)g.Ho && gx_ < (unsigned)g.Wo) {                                        \
        r1[sl] = RW::load(t1b_ + rel[sl]);                                                       \
        r2[sl] = RW::load(bbb_ + rel[sl]);                                                       \
        vmask |= 1u << sl;                                                                       \
      }                                                                                          \
    }                                                                                            \
  }

  const int pix = tid / DW_CV;
  const int px = pix % TW, py = pix / TW;
  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) BD_ISSUE(tl0)
  DCLK(0)
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int dy0 = (S == 1) ? y0 - 1 : (y0 >> 1) - 1;   // first db row/col held in the tile
    const int dx0 = (S == 1) ? x0 - 1 : (x0 >> 1) - 1;
    __syncthreads();   // the previous tile's taps are done (and wl/cf are visible on the first pass)
    DCLK(1)
    DCLK_WAITVM
    DCLK(2)
    {
      float cA[8], cB[8], cC[8];
      lds_ld8(cf + 0 * 32 + cv * 8, cA);
      lds_ld8(cf + 1 * 32 + cv * 8, cB);
      lds_ld8(cf + 2 * 32 + cv * 8, cC);
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int i = tid + sl * NTHR;
        if (i < NI) {
          float f[8];
          if ((vmask >> sl) & 1u) {
            float f2[8];
            RW::cvt(r1[sl], f);
            RW::cvt(r2[sl], f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
          tile[i] = make_float4(f[0], f[1], f[2], f[3]);            // i = p * DW_CV + cv
          tile[plane + i] = make_float4(f[4], f[5], f[6], f[7]);
        }
      }
    }
    DCLK(3)
    // this tile's `a` rows (epilogue operands) first, then the next tile's raw rows: vmcnt is in
    // order, so the epilogue waits only for what it needs
    const int iy = y0 + py, ix = x0 + px;
    const bool p_ok = c_ok && iy < g.H && ix < g.W;
    const int64_t ob = ((((int64_t)b * g.T) * g.H + y0) * g.W + x0) * g.Cp;   // wave-uniform tile origin (a, t2)
    const int orel = (py * g.W + px) * g.Cp + cbase, ofr = g.H * g.W * g.Cp;   // lane offset, frame stride
    typename RW::type ar[TT];
    if (p_ok) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
        if (t < g.T) ar[t] = RW::load(a + ob + (orel + t * ofr));
    }
    if (tl + 1 < tl1) BD_ISSUE(tl + 1)
    DCLK(4)
    __syncthreads();
    DCLK(5)

    float acc[TT][8];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ny = iy + 1 - ky;  // = S * oy
      if (S == 2 && (ny & 1)) continue;
      const int ly = ((S == 1) ? ny : (ny >> 1)) - dy0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int nx = ix + 1 - kx;
        if (S == 2 && (nx & 1)) continue;
        const int lx = ((S == 1) ? nx : (nx >> 1)) - dx0;
        float wk[3][8];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) lds_ld8(wl + (kt * 9 + ky * 3 + kx) * 32 + cv * 8, wk[kt]);
#pragma unroll
        for (int to = 0; to < TT; ++to) {
          if (to < g.T) {
            const int pi = ((to * DH + ly) * DW_ + lx) * DW_CV + cv;
            const float4 h0 = tile[pi], h1 = tile[plane + pi];
            const float v[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
              const int ti = to + kt - 1;  // d in[ti] += d out[to] * w[kt]
              if (ti >= 0 && ti < TT && ti < g.T) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[ti][j] = fmaf(v[j], wk[kt][j], acc[ti][j]);
              }
            }
          }
        }
      }
    }
    DCLK(6)
    if (p_ok) {
      float s1[8], s2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
      float sa[8], sb[8], ma[8], ra[8];
      lds_ld8(cf + 3 * 32 + cv * 8, sa);
      lds_ld8(cf + 4 * 32 + cv * 8, sb);
      lds_ld8(cf + 5 * 32 + cv * 8, ma);
      lds_ld8(cf + 6 * 32 + cv * 8, ra);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < g.T) {
          float av[8], o[8];
          RW::cvt(ar[t], av);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float pa = fmaf(av[j], sa[j], sb[j]);
            const float d = round_as<T>(pa > 0.f ? acc[t][j] : 0.f);
            o[j] = d;
            s1[j] += d; s2[j] += d * ((av[j] - ma[j]) * ra[j]);
          }
          Vec8<T>::store(t2 + ob + (orel + t * ofr), o);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { S1[j] += (double)s1[j]; S2[j] += (double)s2[j]; }
    }
    DCLK(7)
  }
#undef BD_ISSUE
  const int lane = tid & 63, wave = tid >> 6;
  double* red64 = reinterpret_cast<double*>(tile);   // [waves][DW_CV][16]; the tile is dead now
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = DW_CV; o < 64; o <<= 1) {
      S1[j] += __shfl_xor(S1[j], o, 64);
      S2[j] += __shfl_xor(S2[j], o, 64);
    }
  }
  if (lane < DW_CV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red64[(wave * DW_CV + lane) * 16 + j] = S1[j];
      red64[(wave * DW_CV + lane) * 16 + 8 + j] = S2[j];
    }
  }
  __syncthreads();
  if (tid < DW_CV * 16) {
    const int v = tid / 16, k = tid & 15;
    double sacc = 0.0;
    for (int wv = 0; wv < NTHR / 64; ++wv) sacc += red64[(wv * DW_CV + v) * 16 + k];
    const int c = c0 + v * 8 + (k & 7);
    if (c < g.C) atomicAdd(dsums + (size_t)(k >> 3) * g.C + c, sacc);
  }
  if (fin.ticket) {   // last workgroup: BatchNorm_a backward coefficients (no separate c3d_bn_bwd_coef launch)
    // workgroups padded onto the grid by chunk_order_grid() returned at the top without a ticket
    const uint32_t active = (uint32_t)((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8)) * (uint32_t)(gx * g.B);
    if (c3dfin::last_workgroup(fin.ticket, active, reinterpret_cast<int*>(wl)))
      c3dfin::bn_backward(fin, dsums, 1, g.C, g.Cp, tid, NTHR);
  }
  DCLK(8)
  DCLK_FLUSH
}

// Asynchronous LDS vector read of 8 tile elements with an explicit wait, for hand-pipelined
// inner loops (inline asm: the wait names the destination so its consumers cannot move above it).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <typename P> __device__ __forceinline__ uint32_t lds_addr(const P* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));
}
template <typename L> struct LdsVec;
template <> struct LdsVec<bf16_t> {
  static constexpr int N = 1;  // LDS instructions per issue
  struct raw_t { u32x4_t v; };
  static __device__ __forceinline__ void issue(raw_t& r, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(r.v) : "v"(addr));
  }
  template <int CNT> static __device__ __forceinline__ void wait(raw_t& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r.v) : "n"(CNT));
  }
  static __device__ __forceinline__ void cvt(const raw_t& r, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(r.v[i] << 16);
      f[2 * i + 1] = __uint_as_float(r.v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void cvt2(const raw_t& r, f32x2_t (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = f32x2_t{__uint_as_float(r.v[i] << 16), __uint_as_float(r.v[i] & 0xffff0000u)};
  }
};
template <> struct LdsVec<float> {
  static constexpr int N = 2;
  struct raw_t { u32x4_t a, b; };
  static __device__ __forceinline__ void issue(raw_t& r, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(r.a), "=&v"(r.b) : "v"(addr));
  }
  template <int CNT> static __device__ __forceinline__ void wait(raw_t& r) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r.a), "+v"(r.b) : "n"(CNT));
  }
  static __device__ __forceinline__ void cvt(const raw_t& r, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = __uint_as_float(r.a[i]); f[4 + i] = __uint_as_float(r.b[i]); }
  }
  static __device__ __forceinline__ void cvt2(const raw_t& r, f32x2_t (&f)[4]) {
    f[0] = f32x2_t{__uint_as_float(r.a[0]), __uint_as_float(r.a[1])};
    f[1] = f32x2_t{__uint_as_float(r.a[2]), __uint_as_float(r.a[3])};
    f[2] = f32x2_t{__uint_as_float(r.b[0]), __uint_as_float(r.b[1])};
    f[3] = f32x2_t{__uint_as_float(r.b[2]), __uint_as_float(r.b[3])};
  }
};

// Consumer side of the weight gradient: one thread = (output pixel, 8-channel vector, temporal tap kt);
// acc[ky*3+kx][pair] += db[to] * a[to+kt-1][ky][kx] over the frames of one LDS-resident tile.
// Explicit one-deep LDS pipeline: left to itself the compiler issues all nine tap reads, converts
// them, and only then starts the FMAs (~160 live VGPRs; the budget of a 1024-thread workgroup is
// 128) -- the empty asm after each tap pins that tap's FMAs in place.
template <typename L, int S, int TH, int TW>
__device__ __forceinline__ void wgrad_tile_taps(f32x2_t (&acc)[9][4], const L* atile, const L* dtile, const int T,
                                                const int kt, const int px, const int py, const int cv) {
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  for (int to = 0; to < T; ++to) {
    const int ti = to + kt - 1;
    if (ti < 0 || ti >= T) continue;
    const uint32_t d_addr = lds_addr(dtile + ((size_t)(to * TH + py) * TW + px) * 32 + cv * 8);
    const uint32_t a_addr = lds_addr(atile + ((size_t)(ti * IH + py * S) * IW + px * S) * 32 + cv * 8);
    typename LdsVec<L>::raw_t dr, rr[2];
    LdsVec<L>::issue(dr, d_addr);
    LdsVec<L>::issue(rr[0], a_addr);
    LdsVec<L>::template wait<LdsVec<L>::N>(dr);
    f32x2_t d[4];
    LdsVec<L>::cvt2(dr, d);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k < 8) {
        const int kn = k + 1;
        LdsVec<L>::issue(rr[kn & 1], a_addr + (uint32_t)(((kn / 3) * IW + (kn % 3)) * 32 * sizeof(L)));
        LdsVec<L>::template wait<LdsVec<L>::N>(rr[k & 1]);
      } else {
        LdsVec<L>::template wait<0>(rr[k & 1]);
      }
      f32x2_t r[4];
      LdsVec<L>::cvt2(rr[k & 1], r);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] = __builtin_elementwise_fma(d[j], r[j], acc[k][j]);
      asm volatile("" : "+v"(acc[k][0]), "+v"(acc[k][1]), "+v"(acc[k][2]), "+v"(acc[k][3]));
    }
  }
}

// Reduce the per-thread partial sums across the pixels of each wave (lanes with equal channel
// vector; kt is wave-uniform) into the workgroup's [27][32] LDS accumulator.
__device__ __forceinline__ void wgrad_reduce_to_lds(const f32x2_t (&acc)[9][4], float* red, const int kt,
                                                    const int lane) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[k][j >> 1][j & 1];
#pragma unroll
      for (int o = DW_CV; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane < DW_CV) atomicAdd(&red[(kt * 9 + k) * 32 + lane * 8 + j], v);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Weight gradient, producer/consumer waves.
//   * consumer threads = TH*TW pixels x DW_CV channel vectors x 3 temporal taps; each keeps its
//     9x8 partial sums in registers for the whole walk of the workgroup;
//   * DW_LOADERS extra threads stage the NEXT tile (relu(bn(a)) with halo, db = A*t1 + B + C*b)
//     into the other LDS buffer while the consumers run the current one: one barrier per tile and
//     the global-load latency never sits in front of the FMAs (the single-buffer kernel this
//     replaces waited 71 % of its cycles);
//   * a workgroup walks `items_per_wg` consecutive (sample, tile) items of one 32-channel chunk,
//     so even the 32x32 stage-3 maps give every CU a long walk and one flush.
constexpr int DW_LOADERS = 256;
constexpr int DW_LB = 4;  // raw vectors in flight per loader thread and batch

template <typename T, int S, int TH, int TW>
__global__ __launch_bounds__(TH * TW * DW_CV * 3 + DW_LOADERS) void dw_wgrad_kernel(
    const T* __restrict__ t1, const T* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const T* __restrict__ a,
    const float* __restrict__ ss_a, float* __restrict__ dw, const DwGeom g, const int items_per_wg,
    const int nbuf) {
  typedef typename LdsStore<T>::type L;
  typedef typename RawD<T>::type Raw;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int NPIX = TH * TW;
  constexpr int NCOMP = NPIX * DW_CV * 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [27][32] workgroup accumulator
  L* bufs = reinterpret_cast<L*>(red + 27 * 32);
  const int a_elems = g.T * IH * IW * 32, d_elems = g.T * NPIX * 32;
  const int buf_elems = a_elems + d_elems;                      // [T][IH][IW][32] relu(bn(a)) | [T][TH][TW][32] db

  const int tid = threadIdx.x;
  const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int n_wg_items = (g.B * ntiles + items_per_wg - 1) / items_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), n_wg_items);
  if (co.group < 0) return;
  const int c0 = co.chunk * DW_CV * 8;
  const int item0 = co.group * items_per_wg;
  int item1 = item0 + items_per_wg;
  if (item1 > g.B * ntiles) item1 = g.B * ntiles;
  const int nit = item1 - item0;
  for (int i = tid; i < 27 * 32; i += NCOMP + DW_LOADERS) red[i] = 0.f;

  if (tid >= NCOMP) {
    // ------------------------------- producer waves -------------------------------------------
    const int lt = tid - NCOMP;
    const int cv = lt % DW_CV;
    const int cbase = c0 + cv * 8;
    const bool c_ok = cbase < g.Cp;
    float sa[8], sb[8], cA[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sa[j] = c_ok ? ss_a[cbase + j] : 0.f; sb[j] = c_ok ? ss_a[g.Cp + cbase + j] : 0.f;
      cA[j] = c_ok ? coefA[cbase + j] : 0.f; cC[j] = c_ok ? coefC[cbase + j] : 0.f;
    }
    const int NA = g.T * IH * IW * DW_CV, ND = g.T * NPIX * DW_CV;
    auto stage = [&](const int item, L* __restrict__ atile) {
      L* __restrict__ dtile = atile + a_elems;
      const int b = item / ntiles, tl = item - b * ntiles;
      const int tx = tl % tiles_x, ty = tl / tiles_x;
      const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
      float cB[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cB[j] = c_ok ? coefB[(size_t)b * g.Cp + cbase + j] : 0.f;
      for (int base = lt; base < NA; base += DW_LOADERS * DW_LB) {
        Raw raw[DW_LB];
        int pp[DW_LB];
        bool ok[DW_LB];
#pragma unroll
        for (int u = 0; u < DW_LB; ++u) {
          const int i = base + u * DW_LOADERS;
          const int p = i / DW_CV;
          const int ix = p % IW;
          const int q = p / IW;
          const int iy = q % IH, t = q / IH;
          const int gy = iy0 + iy, gx = ix0 + ix;
          pp[u] = i < NA ? p : -1;
          ok[u] = i < NA && c_ok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
          if (ok[u]) raw[u] = RawD<T>::load(a + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * g.Cp + cbase);
        }
#pragma unroll
        for (int u = 0; u < DW_LB; ++u) {
          if (pp[u] < 0) continue;
          float f[8];
          if (ok[u]) {
            RawD<T>::cvt(raw[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sa[j], sb[j]), 0.f);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
          Vec8<L>::store(atile + (size_t)pp[u] * 32 + cv * 8, f);
          __builtin_amdgcn_sched_barrier(0);  // convert one vector at a time (register budget)
        }
      }
      constexpr int DB = DW_LB / 2;
      for (int base = lt; base < ND; base += DW_LOADERS * DB) {
        Raw r1[DB], r2[DB];
        int pp[DB];
        bool ok[DB];
#pragma unroll
        for (int u = 0; u < DB; ++u) {
          const int i = base + u * DW_LOADERS;
          const int p = i / DW_CV;
          const int ox = p % TW;
          const int q = p / TW;
          const int oy = q % TH, t = q / TH;
          const int gy = ty * TH + oy, gx = tx * TW + ox;
          pp[u] = i < ND ? p : -1;
          ok[u] = i < ND && c_ok && gy < g.Ho && gx < g.Wo;
          if (ok[u]) {
            const size_t off = ((((size_t)b * g.T + t) * g.Ho + gy) * g.Wo + gx) * g.Cp + cbase;
            r1[u] = RawD<T>::load(t1 + off);
            r2[u] = RawD<T>::load(bb + off);
          }
        }
#pragma unroll
        for (int u = 0; u < DB; ++u) {
          if (pp[u] < 0) continue;
          float f[8];
          if (ok[u]) {
            float f2[8];
            RawD<T>::cvt(r1[u], f);
            RawD<T>::cvt(r2[u], f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
          Vec8<L>::store(dtile + (size_t)pp[u] * 32 + cv * 8, f);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (nit > 0) stage(item0, bufs);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      if (nbuf == 2) {
        if (it + 1 < nit) stage(item0 + it + 1, bufs + (size_t)((it + 1) & 1) * buf_elems);
        __syncthreads();
      } else {
        __syncthreads();                                   // consumers done with the only buffer
        if (it + 1 < nit) stage(item0 + it + 1, bufs);
        __syncthreads();
      }
    }
    __syncthreads();                                       // matches the consumers' flush barrier
  } else {
    // ------------------------------- consumer waves -------------------------------------------
    const int cv = tid % DW_CV;
    const int pix = (tid / DW_CV) % NPIX;
    const int kt = tid / (DW_CV * NPIX);
    const int px = pix % TW, py = pix / TW;
    f32x2_t acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] = f32x2_t{0.f, 0.f};
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const L* __restrict__ atile = bufs + (size_t)(nbuf == 2 ? (it & 1) : 0) * buf_elems;
      const L* __restrict__ dtile = atile + a_elems;
      wgrad_tile_taps<L, S, TH, TW>(acc, atile, dtile, g.T, kt, px, py, cv);
      __syncthreads();
      if (nbuf != 2) __syncthreads();
    }
    wgrad_reduce_to_lds(acc, red, kt, tid & 63);
    __syncthreads();
    for (int i = tid; i < 27 * 32; i += NCOMP) {
      const int tap = i / 32, c = c0 + (i & 31);
      if (c < g.C) atomicAdd(dw + (size_t)c * 27 + tap, red[i]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Weight gradient, bf16, LDS-DMA producer (gfx950 global_load_lds_dwordx4).
// The register-staged producer above is latency bound: 256 loader threads can keep only ~16 KB
// in flight, one tile costs ~7 us to land and the consumers need ~2.7 us to eat it.  Here the
// loader waves DMA the RAW rows of tile i+2 straight into one of three LDS slots (no staging
// registers, a whole 43 KB tile in flight per CU), then convert tile i+1 IN PLACE
// (raw a -> relu(bn(a)), raw (t1, b) -> db written over t1) while the consumers run tile i.
//   * a wave-instruction lands 64 vectors = 1 KB contiguously, so the raw image IS the
//     [pixel][32 channels] tile layout the consumers read; halo pixels outside the image are
//     loaded from a clamped address and overwritten with zeros by the in-place pass;
//   * every loader wave issues exactly NI DMA instructions per tile (surplus ones land in a sink
//     KB), so "tile i+1 has landed" is the counted  s_waitcnt vmcnt(NI)  with tile i+2 in flight;
//   * barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() fence would drain the DMA queue;
//     the loader's LDS traffic is inline asm for the same reason (the compiler would put
//     vmcnt(0) in front of any LDS access it can see while a DMA is pending).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void wg_barrier_raw() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int S, int TH, int TW, int TT> struct WgDma {
  static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, NPIX = TH * TW;
  static constexpr int NCOMP = NPIX * DW_CV * 3;
  static constexpr int NLW = DW_LOADERS / 64;                                  // loader waves
  static constexpr int NA_V = TT * IH * IW * DW_CV, ND_V = TT * NPIX * DW_CV;  // vectors per tile
  static constexpr int NA_I = (NA_V + 63) / 64, ND_I = (ND_V + 63) / 64;       // wave-instructions
  static constexpr int NA_W = (NA_I + NLW - 1) / NLW, ND_W = (ND_I + NLW - 1) / NLW;
  static constexpr int NI = NA_W + 2 * ND_W;                                   // DMA instr / wave / tile
  static constexpr int A_BYTES = NA_I * 1024, D_BYTES = ND_I * 1024;
  static constexpr int SINK_OFF = A_BYTES + 2 * D_BYTES;
  static constexpr int SLOT_BYTES = SINK_OFF + 1024;
  static constexpr int MAXB = 8;  // samples one workgroup walk may touch (their coefB rows sit in LDS)
  static constexpr int FIXED_BYTES = (27 * 32 + MAXB * 32) * 4;
  // three slots (two tiles in flight ahead of the consumers) when they fit, else two (T = 5: 73 KB slots)
  static constexpr int NSLOT = FIXED_BYTES + 3 * SLOT_BYTES <= 160 * 1024 ? 3 : 2;
  static constexpr int LDS_BYTES = FIXED_BYTES + NSLOT * SLOT_BYTES;
  static_assert(NI <= 63, "vmcnt is a 6-bit counter");
};

template <int S, int TH, int TW, int TT>
__global__ __launch_bounds__(TH * TW * DW_CV * 3 + DW_LOADERS) void dw_wgrad_dma_kernel(
    const bf16_t* __restrict__ t1, const bf16_t* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const bf16_t* __restrict__ a,
    const float* __restrict__ ss_a, float* __restrict__ dw, const DwGeom g, const int items_per_wg) {
  typedef WgDma<S, TH, TW, TT> G;
  constexpr int IH = G::IH, IW = G::IW, NPIX = G::NPIX, NCOMP = G::NCOMP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [27][32] workgroup accumulator
  float* cbs = red + 27 * 32;                    // [MAXB][32] per-sample coefB rows of this chunk
  unsigned char* slots = smem + G::FIXED_BYTES;

  const int tid = threadIdx.x;
  const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int n_wg_items = (g.B * ntiles + items_per_wg - 1) / items_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), n_wg_items);
  if (co.group < 0) return;
  const int c0 = co.chunk * DW_CV * 8;
  const int item0 = co.group * items_per_wg;
  int item1 = item0 + items_per_wg;
  if (item1 > g.B * ntiles) item1 = g.B * ntiles;
  const int nit = item1 - item0;
  const int b0 = item0 / ntiles;
  const int nb = nit > 0 ? (item1 - 1) / ntiles - b0 + 1 : 0;
  for (int i = tid; i < 27 * 32; i += NCOMP + DW_LOADERS) red[i] = 0.f;
  for (int i = tid; i < nb * 32; i += NCOMP + DW_LOADERS) {
    const int c = c0 + (i & 31);
    cbs[i] = c < g.Cp ? coefB[(size_t)(b0 + (i >> 5)) * g.Cp + c] : 0.f;
  }
  wg_barrier_raw();

  if (tid >= NCOMP) {
    // ------------------------------- producer waves -------------------------------------------
    const int lane = tid & 63;
    const int lw = (tid - NCOMP) >> 6;
    const int cv = lane & (DW_CV - 1);            // vector index = instr*64 + lane, so cv = lane % 4
    const int cbase = c0 + cv * 8;
    const bool c_ok = cbase < g.Cp;
    const int cb_ld = c_ok ? cbase : c0;          // clamped channel offset for the DMA source
    float sa[8], sb[8], cA[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sa[j] = c_ok ? ss_a[cbase + j] : 0.f; sb[j] = c_ok ? ss_a[g.Cp + cbase + j] : 0.f;
      cA[j] = c_ok ? coefA[cbase + j] : 0.f; cC[j] = c_ok ? coefC[cbase + j] : 0.f;
    }
    float cB[8];
    int cur_b = -1;

    auto issue = [&](const int item, const int slot) {
      unsigned char* sl = slots + slot * G::SLOT_BYTES;
      const int b = item / ntiles, tl = item - b * ntiles;
      const int tx = tl % tiles_x, ty = tl / tiles_x;
      const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
#pragma unroll
      for (int r = 0; r < G::NA_W; ++r) {
        const int q = lw + G::NLW * r;
        unsigned char* dst = sl + (q < G::NA_I ? q * 1024 : G::SINK_OFF);
        int i = q * 64 + lane;
        if (i > G::NA_V - 1) i = G::NA_V - 1;
        const int p = i / DW_CV;
        const int ix = p % IW;
        const int qq = p / IW;
        const int iy = qq % IH, t = qq / IH;
        int gy = iy0 + iy, gx = ix0 + ix;
        gy = gy < 0 ? 0 : (gy > g.H - 1 ? g.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx > g.W - 1 ? g.W - 1 : gx);
        const bf16_t* src = a + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * g.Cp + cb_ld;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < G::ND_W; ++r) {
        const int q = lw + G::NLW * r;
        unsigned char* dst = sl + (q < G::ND_I ? G::A_BYTES + q * 1024 : G::SINK_OFF);
        unsigned char* dst2 = sl + (q < G::ND_I ? G::A_BYTES + G::D_BYTES + q * 1024 : G::SINK_OFF);
        int i = q * 64 + lane;
        if (i > G::ND_V - 1) i = G::ND_V - 1;
        const int p = i / DW_CV;
        const int ox = p % TW;
        const int qq = p / TW;
        const int oy = qq % TH, t = qq / TH;
        int gy = ty * TH + oy, gx = tx * TW + ox;
        gy = gy > g.Ho - 1 ? g.Ho - 1 : gy;
        gx = gx > g.Wo - 1 ? g.Wo - 1 : gx;
        const size_t off = ((((size_t)b * g.T + t) * g.Ho + gy) * g.Wo + gx) * g.Cp + cb_ld;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(t1 + off), (lds_ptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(bb + off), (lds_ptr_t)dst2, 16, 0, 0);
      }
    };

    auto convert = [&](const int item, const int slot) {
      const uint32_t sl = lds_addr(slots + slot * G::SLOT_BYTES);
      const int b = item / ntiles, tl = item - b * ntiles;
      const int tx = tl % tiles_x, ty = tl / tiles_x;
      const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
      if (b != cur_b) {  // wave-uniform
        LdsVec<float>::raw_t r;
        LdsVec<float>::issue(r, lds_addr(cbs + (b - b0) * 32 + cv * 8));
        LdsVec<float>::template wait<0>(r);
        LdsVec<float>::cvt(r, cB);
        cur_b = b;
      }
#pragma unroll
      for (int r = 0; r < G::NA_W; ++r) {
        const int q = lw + G::NLW * r;
        const int i = q * 64 + lane;
        if (q < G::NA_I && i < G::NA_V) {
          const int p = i / DW_CV;
          const int ix = p % IW;
          const int qq = p / IW;
          const int iy = qq % IH;
          const int gy = iy0 + iy, gx = ix0 + ix;
          const uint32_t addr = sl + (uint32_t)i * 16u;
          u32x4_t out = {0u, 0u, 0u, 0u};
          if (c_ok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
            LdsVec<bf16_t>::raw_t raw;
            LdsVec<bf16_t>::issue(raw, addr);
            LdsVec<bf16_t>::template wait<0>(raw);
            float f[8];
            LdsVec<bf16_t>::cvt(raw, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sa[j], sb[j]), 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
          }
          asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(out) : "memory");
        }
      }
#pragma unroll
      for (int r = 0; r < G::ND_W; ++r) {
        const int q = lw + G::NLW * r;
        const int i = q * 64 + lane;
        if (q < G::ND_I && i < G::ND_V) {
          const int p = i / DW_CV;
          const int ox = p % TW;
          const int oy = (p / TW) % TH;
          const int gy = ty * TH + oy, gx = tx * TW + ox;
          const uint32_t addr = sl + (uint32_t)(G::A_BYTES + i * 16);
          u32x4_t out = {0u, 0u, 0u, 0u};
          if (c_ok && gy < g.Ho && gx < g.Wo) {
            LdsVec<bf16_t>::raw_t r1, r2;
            LdsVec<bf16_t>::issue(r1, addr);
            LdsVec<bf16_t>::issue(r2, addr + (uint32_t)G::D_BYTES);
            LdsVec<bf16_t>::template wait<0>(r2);
            asm volatile("" : "+v"(r1.v));  // r1 is complete as well: LDS returns in order
            float f[8], f2[8];
            LdsVec<bf16_t>::cvt(r1, f);
            LdsVec<bf16_t>::cvt(r2, f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
          }
          asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(out) : "memory");
        }
      }
    };

    constexpr int AH = G::NSLOT - 1;   // tiles whose DMA is issued ahead of the tile being consumed
    if (nit > 0) issue(item0, 0);
    if (AH > 1 && nit > 1) issue(item0 + 1, 1);
    if (nit > 0) {
      if (AH > 1 && nit > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      convert(item0, 0);
    }
    wg_barrier_raw();
    for (int it = 0; it < nit; ++it) {
      // slot (it + AH) % NSLOT was consumed in pass it - 1
      if (it + AH < nit) issue(item0 + it + AH, (it + AH) % G::NSLOT);
      if (it + 1 < nit) {
        if (AH > 1 && it + AH < nit) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert(item0 + it + 1, (it + 1) % G::NSLOT);
      }
      wg_barrier_raw();
    }
    wg_barrier_raw();  // matches the consumers' flush barrier
  } else {
    // ------------------------------- consumer waves -------------------------------------------
    const int cv = tid % DW_CV;
    const int pix = (tid / DW_CV) % NPIX;
    const int kt = tid / (DW_CV * NPIX);
    const int px = pix % TW, py = pix / TW;
    f32x2_t acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] = f32x2_t{0.f, 0.f};
    wg_barrier_raw();
    for (int it = 0; it < nit; ++it) {
      const bf16_t* atile = reinterpret_cast<const bf16_t*>(slots + (it % G::NSLOT) * G::SLOT_BYTES);
      const bf16_t* dtile = reinterpret_cast<const bf16_t*>(slots + (it % G::NSLOT) * G::SLOT_BYTES + G::A_BYTES);
      wgrad_tile_taps<bf16_t, S, TH, TW>(acc, atile, dtile, TT, kt, px, py, cv);
      wg_barrier_raw();
    }
    wgrad_reduce_to_lds(acc, red, kt, tid & 63);
    wg_barrier_raw();
    for (int i = tid; i < 27 * 32; i += NCOMP) {
      const int tap = i / 32, c = c0 + (i & 31);
      if (c < g.C) atomicAdd(dw + (size_t)c * 27 + tap, red[i]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Weight gradient, bf16, stride 1, T = 3: LDS-DMA producer + v_dot2c consumer.
// In the kernel above 53 % of the consumer's VALU time is bf16->f32 unpacking (integer ops run at half
// the f32 FMA rate).  The reduction of this kernel runs over PIXELS, so two output pixels (y, y+4) of a
// tile are paired: with (db[y], db[y+4]) and (a[r], a[r+4]) packed per channel as bf16 pairs,
//     acc[tap][c] = v_dot2c_f32_bf16(acc, dpair[c], apair[c])      -- two MACs, no conversion --
// and the consumer needs half the threads (6 waves).  The DMA makes the pairing free: lane L < 32 of a
// wave-instruction loads the vector of row r, lane L + 32 the vector of row r + 4 of the SAME item, so
// both partners are landed by one wave (its own vmcnt covers them) 512 B apart; the in-place pass
// converts 4 channels of both rows per lane and writes the packed pairs over its own raw vector.
// Item = (frame, row r < 6 | y < 4, column, channel vector); an item occupies 16 B in each half
// of its wave-instruction's KB:  [instr][half][32 items] -- consecutive items are 16 B apart, so the
// consumer's two b128 reads per operand are conflict free.
constexpr int D2_TH = 8, D2_TW = 8, D2_IW = D2_TW + 2, D2_AR = D2_TH / 2 + 2, D2_DR = D2_TH / 2;   // 6 a-rows, 4 d-rows
constexpr int D2_LOADERS = 512;   // 8 producer waves: with 6 consumer waves the in-place conversion is the longer job
template <int TT> struct WgDot2 {
  static constexpr int NCOMP = D2_DR * D2_TW * DW_CV * 3;                 // 384 consumer threads
  static constexpr int NLW = D2_LOADERS / 64;
  static constexpr int NA_IT = TT * D2_AR * D2_IW * DW_CV, ND_IT = TT * D2_DR * D2_TW * DW_CV;   // items
  static constexpr int NA_I = (NA_IT + 31) / 32, ND_I = (ND_IT + 31) / 32;                      // wave-instructions
  static constexpr int NA_W = (NA_I + NLW - 1) / NLW, ND_W = (ND_I + NLW - 1) / NLW;
  static constexpr int NI = NA_W + 2 * ND_W;
  static constexpr int A_BYTES = NA_I * 1024, D_BYTES = ND_I * 1024;
  static constexpr int SINK_OFF = A_BYTES + 2 * D_BYTES;
  static constexpr int SLOT_BYTES = SINK_OFF + 1024;
  static constexpr int NSLOT = 3;
  static constexpr int MAXB = 8;
  static constexpr int FIXED_BYTES = (27 * 32 + MAXB * 32) * 4;
  static constexpr int LDS_BYTES = FIXED_BYTES + NSLOT * SLOT_BYTES;
  static_assert(NI <= 63, "vmcnt is a 6-bit counter");
};

__device__ __forceinline__ uint32_t d2_item_off(const int item) {   // byte offset of an item's half 0 in its region
  return (uint32_t)((item >> 5) * 1024 + (item & 31) * 16);
}

template <int TT>
__global__ __launch_bounds__(WgDot2<TT>::NCOMP + D2_LOADERS) void dw_wgrad_dot2_kernel(
    const bf16_t* __restrict__ t1, const bf16_t* __restrict__ bb, const float* __restrict__ coefA,
    const float* __restrict__ coefB, const float* __restrict__ coefC, const bf16_t* __restrict__ a,
    const float* __restrict__ ss_a, float* __restrict__ dw, const DwGeom g, const int items_per_wg) {
  typedef WgDot2<TT> G;
  constexpr int NCOMP = G::NCOMP;
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [27][32] workgroup accumulator
  float* cbs = red + 27 * 32;                    // [MAXB][32] per-sample coefB rows of this chunk
  unsigned char* slots = smem + G::FIXED_BYTES;

  const int tid = threadIdx.x;
  const int tiles_x = (g.Wo + D2_TW - 1) / D2_TW, tiles_y = (g.Ho + D2_TH - 1) / D2_TH;
  const int ntiles = tiles_x * tiles_y;
  const int n_wg_items = (g.B * ntiles + items_per_wg - 1) / items_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), n_wg_items);
  if (co.group < 0) return;
  const int c0 = co.chunk * DW_CV * 8;
  const int item0 = co.group * items_per_wg;
  int item1 = item0 + items_per_wg;
  if (item1 > g.B * ntiles) item1 = g.B * ntiles;
  const int nit = item1 - item0;
  const int b0 = item0 / ntiles;
  const int nb = nit > 0 ? (item1 - 1) / ntiles - b0 + 1 : 0;
  for (int i = tid; i < 27 * 32; i += NCOMP + D2_LOADERS) red[i] = 0.f;
  for (int i = tid; i < nb * 32; i += NCOMP + D2_LOADERS) {
    const int c = c0 + (i & 31);
    cbs[i] = c < g.Cp ? coefB[(size_t)(b0 + (i >> 5)) * g.Cp + c] : 0.f;
  }
  wg_barrier_raw();

  if (tid >= NCOMP) {
    // ------------------------------- producer waves -------------------------------------------
    const int lane = tid & 63;
    const int lw = (tid - NCOMP) >> 6;
    const int hl = lane >> 5, li = lane & 31;     // half (row r / row r+4 partner), item within the instruction
    const int cv = li & (DW_CV - 1);              // item = instr*32 + li, so cv = li % 4
    const int cbase = c0 + cv * 8;
    const bool c_ok = cbase < g.Cp;
    const int cb_ld = c_ok ? cbase : c0;
    const int ch0 = hl * 4;                       // this lane converts channels [ch0, ch0+4) of both rows
    float sa[4], sb[4], cA[4], cC[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sa[j] = c_ok ? ss_a[cbase + ch0 + j] : 0.f; sb[j] = c_ok ? ss_a[g.Cp + cbase + ch0 + j] : 0.f;
      cA[j] = c_ok ? coefA[cbase + ch0 + j] : 0.f; cC[j] = c_ok ? coefC[cbase + ch0 + j] : 0.f;
    }
    float cB[4];
    int cur_b = -1;

    // Tile-independent part of this lane's units, decoded ONCE: pixel offset relative to the tile origin
    // (frame, row incl. the +4 of the partner half, column) and a packed (row, column, real) word for
    // the conversion's in-image tests.  Per tile the DMA source is then one add and ONE clamp of the
    // pixel index: halo positions outside the image may be fetched from any valid address, the
    // conversion overwrites them with zeros.  (Decoding per tile cost ~40 integer VALU per unit and the
    // eight producer waves, not the six consumer waves, set the pace.)
    int relA[G::NA_W], codeA[G::NA_W], relD[G::ND_W], codeD[G::ND_W];
#pragma unroll
    for (int r_ = 0; r_ < G::NA_W; ++r_) {
      const int q = lw + G::NLW * r_;
      int it = q * 32 + li;
      const bool real = q < G::NA_I && it < G::NA_IT;
      if (it > G::NA_IT - 1) it = G::NA_IT - 1;
      const int p = it / DW_CV;
      const int ix = p % D2_IW, qq = p / D2_IW;
      const int rr = qq % D2_AR, t = qq / D2_AR;
      relA[r_] = (t * g.H + rr + 4 * hl) * g.W + ix;
      codeA[r_] = rr | (ix << 8) | (real ? (1 << 30) : 0);
    }
#pragma unroll
    for (int r_ = 0; r_ < G::ND_W; ++r_) {
      const int q = lw + G::NLW * r_;
      int it = q * 32 + li;
      const bool real = q < G::ND_I && it < G::ND_IT;
      if (it > G::ND_IT - 1) it = G::ND_IT - 1;
      const int p = it / DW_CV;
      const int ox = p % D2_TW, qq = p / D2_TW;
      const int yy = qq % D2_DR, t = qq / D2_DR;
      relD[r_] = (t * g.Ho + yy + 4 * hl) * g.Wo + ox;
      codeD[r_] = yy | (ox << 8) | (real ? (1 << 30) : 0);
    }
    const int npix_a = g.B * g.T * g.H * g.W - 1, npix_d = g.B * g.T * g.Ho * g.Wo - 1;

    auto issue = [&](const int item, const int slot) {
      unsigned char* sl = slots + slot * G::SLOT_BYTES;
      const int b = item / ntiles, tl = item - b * ntiles;
      const int tx = tl % tiles_x, ty = tl / tiles_x;
      const int oA = (b * g.T * g.H + ty * D2_TH - 1) * g.W + tx * D2_TW - 1;
      const int oD = (b * g.T * g.Ho + ty * D2_TH) * g.Wo + tx * D2_TW;
#pragma unroll
      for (int r_ = 0; r_ < G::NA_W; ++r_) {
        const int q = lw + G::NLW * r_;
        unsigned char* dst = sl + (q < G::NA_I ? q * 1024 : G::SINK_OFF);
        int rel = relA[r_];
        asm volatile("" : "+v"(rel));
        int pix = oA + rel;
        pix = pix < 0 ? 0 : (pix > npix_a ? npix_a : pix);
        const bf16_t* src = a + (size_t)pix * g.Cp + cb_ld;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
      }
#pragma unroll
      for (int r_ = 0; r_ < G::ND_W; ++r_) {
        const int q = lw + G::NLW * r_;
        unsigned char* dst = sl + (q < G::ND_I ? G::A_BYTES + q * 1024 : G::SINK_OFF);
        unsigned char* dst2 = sl + (q < G::ND_I ? G::A_BYTES + G::D_BYTES + q * 1024 : G::SINK_OFF);
        int rel = relD[r_];
        asm volatile("" : "+v"(rel));
        int pix = oD + rel;
        pix = pix < 0 ? 0 : (pix > npix_d ? npix_d : pix);
        const size_t off = (size_t)pix * g.Cp + cb_ld;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(t1 + off), (lds_ptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(bb + off), (lds_ptr_t)dst2, 16, 0, 0);
      }
    };

    // 4 channels [ch0, ch0+4) of a raw 8-channel bf16 vector
    auto up4 = [&](const u32x2_t v, float (&f)[4]) {
      f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
      f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
    };

    auto convert = [&](const int item, const int slot) {
      const uint32_t sl = lds_addr(slots + slot * G::SLOT_BYTES);
      const int b = item / ntiles, tl = item - b * ntiles;
      const int tx = tl % tiles_x, ty = tl / tiles_x;
      const int iy0 = ty * D2_TH - 1, ix0 = tx * D2_TW - 1;
      if (b != cur_b) {  // wave-uniform
        LdsVec<float>::raw_t r;
        LdsVec<float>::issue(r, lds_addr(cbs + (b - b0) * 32 + cv * 8));
        LdsVec<float>::template wait<0>(r);
        float c8[8];
        LdsVec<float>::cvt(r, c8);
#pragma unroll
        for (int j = 0; j < 4; ++j) cB[j] = hl ? c8[4 + j] : c8[j];
        cur_b = b;
      }
#pragma unroll
      for (int r_ = 0; r_ < G::NA_W; ++r_) {
        const int q = lw + G::NLW * r_;
        int cd = codeA[r_];
        asm volatile("" : "+v"(cd));
        if (cd & (1 << 30)) {
          const int rr = cd & 255, ix = (cd >> 8) & 255;
          const int gx = ix0 + ix, gy0 = iy0 + rr, gy1 = gy0 + 4;
          const bool okx = c_ok && gx >= 0 && gx < g.W;
          const bool ok0 = okx && gy0 >= 0 && gy0 < g.H, ok1 = okx && gy1 >= 0 && gy1 < g.H;
          const uint32_t base = sl + (uint32_t)(q * 1024 + li * 16);   // half 0 (row r); half 1 (row r+4) at +512
          u32x2_t w0, w1;   // row r, row r+4 (both landed by this wave)
          asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1) : "v"(base + (uint32_t)(ch0 * 2)));
          float f0[4], f1[4];
          up4(w0, f0);
          up4(w1, f1);
          u32x4_t out;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            out[j] = pack_bf16x2(ok0 ? fmaxf(fmaf(f0[j], sa[j], sb[j]), 0.f) : 0.f,
                                 ok1 ? fmaxf(fmaf(f1[j], sa[j], sb[j]), 0.f) : 0.f);
          asm volatile("ds_write_b128 %0, %1" : : "v"(base + (uint32_t)(hl * 512)), "v"(out) : "memory");
        }
      }
#pragma unroll
      for (int r_ = 0; r_ < G::ND_W; ++r_) {
        const int q = lw + G::NLW * r_;
        int cd = codeD[r_];
        asm volatile("" : "+v"(cd));
        if (cd & (1 << 30)) {
          const int yy = cd & 255, ox = (cd >> 8) & 255;
          const int gx = tx * D2_TW + ox, gy0 = ty * D2_TH + yy, gy1 = gy0 + 4;
          const bool okx = c_ok && gx < g.Wo;
          const bool ok0 = okx && gy0 < g.Ho, ok1 = okx && gy1 < g.Ho;
          const uint32_t base = sl + (uint32_t)(G::A_BYTES + q * 1024 + li * 16);
          u32x2_t w0, w1, w2, w3;   // t1 row y, t1 row y+4, b row y, b row y+4
          asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %5\n\t"
                       "ds_read_b64 %3, %5 offset:512\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                       : "v"(base + (uint32_t)(ch0 * 2)), "v"(base + (uint32_t)(G::D_BYTES + ch0 * 2)));
          float u0[4], u1[4], v0[4], v1[4];
          up4(w0, u0);
          up4(w1, u1);
          up4(w2, v0);
          up4(w3, v1);
          u32x4_t out;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            out[j] = pack_bf16x2(ok0 ? fmaf(cA[j], u0[j], fmaf(cC[j], v0[j], cB[j])) : 0.f,
                                 ok1 ? fmaf(cA[j], u1[j], fmaf(cC[j], v1[j], cB[j])) : 0.f);
          asm volatile("ds_write_b128 %0, %1" : : "v"(base + (uint32_t)(hl * 512)), "v"(out) : "memory");
        }
      }
    };

    if (nit > 0) issue(item0, 0);
    if (nit > 1) issue(item0 + 1, 1);
    if (nit > 0) {
      if (nit > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      convert(item0, 0);
    }
    wg_barrier_raw();
    for (int it = 0; it < nit; ++it) {
      if (it + 2 < nit) issue(item0 + it + 2, (it + 2) % G::NSLOT);
      if (it + 1 < nit) {
        if (it + 2 < nit) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert(item0 + it + 1, (it + 1) % G::NSLOT);
      }
      wg_barrier_raw();
    }
    wg_barrier_raw();  // matches the consumers' flush barrier
  } else {
    // ------------------------------- consumer waves -------------------------------------------
    // thread = (pixel pair (y, y+4; x), channel vector, temporal tap kt); acc[ky*3+kx][channel]
    const int cv = tid % DW_CV;
    const int pp = (tid / DW_CV) % (D2_DR * D2_TW);
    const int kt = tid / (DW_CV * D2_DR * D2_TW);
    const int px = pp % D2_TW, py = pp / D2_TW;
    float acc[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    wg_barrier_raw();
    for (int it = 0; it < nit; ++it) {
      const uint32_t sl = lds_addr(slots + (it % G::NSLOT) * G::SLOT_BYTES);
#pragma unroll 1
      for (int to = 0; to < TT; ++to) {
        const int ti = to + kt - 1;
        if (ti < 0 || ti >= TT) continue;
        const uint32_t d_addr = sl + (uint32_t)G::A_BYTES + d2_item_off(((to * D2_DR + py) * D2_TW + px) * DW_CV + cv);
        u32x4_t d0, d1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(d0), "=&v"(d1) : "v"(d_addr));
        u32x4_t r0[2], r1[2];
        auto a_addr = [&](const int k) {
          return sl + d2_item_off(((ti * D2_AR + py + k / 3) * D2_IW + px + k % 3) * DW_CV + cv);
        };
        {
          const uint32_t ad = a_addr(0);
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512" : "=&v"(r0[0]), "=&v"(r1[0]) : "v"(ad));
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (k < 8) {
            const uint32_t ad = a_addr(k + 1);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512"
                         : "=&v"(r0[(k + 1) & 1]), "=&v"(r1[(k + 1) & 1]) : "v"(ad));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r0[k & 1]), "+v"(r1[k & 1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0[k & 1]), "+v"(r1[k & 1]));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (copy the elements out first: __builtin_bit_cast of a vector-element lvalue reads element 0)
            const uint32_t dl = d0[j], dh = d1[j], al = r0[k & 1][j], ah = r1[k & 1][j];
            acc[k][j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, dl), __builtin_bit_cast(bf2_t, al),
                                                        acc[k][j], false);
            acc[k][4 + j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, dh), __builtin_bit_cast(bf2_t, ah),
                                                            acc[k][4 + j], false);
          }
          asm volatile("" : "+v"(acc[k][0]), "+v"(acc[k][1]), "+v"(acc[k][2]), "+v"(acc[k][3]),
                       "+v"(acc[k][4]), "+v"(acc[k][5]), "+v"(acc[k][6]), "+v"(acc[k][7]));
        }
      }
      wg_barrier_raw();
    }
    // reduce across the pixel pairs of each wave (lanes with equal cv; kt is wave-uniform: 128 threads per kt)
    const int lane = tid & 63;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[k][j];
#pragma unroll
        for (int o = DW_CV; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (lane < DW_CV) atomicAdd(&red[(kt * 9 + k) * 32 + lane * 8 + j], v);
      }
    }
    wg_barrier_raw();
    for (int i = tid; i < 27 * 32; i += NCOMP) {
      const int tap = i / 32, c = c0 + (i & 31);
      if (c < g.C) atomicAdd(dw + (size_t)c * 27 + tap, red[i]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Forward v2 (stride 1).  Mapping chosen from the round-1 profile (v1 was LDS-read bound at
// 3.7 FMA per LDS read):
//   * wave  = one 8-channel vector; the LDS tile is stored as per-vector planes [cv][t][y][x][8] so a
//     wave's lanes read consecutive 16-byte pixels (conflict-free) and weights are wave-uniform
//     (LDS broadcast reads);
//   * lane  = a 2-pixel strip along x; the 4 input pixels of a kernel row stay in registers for all
//     kx / kt taps and all T frames (about 11 FMA per LDS read);
//   * workgroup = 8x16 output pixels x 32 channels, walks `tiles_per_wg` tiles of one sample keeping
//     the per-(sample,channel) statistics in registers, with the next tile's raw rows prefetched.
// PYR = rows per lane: 2 (8 x 16 tiles) for three frames; 1 (4 x 16 tiles) for five frames, whose 8 x 16 tile is 115 KB of
// LDS = one workgroup per CU (1.3 TB/s); the 4 x 16 tile is 69 KB = two.
constexpr int V2_TW = 16, V2_IW = V2_TW + 2;

__device__ __forceinline__ void lds_ld8v2(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// LDS plane geometry of the v2 kernels: one plane per (channel vector, half vector) holding a float4
// per pixel, [t][iy][ix]; the plane stride is padded by one float4 so the 8 planes of a pixel fall on
// distinct bank groups (conflict-free staging writes), and lanes walk x so stencil reads are dense.
template <int TT> struct V2Geo {
  static constexpr int PYR = TT <= 3 ? 2 : 1;
  static constexpr int TH = 4 * PYR, IH = TH + 2;
  static constexpr int PLANE = TT * IH * V2_IW + 1;             // float4 units
  static constexpr int NI = TT * IH * V2_IW * DW_CV;            // staged 8-channel vectors per tile
  static constexpr int SL = (NI + 255) / 256;
};

template <typename T, int TT>
__global__ __launch_bounds__(256) void dw_fwd_v2_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                        const float* __restrict__ w, T* __restrict__ y,
                                                        double* __restrict__ nc, const DwGeom g,
                                                        const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef RawD<T> RW;
  typedef V2Geo<TT> G;
  constexpr int NI = G::NI, SL = G::SL, PLANE = G::PLANE, PYR = G::PYR, V2_TH = G::TH, V2_IH = G::IH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                  // [27][32]
  float* fss = wl + 27 * 32;                                   // [2][32] scale | shift of this chunk (fin.sums mode)
  float4* tile = reinterpret_cast<float4*>(fss + 64);          // [4 cv][2 halves][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wcv = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave's channel vector
  const int lx = lane & 15, yp = lane >> 4;                    // lane = column x, rows PYR*yp .. PYR*yp + PYR-1
  const int tiles_x = (g.W + V2_TW - 1) / V2_TW, tiles_y = (g.H + V2_TH - 1) / V2_TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx, tg = co.group % gx;
  const int c0 = chunk * DW_CV * 8;
  // staging role: item i = tid + 256*slot -> (cv = i & 3, pixel = i >> 2)
  const int scv = tid & 3;
  const int sbase = c0 + scv * 8;
  const bool s_ok = sbase < g.Cp;
  const int cbase = c0 + wcv * 8;       // compute role
  const bool c_ok = cbase < g.Cp;

  for (int i = tid; i < 27 * 32; i += 256) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  typename RW::type raw[SL];
  unsigned vmask = 0;
#define V2_ISSUE(TL)                                                                            \
  {                                                                                             \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const int i_ = tid + sl * 256;                                                            \
      const int p_ = i_ >> 2;                                                                   \
      const int ix_ = p_ % V2_IW, q_ = p_ / V2_IW;                                              \
      const int iy_ = q_ % V2_IH, t_ = q_ / V2_IH;                                              \
      const int gy_ = ty_ * V2_TH - 1 + iy_, gx_ = tx_ * V2_TW - 1 + ix_;                       \
      if (i_ < NI && s_ok && t_ < g.T && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) {     \
        raw[sl] = RW::load(x + ((((size_t)b * g.T + t_) * g.H + gy_) * g.W + gx_) * g.Cp + sbase); \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) V2_ISSUE(tl0)
  // BatchNorm scale / shift of this chunk: given, or rebuilt from the producer's completed sums while the first
  // tile's loads are in flight (csrc/bn_fin.h; the workgroup of (sample 0, walk 0) owns the chunk's global outputs)
  float sc[8], sh[8];
  if (fin.sums) {
    if (tid == 0 && co.chunk == 0 && co.group == 0 && fin.training && fin.nbt) *fin.nbt += 1;
    c3dfin::bn_consume(fin, g.C, g.Cp, c0, 32, co.group == 0, fss, fss + 32, tid, 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? fss[scv * 8 + j] : 0.f; sh[j] = s_ok ? fss[32 + scv * 8 + j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? ss[sbase + j] : 0.f; sh[j] = s_ok ? ss[g.Cp + sbase + j] : 0.f; }
  }
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i = tid + sl * 256;
      if (i < NI) {
        float f[8];
        if ((vmask >> sl) & 1u) {
          RW::cvt(raw[sl], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        const int p = i >> 2;  // = (t*IH + iy)*IW + ix
        tile[(scv * 2 + 0) * PLANE + p] = make_float4(f[0], f[1], f[2], f[3]);
        tile[(scv * 2 + 1) * PLANE + p] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
    if (tl + 1 < tl1) V2_ISSUE(tl + 1)
    __syncthreads();

    float acc[TT][PYR][8];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int py = 0; py < PYR; ++py)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][py][j] = 0.f;
    const float4* pl0 = tile + (wcv * 2 + 0) * PLANE;
    const float4* pl1 = tile + (wcv * 2 + 1) * PLANE;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        float wk[3][8];  // wave-uniform: broadcast LDS reads
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) lds_ld8v2(wl + (kt * 9 + ky * 3 + kx) * 32 + wcv * 8, wk[kt]);
#pragma unroll
        for (int ti = 0; ti < TT; ++ti) {
          float in[PYR][8];
#pragma unroll
          for (int py = 0; py < PYR; ++py) {
            const int p = (ti * V2_IH + PYR * yp + py + ky) * V2_IW + lx + kx;
            const float4 h0 = pl0[p], h1 = pl1[p];
            in[py][0] = h0.x; in[py][1] = h0.y; in[py][2] = h0.z; in[py][3] = h0.w;
            in[py][4] = h1.x; in[py][5] = h1.y; in[py][6] = h1.z; in[py][7] = h1.w;
          }
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT) {
#pragma unroll
              for (int py = 0; py < PYR; ++py)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[to][py][j] = fmaf(in[py][j], wk[kt][j], acc[to][py][j]);
            }
          }
        }
      }
    }
    const int ox = tx * V2_TW + lx;
#pragma unroll
    for (int py = 0; py < PYR; ++py) {
      const int oy = ty * V2_TH + PYR * yp + py;
      if (c_ok && oy < g.H && ox < g.W) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < g.T) {
            T* dst = y + ((((size_t)b * g.T + t) * g.H + oy) * g.W + ox) * g.Cp + cbase;
            Vec8<T>::store(dst, acc[t][py]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float r = round_as<T>(acc[t][py][j]);
              s1[j] += r; s2[j] = fmaf(r, r, s2[j]);
            }
          }
        }
      }
    }
  }
#undef V2_ISSUE
  if (nc == nullptr) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r1 = wave_sum(s1[j]), r2 = wave_sum(s2[j]);
    const int c = cbase + j;
    if (lane == 0 && c < g.C) {
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)r1);
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)r2);
    }
  }
}

template <typename T, int TT>
int launch_fwd_v2(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                  hipStream_t stream, const c3d_bn_fin* fin = nullptr) {
  const size_t lds = (27 * 32 + 64) * sizeof(float) + (size_t)DW_CV * 2 * V2Geo<TT>::PLANE * sizeof(float4);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2_kernel<T, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  constexpr int V2_TH = V2Geo<TT>::TH;
  const int ntiles = ((g.W + V2_TW - 1) / V2_TW) * ((g.H + V2_TH - 1) / V2_TH);
  int tpw = 16 * 2 / V2Geo<TT>::PYR;  // swept on MI355X: 1:427us 4:255 8:240 16:233 32:250 (stage-1 shape)
  // ...but a walk is a serial chain (~5.5 us per tile): keep ~2 workgroups per CU in the grid
  // (the 64x64 / 32x32 stages launched 256 / 224 workgroups of 16 / 8 tiles: one per CU, 98 / 52 us)
  const int chunks_ = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  // measured best: 16 / 8 / 4 tiles for the 128x128 / 64x64 / 32x32 stages = ~2 workgroups per CU, and
  // never fewer than 4 tiles (the prefetch pipeline needs a walk)
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks_ * g.B < 2L * device_cus()) tpw >>= 1;
  if (const char* e = c3d_env("C3D_DW_TPW")) tpw = atoi(e) > 0 ? atoi(e) : tpw;  // tuning knob
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), (long)((ntiles + tpw - 1) / tpw) * g.B));
  c3d_bn_fin f0;
  std::memset(&f0, 0, sizeof(f0));
  dw_fwd_v2_kernel<T, TT><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                            reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  C3D_CHECK_LAUNCH();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// Forward v2, stride 2 (the first block of every stage).  Same roles as v2 -- wave = one 8-channel vector with
// wave-uniform weights, lane = an output pixel, walking workgroup with the next tile's raw rows prefetched -- over a
// POLYPHASE tile: the 9 x 33 input pixels a 4 x 16 output tile reads are staged as four parity planes
// [t][iy & 1][ix & 1][iy >> 1][ix >> 1], so tap (ky, kx) of output (oy, ox) is plane (ky & 1, kx & 1) at
// (oy + (ky == 2), ox + (kx == 2)): the lanes of a wave (consecutive ox) read consecutive 16-byte pixels, as in the
// stride-1 kernel (reading every second pixel of a dense tile is a two-way bank conflict on every ds_read_b128).
// The v1 kernel this replaces (thread = pixel x channel vector, 4 x 8 tiles, no prefetch) ran the three stride-2 blocks
// of the BCD step at 0.42 TB/s.
// Empty asm that "uses" the 24 accumulators: keeps the scheduler from hoisting the LDS reads of all nine (ky, kx) steps
// above the first FMA (432 live registers, 836 B of scratch per lane without it).
__device__ __forceinline__ void pin_acc3(float (&a)[3][8]) {
  asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]),
                    "+v"(a[0][7]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]),
                    "+v"(a[1][6]), "+v"(a[1][7]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[2][4]),
                    "+v"(a[2][5]), "+v"(a[2][6]), "+v"(a[2][7]));
}
template <int TT> __device__ __forceinline__ void pin_acc_tt(float (&a)[TT][8]) {
  if constexpr (TT == 3) pin_acc3(a);
}
constexpr int S2_TH = 4, S2_TW = 16, S2_IH = 2 * S2_TH + 1, S2_IW = 2 * S2_TW + 1, S2_HY = S2_TH + 1, S2_HX = S2_TW + 1;
template <int TT> struct V2S2Geo {
  static constexpr int PAR = S2_HY * S2_HX;                     // float4 units per (frame, parity) plane
  static constexpr int PLANE = TT * 4 * PAR + 1;                // per (channel vector, half vector)
  static constexpr int NI = TT * S2_IH * S2_IW * DW_CV;         // staged 8-channel vectors per tile
  static constexpr int SL = (NI + 255) / 256;
};

template <typename T, int TT>
__global__ __launch_bounds__(256) void dw_fwd_v2s2_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                          const float* __restrict__ w, T* __restrict__ y,
                                                          double* __restrict__ nc, const DwGeom g,
                                                          const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef RawD<T> RW;
  typedef V2S2Geo<TT> G;
  constexpr int NI = G::NI, SL = G::SL, PLANE = G::PLANE, PAR = G::PAR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                  // [27][32]
  float* fss = wl + 27 * 32;                                   // [2][32] scale | shift of this chunk (fin.sums mode)
  float4* tile = reinterpret_cast<float4*>(fss + 64);          // [4 cv][2 halves][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wcv = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave's channel vector
  const int lx = lane & 15, ly = lane >> 4;                    // lane = output pixel (ly, lx) of the tile
  const int tiles_x = (g.Wo + S2_TW - 1) / S2_TW, tiles_y = (g.Ho + S2_TH - 1) / S2_TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx, tg = co.group % gx;
  const int c0 = chunk * DW_CV * 8;
  const int scv = tid & 3;              // staging role: item i = tid + 256*slot -> (cv = i & 3, pixel = i >> 2)
  const int sbase = c0 + scv * 8;
  const bool s_ok = sbase < g.Cp;
  const int cbase = c0 + wcv * 8;       // compute role
  const bool c_ok = cbase < g.Cp;

  for (int i = tid; i < 27 * 32; i += 256) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  // per-slot constants of the staging role: offset inside the input tile (global) and inside the parity planes (LDS)
  typename RW::type raw[SL];
  unsigned vmask = 0;
#define S2_ISSUE(TL)                                                                            \
  {                                                                                             \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const int i_ = tid + sl * 256;                                                            \
      const int p_ = i_ >> 2;                                                                   \
      const int ix_ = p_ % S2_IW, q_ = p_ / S2_IW;                                              \
      const int iy_ = q_ % S2_IH, t_ = q_ / S2_IH;                                              \
      const int gy_ = ty_ * (2 * S2_TH) - 1 + iy_, gx_ = tx_ * (2 * S2_TW) - 1 + ix_;           \
      if (i_ < NI && s_ok && t_ < g.T && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) {     \
        raw[sl] = RW::load(x + ((((size_t)b * g.T + t_) * g.H + gy_) * g.W + gx_) * g.Cp + sbase); \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) S2_ISSUE(tl0)
  float sc[8], sh[8];
  if (fin.sums) {   // BatchNorm_a scale / shift rebuilt from conv_a's completed sums (csrc/bn_fin.h), as in the stride-1 kernel
    if (tid == 0 && co.chunk == 0 && co.group == 0 && fin.training && fin.nbt) *fin.nbt += 1;
    c3dfin::bn_consume(fin, g.C, g.Cp, c0, 32, co.group == 0, fss, fss + 32, tid, 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? fss[scv * 8 + j] : 0.f; sh[j] = s_ok ? fss[32 + scv * 8 + j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? ss[sbase + j] : 0.f; sh[j] = s_ok ? ss[g.Cp + sbase + j] : 0.f; }
  }
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i = tid + sl * 256;
      if (i < NI) {
        float f[8];
        if ((vmask >> sl) & 1u) {
          RW::cvt(raw[sl], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        const int p = i >> 2;
        const int ix = p % S2_IW, q = p / S2_IW;
        const int iy = q % S2_IH, t = q / S2_IH;
        const int d = ((t * 2 + (iy & 1)) * 2 + (ix & 1)) * PAR + (iy >> 1) * S2_HX + (ix >> 1);
        tile[(scv * 2 + 0) * PLANE + d] = make_float4(f[0], f[1], f[2], f[3]);
        tile[(scv * 2 + 1) * PLANE + d] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
    if (tl + 1 < tl1) S2_ISSUE(tl + 1)
    __syncthreads();

    float acc[TT][8];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const float4* pl0 = tile + (wcv * 2 + 0) * PLANE + ly * S2_HX + lx;
    const float4* pl1 = tile + (wcv * 2 + 1) * PLANE + ly * S2_HX + lx;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float wk[3][8];  // wave-uniform: broadcast LDS reads
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) lds_ld8v2(wl + (kt * 9 + ky * 3 + kx) * 32 + wcv * 8, wk[kt]);
        const int off = ((ky & 1) * 2 + (kx & 1)) * PAR + (ky >> 1) * S2_HX + (kx >> 1);   // compile-time immediate
#pragma unroll
        for (int ti = 0; ti < TT; ++ti) {
          const float4 h0 = pl0[ti * 4 * PAR + off], h1 = pl1[ti * 4 * PAR + off];
          const float in[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[to][j] = fmaf(in[j], wk[kt][j], acc[to][j]);
            }
          }
        }
        pin_acc_tt<TT>(acc);
      }
    }
    const int ox = tx * S2_TW + lx, oy = ty * S2_TH + ly;
    if (c_ok && oy < g.Ho && ox < g.Wo) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < g.T) {
          T* dst = y + ((((size_t)b * g.T + t) * g.Ho + oy) * g.Wo + ox) * g.Cp + cbase;
          Vec8<T>::store(dst, acc[t]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float r = round_as<T>(acc[t][j]);
            s1[j] += r; s2[j] = fmaf(r, r, s2[j]);
          }
        }
      }
    }
  }
#undef S2_ISSUE
  if (nc == nullptr) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r1 = wave_sum(s1[j]), r2 = wave_sum(s2[j]);
    const int c = cbase + j;
    if (lane == 0 && c < g.C) {
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)r1);
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)r2);
    }
  }
}

template <typename T, int TT>
int launch_fwd_v2s2(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                    hipStream_t stream, const c3d_bn_fin* fin = nullptr) {
  const size_t lds = (27 * 32 + 64) * sizeof(float) + (size_t)DW_CV * 2 * V2S2Geo<TT>::PLANE * sizeof(float4);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;   // (five frames: 218 KB -- the v1 kernel)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2s2_kernel<T, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.Wo + S2_TW - 1) / S2_TW) * ((g.Ho + S2_TH - 1) / S2_TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  int tpw = c3d_knob("C3D_DWF2_TPW", 16);   // one workgroup per CU (LDS): ~2 rounds of workgroups, >= 4 tiles for the prefetch
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 2L * device_cus()) tpw >>= 1;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  c3d_bn_fin f0;
  std::memset(&f0, 0, sizeof(f0));
  dw_fwd_v2s2_kernel<T, TT><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                              reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int S> struct DwTile;  // forward / wgrad output tile per workgroup
template <typename T> struct DwTile<T, 1> { static constexpr int TH = 8, TW = 8; };
template <typename T> struct DwTile<T, 2> { static constexpr int TH = 4, TW = 8; };

template <typename T, int S, int TT>
int launch_fwd_t(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                 hipStream_t stream) {
  constexpr int TH = DwTile<T, S>::TH, TW = DwTile<T, S>::TW;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, NTHR = TH * TW * DW_CV;
  const size_t lds = (27 * 32 + (NTHR / 64) * DW_CV * 16) * sizeof(float) +
                     (size_t)g.T * IH * IW * 32 * sizeof(typename LdsStore<T>::type);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_kernel<T, S, TH, TW, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.Wo + TW - 1) / TW) * ((g.Ho + TH - 1) / TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  static const int env_tpw = c3d_env("C3D_DWF1_TPW") ? atoi(c3d_env("C3D_DWF1_TPW")) : 0;
  int tpw = 8;   // no prefetch in this kernel: the walk only amortises the weight staging and the statistics flush
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 2L * device_cus()) tpw >>= 1;   // swept: 8 is best
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  dw_fwd_kernel<T, S, TH, TW, TT><<<grid, dim3(NTHR), lds, stream>>>(
      reinterpret_cast<const T*>(x), ss, w, reinterpret_cast<T*>(y), nc, g, tpw);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int S>
int launch_fwd(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
               hipStream_t stream) {
  if (g.T <= 3) return launch_fwd_t<T, S, 3>(x, ss, w, y, nc, g, stream);
  return launch_fwd_t<T, S, 5>(x, ss, w, y, nc, g, stream);
}

thread_local c3d_bn_fin g_bwd_data_fin = {};   // set by c3d_dw333_bwd_data_fin for the launch it wraps

template <typename T, int S, int TT>
int launch_bwd_data_t(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC,
                      const float* w, const void* a, const float* ss_a, const float* mr_a, void* t2, double* dsums,
                      const DwGeom& g, hipStream_t stream) {
  constexpr int TH = 8, TW = 8;
  constexpr int DH = (S == 1) ? TH + 2 : TH / 2 + 2, DW_ = (S == 1) ? TW + 2 : TW / 2 + 2;
  constexpr int NTHR = TH * TW * DW_CV;
  const size_t lds = (27 * 32 + 7 * 32 + (NTHR / 64) * DW_CV * 16) * sizeof(float) +
                     (size_t)TT * DH * DW_ * 32 * sizeof(float);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_data_kernel<T, S, TH, TW, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.W + TW - 1) / TW) * ((g.H + TH - 1) / TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  // walk length: long enough to amortise the prologue and pipeline the loads, short enough for ~2 workgroups per CU
  static const int env_tpw = c3d_env("C3D_DWBD_TPW") ? atoi(c3d_env("C3D_DWBD_TPW")) : 0;
  // measured best with the GPU to itself: 16 / 16 / 8 tiles for the 128x128 / 64x64 / 32x32 stages (~2 workgroups per
  // CU); in the train step this kernel shares the CUs with the side stream's weight gradients and shorter walks win
  // (finer units for the dispatcher): 8 everywhere -> 31.66 -> 31.46 ms, 32.08 -> 31.92 ms per step on two boxes
  // (4: no better, 32: +1.3 ms; non-powers of two +0.4-0.9 ms)
  int tpw = 8;
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 7L * device_cus() / 4) tpw >>= 1;
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  dw_bwd_data_kernel<T, S, TH, TW, TT><<<grid, dim3(NTHR), lds, stream>>>(
      reinterpret_cast<const T*>(t1), reinterpret_cast<const T*>(bb), cA, cB, cC, w, reinterpret_cast<const T*>(a),
      ss_a, mr_a, reinterpret_cast<T*>(t2), dsums, g, tpw, g_bwd_data_fin);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int S>
int launch_bwd_data(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC,
                    const float* w, const void* a, const float* ss_a, const float* mr_a, void* t2, double* dsums,
                    const DwGeom& g, hipStream_t stream) {
  if (g.T <= 3) return launch_bwd_data_t<T, S, 3>(t1, bb, cA, cB, cC, w, a, ss_a, mr_a, t2, dsums, g, stream);
  return launch_bwd_data_t<T, S, 5>(t1, bb, cA, cB, cC, w, a, ss_a, mr_a, t2, dsums, g, stream);
}

template <typename T, int S>
int launch_wgrad(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const void* a,
                 const float* ss_a, float* dw, const DwGeom& g, hipStream_t stream) {
  constexpr int TH = DwTile<T, S>::TH, TW = DwTile<T, S>::TW;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, NTHR = TH * TW * DW_CV * 3 + DW_LOADERS;
  typedef typename LdsStore<T>::type L;
  const size_t buf = (size_t)g.T * (IH * IW + TH * TW) * 32 * sizeof(L);
  const size_t fixed = 27 * 32 * sizeof(float);
  const int nbuf = fixed + 2 * buf <= 160 * 1024 ? 2 : 1;
  const size_t lds = fixed + nbuf * buf;
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad_kernel<T, S, TH, TW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.Wo + TW - 1) / TW) * ((g.Ho + TH - 1) / TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  const long items = (long)g.B * ntiles;
  // one workgroup is resident per CU: size the walk so that the whole grid is ONE round of workgroups
  const long target = dw_wgrad_target_wgs();
  long gx = target / chunks;
  if (gx >= N_XCD) gx = gx / N_XCD * N_XCD;  // whole XCD rows: the chunk-sibling order deals groups 8 at a time
  if (gx < 1) gx = 1;
  const long ipw = (items + gx - 1) / gx;
  dim3 grid(chunk_order_grid(chunks, (items + ipw - 1) / ipw));
  dw_wgrad_kernel<T, S, TH, TW><<<grid, dim3(NTHR), lds, stream>>>(
      reinterpret_cast<const T*>(t1), reinterpret_cast<const T*>(bb), cA, cB, cC, reinterpret_cast<const T*>(a),
      ss_a, dw, g, (int)ipw, nbuf);
  C3D_CHECK_LAUNCH();
  return 0;
}

// bf16, T = 3 (three slots) or T = 5 (two slots): LDS-DMA producer.  Returns C3D_E_UNSUPPORTED when the
// geometry does not fit (the caller then uses the register-staged kernel).
template <int S, int TT>
int launch_wgrad_dma_t(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const void* a,
                       const float* ss_a, float* dw, const DwGeom& g, hipStream_t stream) {
  constexpr int TH = DwTile<bf16_t, S>::TH, TW = DwTile<bf16_t, S>::TW;
  typedef WgDma<S, TH, TW, TT> G;
  if (g.T != TT || G::LDS_BYTES > 160 * 1024) return C3D_E_UNSUPPORTED;
  const int ntiles = ((g.Wo + TW - 1) / TW) * ((g.Ho + TH - 1) / TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  const long items = (long)g.B * ntiles;
  const long target = dw_wgrad_target_wgs();
  long gx = target / chunks;
  if (gx >= N_XCD) gx = gx / N_XCD * N_XCD;  // whole XCD rows: the chunk-sibling order deals groups 8 at a time
  if (gx < 1) gx = 1;
  const long ipw = (items + gx - 1) / gx;
  if ((ipw + ntiles - 2) / ntiles + 1 > G::MAXB) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad_dma_kernel<S, TH, TW, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(chunk_order_grid(chunks, (items + ipw - 1) / ipw));
  dw_wgrad_dma_kernel<S, TH, TW, TT><<<grid, dim3(G::NCOMP + DW_LOADERS), G::LDS_BYTES, stream>>>(
      reinterpret_cast<const bf16_t*>(t1), reinterpret_cast<const bf16_t*>(bb), cA, cB, cC,
      reinterpret_cast<const bf16_t*>(a), ss_a, dw, g, (int)ipw);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int S>
int launch_wgrad_dma(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const void* a,
                     const float* ss_a, float* dw, const DwGeom& g, hipStream_t stream) {
  if (g.T == 3) return launch_wgrad_dma_t<S, 3>(t1, bb, cA, cB, cC, a, ss_a, dw, g, stream);
  if (g.T == 5) return launch_wgrad_dma_t<S, 5>(t1, bb, cA, cB, cC, a, ss_a, dw, g, stream);
  return C3D_E_UNSUPPORTED;
}

// bf16, stride 1, T = 3: pixel-paired v_dot2c consumer (see dw_wgrad_dot2_kernel).
int launch_wgrad_dot2(const void* t1, const void* bb, const float* cA, const float* cB, const float* cC, const void* a,
                      const float* ss_a, float* dw, const DwGeom& g, hipStream_t stream) {
  typedef WgDot2<3> G;
  if (g.T != 3 || g.stride != 1 || G::LDS_BYTES > 160 * 1024) return C3D_E_UNSUPPORTED;
  const int ntiles = ((g.Wo + D2_TW - 1) / D2_TW) * ((g.Ho + D2_TH - 1) / D2_TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  const long items = (long)g.B * ntiles;
  const long target = dw_wgrad_target_wgs();
  long gx = target / chunks;
  if (gx >= N_XCD) gx = gx / N_XCD * N_XCD;
  if (gx < 1) gx = 1;
  const long ipw = (items + gx - 1) / gx;
  if ((ipw + ntiles - 2) / ntiles + 1 > G::MAXB) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_wgrad_dot2_kernel<3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(chunk_order_grid(chunks, (items + ipw - 1) / ipw));
  dw_wgrad_dot2_kernel<3><<<grid, dim3(G::NCOMP + D2_LOADERS), G::LDS_BYTES, stream>>>(
      reinterpret_cast<const bf16_t*>(t1), reinterpret_cast<const bf16_t*>(bb), cA, cB, cC,
      reinterpret_cast<const bf16_t*>(a), ss_a, dw, g, (int)ipw);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int c3d_dw333_fwd(const void* x, const float* ss, const float* w, void* y, double* nc_sums, int32_t B,
                             int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                             void* stream) {
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!x || !ss || !w || !y || !geom_ok(g)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef C3D_TUNING
  if (stride == 1 && dtype == C3D_DT_BF16 && T <= 3 && c3d_dw_toeplitz_enabled()) {   // matrix-core kernel (dw_toeplitz.hip)
    const int rc = c3d_dw333_fwd_toeplitz(x, ss, w, y, nc_sums, B, T, H, W, C, Cp, s);
    if (rc != C3D_E_UNSUPPORTED) { if (rc == 0) C3D_CHECK_LAUNCH(); return rc; }
  }
#endif
  if (stride == 1) {  // v2 mapping (wave = channel vector, lane = x-strip)
    int rc = C3D_E_UNSUPPORTED;
    if (dtype == C3D_DT_F32) rc = T <= 3 ? launch_fwd_v2<float, 3>(x, ss, w, y, nc_sums, g, s)
                                         : launch_fwd_v2<float, 5>(x, ss, w, y, nc_sums, g, s);
    else if (dtype == C3D_DT_BF16) rc = T <= 3 ? launch_fwd_v2<bf16_t, 3>(x, ss, w, y, nc_sums, g, s)
                                               : launch_fwd_v2<bf16_t, 5>(x, ss, w, y, nc_sums, g, s);
    else return C3D_E_BADARG;
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (stride == 2 && T <= 3 && c3d_knob("C3D_DWF2_V2", 1)) {   // polyphase v2 (three frames: the tile fits LDS)
    int rc = C3D_E_UNSUPPORTED;
    if (dtype == C3D_DT_F32) rc = launch_fwd_v2s2<float, 3>(x, ss, w, y, nc_sums, g, s);
    else if (dtype == C3D_DT_BF16) rc = launch_fwd_v2s2<bf16_t, 3>(x, ss, w, y, nc_sums, g, s);
    else return C3D_E_BADARG;
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (dtype == C3D_DT_F32) return stride == 1 ? launch_fwd<float, 1>(x, ss, w, y, nc_sums, g, s)
                                              : launch_fwd<float, 2>(x, ss, w, y, nc_sums, g, s);
  if (dtype == C3D_DT_BF16) return stride == 1 ? launch_fwd<bf16_t, 1>(x, ss, w, y, nc_sums, g, s)
                                               : launch_fwd<bf16_t, 2>(x, ss, w, y, nc_sums, g, s);
  return C3D_E_BADARG;
}

extern "C" int c3d_dw333_fwd_fin(const void* x, const c3d_bn_fin* fin, const float* w, void* y, double* nc_sums,
                                 int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride,
                                 int32_t dtype, void* stream) {
  if (!fin || !fin->sums || !fin->ss || !fin->gamma || !fin->beta || !fin->training) return C3D_E_BADARG;
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!x || !w || !y || !geom_ok(g)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef C3D_TUNING
  const bool tz = stride == 1 && dtype == C3D_DT_BF16 && T <= 3 && c3d_dw_toeplitz_enabled();
#else
  const bool tz = false;
#endif
  if (!tz && stride == 1 && (dtype == C3D_DT_F32 || dtype == C3D_DT_BF16)) {
    int rc;
    if (dtype == C3D_DT_F32) rc = T <= 3 ? launch_fwd_v2<float, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                                         : launch_fwd_v2<float, 5>(x, fin->ss, w, y, nc_sums, g, s, fin);
    else rc = T <= 3 ? launch_fwd_v2<bf16_t, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                     : launch_fwd_v2<bf16_t, 5>(x, fin->ss, w, y, nc_sums, g, s, fin);
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (stride == 2 && T <= 3 && (dtype == C3D_DT_F32 || dtype == C3D_DT_BF16) && c3d_knob("C3D_DWF2_V2", 1)) {
    const int rc = dtype == C3D_DT_F32 ? launch_fwd_v2s2<float, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                                       : launch_fwd_v2s2<bf16_t, 3>(x, fin->ss, w, y, nc_sums, g, s, fin);
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  // no folded kernel for this shape: the separate launch, then the plain kernel
  const int rc = c3d_bn_finalize(fin->sums, C3D_STAT_STRIPES, fin->count, fin->gamma, fin->beta, fin->running_mean,
                                 fin->running_var, fin->nbt, fin->momentum, fin->eps, C, Cp, fin->training, fin->ss,
                                 fin->mr, stream);
  if (rc) return rc;
  return c3d_dw333_fwd(x, fin->ss, w, y, nc_sums, B, T, H, W, C, Cp, stride, dtype, stream);
}

extern "C" int c3d_dw333_bwd_data(const void* t1, const void* b, const float* coefA, const float* coefB,
                                  const float* coefC, const float* w, const void* a, const float* ss_a,
                                  const float* mr_a, void* t2, double* dsums, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp,
                                  int32_t stride, int32_t dtype, void* stream) {
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!t1 || !b || !coefA || !coefB || !coefC || !w || !a || !ss_a || !mr_a || !t2 || !dsums || !geom_ok(g))
    return C3D_E_BADARG;
  if (stride == 2 && ((H | W) & 1)) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32)
    return stride == 1 ? launch_bwd_data<float, 1>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, g, s)
                       : launch_bwd_data<float, 2>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, g, s);
  if (dtype == C3D_DT_BF16)
    return stride == 1 ? launch_bwd_data<bf16_t, 1>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, g, s)
                       : launch_bwd_data<bf16_t, 2>(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, g, s);
  return C3D_E_BADARG;
}

extern "C" int c3d_dw333_bwd_data_fin(const void* t1, const void* b, const float* coefA, const float* coefB,
                                      const float* coefC, const float* w, const void* a, const float* ss_a,
                                      const float* mr_a, void* t2, double* dsums, int32_t B, int32_t T, int32_t H, int32_t W,
                                      int32_t C, int32_t Cp, int32_t stride, int32_t dtype, const c3d_bn_fin* fin, void* stream) {
  if (fin && fin->ticket && (!fin->gamma || !fin->ss || !fin->mr || !(fin->count > 0))) return C3D_E_BADARG;
  g_bwd_data_fin = fin ? *fin : c3d_bn_fin{};
  const int rc = c3d_dw333_bwd_data(t1, b, coefA, coefB, coefC, w, a, ss_a, mr_a, t2, dsums, B, T, H, W, C, Cp, stride, dtype, stream);
  g_bwd_data_fin = c3d_bn_fin{};
  return rc;
}

extern "C" int c3d_dw333_wgrad(const void* t1, const void* b, const float* coefA, const float* coefB,
                               const float* coefC, const void* a, const float* ss_a, float* dw, int32_t B,
                               int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride,
                               int32_t dtype, void* stream) {
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!t1 || !b || !coefA || !coefB || !coefC || !a || !ss_a || !dw || !geom_ok(g)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32)
    return stride == 1 ? launch_wgrad<float, 1>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s)
                       : launch_wgrad<float, 2>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s);
  if (dtype == C3D_DT_BF16) {
    static const bool no_dma = c3d_env("C3D_DWWG_NODMA") != nullptr;
    static const bool no_dot2 = c3d_env("C3D_DWWG_NODOT2") != nullptr;
    if (!no_dma && !no_dot2 && stride == 1) {
      const int rc = launch_wgrad_dot2(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s);
      if (rc != C3D_E_UNSUPPORTED) return rc;
    }
    if (!no_dma) {
      const int rc = stride == 1 ? launch_wgrad_dma<1>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s)
                                 : launch_wgrad_dma<2>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s);
      if (rc != C3D_E_UNSUPPORTED) return rc;
    }
    return stride == 1 ? launch_wgrad<bf16_t, 1>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s)
                       : launch_wgrad<bf16_t, 2>(t1, b, coefA, coefB, coefC, a, ss_a, dw, g, s);
  }
  return C3D_E_BADARG;
}

#ifdef C3D_PW_CLOCK
extern "C" int c3d_debug_dw_clock(unsigned long long* out, int reset) {   // out[DCLK_WAVES][10]
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_dw_clk), sizeof(unsigned long long) * DCLK_WAVES * 10);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(c3d_dw_clk));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * DCLK_WAVES * 10);
  }
  return (int)e;
}
#endif
