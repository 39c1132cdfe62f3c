// X3D stem on the matrix cores (reference model/x3d.py:70-106; scalar-FMA version and the interface: stem.hip).
//
// The spatial 1x3x3 convolution 3 -> 24 is a GEMM with K = 27: per 16 pixels of a row and frame
//     V[24 ch (2 x 16)][16 px] = W[24][27] * patch[27][16 px]
// run as 7 k-steps of v_mfma_f32_16x16x4_f32 (FULL f32 operands: the parity path and the bf16 path share the kernel,
// only the storage type of u / g0 / dv differs).  The weight fragments live in registers for the whole launch, a
// patch fragment is ONE scalar LDS read per lane (pixel n = lane % 16, k = 4*step + lane / 16) from the f32 x tile.
// The result layout D[m = channel 4*(lane/16)+i][n = pixel lane%16] gives each lane 4 consecutive channels of one
// pixel for every frame: the temporal 5x1x1 depthwise convolution, the BatchNorm statistics, the backward affine and
// the d w_xy products are per-lane register arithmetic, and stores are 8/16-byte pieces of contiguous pixel rows.
// The scalar kernels spent 648 FMA per pixel-frame on the VALU: 0.39 / 0.58 / 0.60 ms per step at B=32 (fwd / dv / wx);
// these take 0.18 / 0.28 / 0.43 ms.  Phase timing of stem_bwd_wx (launches with a phase compiled out): staging 0.13 ms
// (3.6 TB/s), input gradient +0.12 ms (LDS-read bound: 12 b128 reads per 72 FMA), d w_t +0.18 ms (f32 MFMA at ~45 % of
// its 82 us floor).
//
//   stem_fwd     x -> u (+ per-channel sum, sum of squares)
//   stem_bwd_dv  (g0, u) -> du on load, v recomputed on the MFMA, dv = conv_xy^T(du), d w_xy
//   stem_bwd_wx  dv -> d w_t (GEMM over pixels: D[ch][k] += dv[px][ch] * patch[px][k]) and the input gradient of the
//                perception frames (3 outputs per pixel: VALU)
#include "common.h"
#include "stem_mfma.h"
#include "launch_hints.h"
#include <cstdlib>
#include "../../include/change3d_hip.h"

namespace {

constexpr int SC = 24, SCI = 3;
constexpr int TH = 8, TW = 32;           // output tile of a workgroup: 4 waves x 2 rows x 2 pixel groups of 16
constexpr int IH = TH + 2, IW = TW + 2;  // x tile with the 3x3 halo
constexpr int IHW = IH * IW;
constexpr int NTHR = 256;
constexpr int KS = 7;                    // k-steps of 4 (27 -> 28)

struct Geom { int B, T, H, W; };

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// x tile [ci][t][IH][IW] (f32, zero outside the image).  The loads of a tile are ALL issued before the first LDS
// store (a load -> wait -> store loop is one exposed global round trip per iteration: 12 per tile), and the kernels
// issue the next tile's loads before they compute the current one.
template <int TT> struct XTile {
  static constexpr int ITEMS = SCI * TT * IHW;
  static constexpr int NL = (ITEMS + NTHR - 1) / NTHR;
  float r[NL];
  __device__ __forceinline__ void issue(const float* __restrict__ x, const Geom& g, int b, int y0, int x0, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = tid + j * NTHR;
      const int ix = i % IW;
      int q = i / IW;
      const int iy = q % IH;
      q /= IH;  // ci*T + t
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      r[j] = 0.f;
      if (i < ITEMS && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        r[j] = x[(((size_t)b * SCI * TT + q) * g.H + gy) * g.W + gx];
    }
  }
  __device__ __forceinline__ void commit(float* xt, int tid) const {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = tid + j * NTHR;
      if (i < ITEMS) xt[i] = r[j];
    }
  }
};

// per-lane constants of the spatial GEMM: weight fragments (A operand) and LDS offsets of the patch values (B operand)
struct SpatialFrag {
  float wa[2][KS];
  int koff[KS];
  __device__ __forceinline__ void init(const float* __restrict__ w_t, int lane, int T) {
    const int kk = lane >> 4, m = lane & 15;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + kk;
      const bool kv = k < 27;
      const int ci = k / 9, r = k - ci * 9, ky = r / 3, kx = r - ky * 3;
      koff[s] = kv ? ci * T * IHW + ky * IW + kx : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int ch = mt * 16 + m;
        wa[mt][s] = (kv && ch < SC) ? w_t[ch * 27 + k] : 0.f;
      }
    }
  }
  // v[t][mt] (4 channels x this lane's pixel) for the TT frames of tile row `row`, pixel group `pt`
  template <int TT>
  __device__ __forceinline__ void conv(f32x4_t (&v)[TT][2], const float* xt, int row, int pt, int n) const {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      v[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      v[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const float* base = xt + t * IHW + row * IW + pt * 16 + n;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float bv = base[koff[s]];
        v[t][0] = mfma4(wa[0][s], bv, v[t][0]);
        v[t][1] = mfma4(wa[1][s], bv, v[t][1]);
      }
    }
  }
};

template <typename T> struct Q4;   // 4 consecutive channels of a pixel
template <> struct Q4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ raw load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void cvt(const raw& r, float (&f)[4]) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};
template <> struct Q4<bf16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void cvt(const raw& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[4]) {
    uint2 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    *reinterpret_cast<uint2*>(p) = v;
  }
};

template <typename T> struct RawV;   // unconverted 8-channel vector
template <> struct RawV<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type zero() { type t; t.a = make_float4(0.f, 0.f, 0.f, 0.f); t.b = t.a; return t; }
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ void store(float* p, const type& v) {
    *reinterpret_cast<float4*>(p) = v.a; *reinterpret_cast<float4*>(p + 4) = v.b;
  }
};
template <> struct RawV<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void store(bf16_t* p, const type& v) { *reinterpret_cast<uint4*>(p) = v; }
};

// sum over the 16 lanes that share lane/16 (same channels, different pixels): fixed butterfly order
template <typename V> __device__ __forceinline__ V row16_sum(V v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------- forward
template <typename T, int TT>
__global__ __launch_bounds__(NTHR) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w_t,
                                                             const float* __restrict__ w_xy, T* __restrict__ u,
                                                             double* __restrict__ sums, const Geom g,
                                                             const int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  double* red = reinterpret_cast<double*>(sm);       // [4 waves][2 (sum, sumsq)][32 channels]
  float* xt = sm + 2 * 4 * 2 * 32;                   // [3][T][IH][IW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int b = blockIdx.y;
  SpatialFrag sf;
  sf.init(w_t, lane, TT);
  float wx[5][2][4];               // temporal taps of this lane's channels
#pragma unroll
  for (int dt = 0; dt < 5; ++dt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = mt * 16 + 4 * kk + i;
        wx[dt][mt][i] = ch < SC ? w_xy[ch * 5 + dt] : 0.f;
      }
  // BatchNorm statistics per lane in f64 (full-rate on this chip): an f32 running sum over ~100 values moved the stem's
  // scale / shift in the 8th digit, enough to push a ReLU pre-activation somewhere downstream across zero
  double s1[2][4], s2[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[mt][i] = 0.0; s2[mt][i] = 0.0; }

  const int tl0 = blockIdx.x * tiles_per_wg;
  const int tl1 = tl0 + tiles_per_wg < ntiles ? tl0 + tiles_per_wg : ntiles;
  XTile<TT> xq;
  if (tl0 < tl1) xq.issue(x, g, b, (tl0 / tiles_x) * TH, (tl0 % tiles_x) * TW, tid);
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
    xq.commit(xt, tid);
    __syncthreads();
    if (tl + 1 < tl1) xq.issue(x, g, b, ((tl + 1) / tiles_x) * TH, ((tl + 1) % tiles_x) * TW, tid);
#pragma unroll 1
    for (int rp = 0; rp < 4; ++rp) {
      const int row = 2 * wave + (rp >> 1), pt = rp & 1;
      const int gy = ty * TH + row, gx = tx * TW + pt * 16 + n;
      f32x4_t v[TT][2];
      sf.conv<TT>(v, xt, row, pt, n);
      if (gy >= g.H || gx >= g.W) continue;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (mt * 16 + 4 * kk >= SC) continue;
          float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dt = 0; dt < 5; ++dt) {
            const int ti = t + dt - 2;
            if (ti >= 0 && ti < TT) {
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = fmaf(wx[dt][mt][i], v[ti][mt][i], o[i]);
            }
          }
          Q4<T>::store(u + ((((size_t)b * TT + t) * g.H + gy) * g.W + gx) * SC + mt * 16 + 4 * kk, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) { const double r = (double)round_as<T>(o[i]); s1[mt][i] += r; s2[mt][i] = fma(r, r, s2[mt][i]); }
        }
      }
    }
  }
  if (!sums) return;
  // lanes -> 16-lane rows (fixed butterfly) -> waves (LDS, fixed order) -> one f64 atomic per value and workgroup
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double r1 = row16_sum(s1[mt][i]), r2 = row16_sum(s2[mt][i]);
      if (n == 0) {
        red[(wave * 2 + 0) * 32 + mt * 16 + 4 * kk + i] = r1;
        red[(wave * 2 + 1) * 32 + mt * 16 + 4 * kk + i] = r2;
      }
    }
  __syncthreads();
  if (tid < 2 * SC) {
    const int which = tid / SC, c = tid - which * SC;
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += red[(w * 2 + which) * 32 + c];
    atomicAdd(sums + (size_t)which * SC + c, a);
  }
}

// ------------------------------------------------------------------------------------------------- backward: dv, d w_xy
template <typename T, int TT>
#ifndef STEM_DV_OCC2
#define STEM_DV_OCC2 0
#endif
__global__ __launch_bounds__(NTHR, (TT == 3 && STEM_DV_OCC2) ? 2 : 1) void stem_bwd_dv_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w_t, const float* __restrict__ w_xy,
    const T* __restrict__ g0, const T* __restrict__ u, const float* __restrict__ coef, T* __restrict__ dv,
    float* __restrict__ dw_xy, const Geom g, const int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                 // [4 waves][5][32]
  float* xt = red + 4 * 5 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int b = blockIdx.y;
  SpatialFrag sf;
  sf.init(w_t, lane, TT);
  float wx[5][2][4], cA[2][4], cB[2][4], cC[2][4], dwx[5][2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = mt * 16 + 4 * kk + i;
      const bool ok = ch < SC;
      cA[mt][i] = ok ? coef[ch] : 0.f; cB[mt][i] = ok ? coef[SC + ch] : 0.f; cC[mt][i] = ok ? coef[2 * SC + ch] : 0.f;
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) { wx[dt][mt][i] = ok ? w_xy[ch * 5 + dt] : 0.f; dwx[dt][mt][i] = 0.f; }
    }
  const int tl0 = blockIdx.x * tiles_per_wg;
  const int tl1 = tl0 + tiles_per_wg < ntiles ? tl0 + tiles_per_wg : ntiles;
  constexpr bool PREF = TT <= 3;    // the T = 4, 5 instances have no registers left for the next tile
  XTile<TT> xq;
  if (PREF && tl0 < tl1) xq.issue(x, g, b, (tl0 / tiles_x) * TH, (tl0 % tiles_x) * TW, tid);
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
    if (!PREF) xq.issue(x, g, b, ty * TH, tx * TW, tid);
    xq.commit(xt, tid);
    __syncthreads();
    if (PREF && tl + 1 < tl1) xq.issue(x, g, b, ((tl + 1) / tiles_x) * TH, ((tl + 1) % tiles_x) * TW, tid);
#pragma unroll 1
    for (int rp = 0; rp < 4; ++rp) {
      const int row = 2 * wave + (rp >> 1), pt = rp & 1;
      const int gy = ty * TH + row, gx = tx * TW + pt * 16 + n;
      const bool in_img = gy < g.H && gx < g.W;
      // this lane's (g0, u) pieces are requested before the MFMA phase
      typename Q4<T>::raw gr[TT][2], ur[TT][2];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (in_img && mt * 16 + 4 * kk < SC) {
            const size_t off = ((((size_t)b * TT + t) * g.H + gy) * g.W + gx) * SC + mt * 16 + 4 * kk;
            gr[t][mt] = Q4<T>::load(g0 + off);
            ur[t][mt] = Q4<T>::load(u + off);
          }
        }
      f32x4_t v[TT][2];
      sf.conv<TT>(v, xt, row, pt, n);
      if (!in_img) continue;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (mt * 16 + 4 * kk >= SC) continue;
        float du[TT][4];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          float gg[4], uu[4];
          Q4<T>::cvt(gr[t][mt], gg);
          Q4<T>::cvt(ur[t][mt], uu);
#pragma unroll
          for (int i = 0; i < 4; ++i) du[t][i] = fmaf(cA[mt][i], gg[i], fmaf(cC[mt][i], uu[i], cB[mt][i]));
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dt = 0; dt < 5; ++dt) {
            // u[to] += wxy[dt]*v[to+dt-2]  =>  dv[t] += wxy[dt]*du[t-dt+2];  d wxy[dt] += du[t]*v[t+dt-2]
            const int to = t - dt + 2;
            if (to >= 0 && to < TT) {
#pragma unroll
              for (int i = 0; i < 4; ++i) o[i] = fmaf(wx[dt][mt][i], du[to][i], o[i]);
            }
            const int ti = t + dt - 2;
            if (ti >= 0 && ti < TT) {
#pragma unroll
              for (int i = 0; i < 4; ++i) dwx[dt][mt][i] = fmaf(du[t][i], v[ti][mt][i], dwx[dt][mt][i]);
            }
          }
          Q4<T>::store(dv + ((((size_t)b * TT + t) * g.H + gy) * g.W + gx) * SC + mt * 16 + 4 * kk, o);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 5; ++dt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = row16_sum(dwx[dt][mt][i]);
        if (n == 0) red[(wave * 5 + dt) * 32 + mt * 16 + 4 * kk + i] = r;
      }
  __syncthreads();
  for (int i = tid; i < 5 * SC; i += NTHR) {
    const int dt = i / SC, c = i - dt * SC;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += red[(w * 5 + dt) * 32 + c];
    atomicAdd(dw_xy + c * 5 + dt, a);
  }
}

// ------------------------------------------------------------------------------------------------- backward: d w_t, d input
// One workgroup owns ONE spatial tile and walks the samples (the batch-summed input gradient and d w_t stay in
// registers, as in the scalar kernel).  dv is staged with its halo as raw T [t][IH][IW][24].
template <typename T, int TT>
__global__ __launch_bounds__(NTHR) void stem_bwd_wx_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w_t, const T* __restrict__ dv, float* __restrict__ dw_t,
    float* __restrict__ dP, const Geom g, const int t_first, const int n_frames, const int per_sample) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wt = sm;                                   // [9 taps][3 ci][24]  (input gradient)
  float* xt = wt + 27 * SC;                         // [3][T][IH][IW]
  T* dt_ = reinterpret_cast<T*>(xt + SCI * TT * IHW);   // [T][IH][IW][24]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, n = lane & 15;
  const int tiles_x = (g.W + TW - 1) / TW;
  for (int i = tid; i < 27 * SC; i += NTHR) {       // wt[ky*3+kx][ci][c] <- w_t[c][ci][ky][kx]
    const int q = i / SC, c = i - q * SC;
    const int sp = q / SCI, ci = q - sp * SCI;
    wt[i] = w_t[c * 27 + ci * 9 + sp];
  }
  // weight-gradient GEMM: D[mt][nt] (m = channel, n = patch index k) += sum_px dv[px][ch] * patch[px][k]
  //   A lane (m = lane%16 -> channel, k-step pixel = 4*s + lane/16);  B lane (pixel = 4*s + lane/16, n = lane%16 -> k)
  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int k = nt * 16 + n;
    const bool kv = k < 27;
    const int ci = k / 9, r = k - ci * 9, ky = r / 3, kx = r - ky * 3;
    boff[nt] = kv ? ci * TT * IHW + ky * IW + kx : 0;
  }
  f32x4_t dwacc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) dwacc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int px = tid % TW, py = tid / TW;           // input-gradient role: one pixel per thread (256 = TH*TW)
  float dxacc[TT][SCI];
#pragma unroll
  for (int k = 0; k < TT; ++k)
#pragma unroll
    for (int ci = 0; ci < SCI; ++ci) dxacc[k][ci] = 0.f;
  const int tl = blockIdx.x;
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  for (int b = blockIdx.y; b < g.B; b += gridDim.y) {
    // all loads of this sample's tiles are issued before the workgroup waits for the previous sample's readers
    XTile<TT> xq;
    xq.issue(x, g, b, y0, x0, tid);
    constexpr int DITEMS = TT * IHW * 3, DNL = (DITEMS + NTHR - 1) / NTHR;
    typename RawV<T>::type dr[DNL];
#pragma unroll
    for (int j = 0; j < DNL; ++j) {
      const int i = tid + j * NTHR;
      const int cvv = i % 3;
      int q = i / 3;
      const int ix = q % IW;
      q /= IW;
      const int iy = q % IH, t = q / IH;
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      dr[j] = RawV<T>::zero();
      if (i < DITEMS && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        dr[j] = RawV<T>::load(dv + ((((size_t)b * TT + t) * g.H + gy) * g.W + gx) * SC + cvv * 8);
    }
    __syncthreads();
    xq.commit(xt, tid);
#pragma unroll
    for (int j = 0; j < DNL; ++j) {
      const int i = tid + j * NTHR;     // dt_ index = ((t*IH + iy)*IW + ix)*3 + cvv = i
      if (i < DITEMS) RawV<T>::store(dt_ + (size_t)i * 8, dr[j]);
    }
    __syncthreads();
    // ---- d input for the perception frames (3 outputs per pixel)
    const int gy = y0 + py, gx = x0 + px;
    if (dP && gy < g.H && gx < g.W) {
#pragma unroll 1
      for (int k = 0; k < n_frames; ++k) {
        const int t = t_first + k;
        float dx[SCI] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
          for (int kx = 0; kx < 3; ++kx) {
            const T* dp = dt_ + ((size_t)(t * IH + py + 2 - ky) * IW + px + 2 - kx) * SC;
            const float* wp = wt + (ky * 3 + kx) * SCI * SC;     // [ci][24]: wave-uniform, read as float4 (broadcast)
#pragma unroll
            for (int cv = 0; cv < 3; ++cv) {
              float d[8];
              Vec8<T>::load(dp + cv * 8, d);
#pragma unroll
              for (int ci = 0; ci < SCI; ++ci) {
                float w8[8];
                Vec8<float>::load(wp + ci * SC + cv * 8, w8);
#pragma unroll
                for (int j = 0; j < 8; ++j) dx[ci] = fmaf(d[j], w8[j], dx[ci]);
              }
            }
          }
        }
        if (per_sample) {   // dP is a full NCDHW gradient [B][3][T][H][W]
#pragma unroll
          for (int ci = 0; ci < SCI; ++ci) dP[((((size_t)b * SCI + ci) * TT + t) * g.H + gy) * g.W + gx] = dx[ci];
        } else {            // dP is [3][n_frames][H][W], summed over the batch: flushed after the walk
#pragma unroll
          for (int q = 0; q < TT; ++q)
#pragma unroll
            for (int ci = 0; ci < SCI; ++ci) dxacc[q][ci] += q == k ? dx[ci] : 0.f;
        }
      }
    }
    // ---- d w_t: wave w takes tile rows {2w, 2w+1} of every frame
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
#pragma unroll 1
      for (int r = 0; r < 2; ++r) {
        const int qy = 2 * wave + r;
        const T* arow = dt_ + ((size_t)(t * IH + qy + 1) * IW + 1) * SC;     // inner pixels of this row
        const float* brow = xt + t * IHW + qy * IW;
#pragma unroll
        for (int s = 0; s < TW / 4; ++s) {
          const int qx = 4 * s + kk;
          const float a0 = ld1<T>(arow + (size_t)qx * SC + n);
          const float a1 = n < SC - 16 ? ld1<T>(arow + (size_t)qx * SC + 16 + n) : 0.f;
          const float b0 = brow[boff[0] + qx], b1 = brow[boff[1] + qx];
          dwacc[0][0] = mfma4(a0, b0, dwacc[0][0]); dwacc[0][1] = mfma4(a0, b1, dwacc[0][1]);
          dwacc[1][0] = mfma4(a1, b0, dwacc[1][0]); dwacc[1][1] = mfma4(a1, b1, dwacc[1][1]);
        }
      }
    }
  }
  if (dP && !per_sample) {
    const int gy = y0 + py, gx = x0 + px;
    if (gy < g.H && gx < g.W) {
#pragma unroll
      for (int k = 0; k < TT; ++k) {
        if (k < n_frames) {
#pragma unroll
          for (int ci = 0; ci < SCI; ++ci) {
            float* dst = dP + (((size_t)ci * n_frames + k) * g.H + gy) * g.W + gx;
            if (gridDim.y == 1) *dst += dxacc[k][ci];     // sole owner of this pixel
            else atomicAdd(dst, dxacc[k][ci]);
          }
        }
      }
    }
  }
  // d w_t: the four waves' partial tiles through LDS (fixed order), then one atomic per weight and workgroup
  __syncthreads();
  float* wred = reinterpret_cast<float*>(dt_);      // [4 waves][32 ch][32 k]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) wred[(wave * 32 + mt * 16 + 4 * kk + i) * 32 + nt * 16 + n] = dwacc[mt][nt][i];
  __syncthreads();
  for (int i = tid; i < SC * 27; i += NTHR) {
    const int c = i / 27, k = i - c * 27;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += wred[(w * 32 + c) * 32 + k];
    atomicAdd(dw_t + (size_t)c * 27 + k, a);
  }
}


// ---- stem_bwd_wx on the bf16 matrix cores (bf16 storage only; round 4).  Same tile, same walk over the samples as the f32
// kernel above; what changes is the arithmetic:
//   d w_t   D[ch][(tap, ci)] += sum_px dv[px][ch] * x[px + tap][ci]: PIXELS are the contraction index, both operands come out of
//           [pixel][channel] LDS tiles through ds_read_b64_tr_b16 -- dv as it is staged (24 channels per pixel), x as a bf16
//           [pixel][3 ci + 1 zero] tile whose 8-byte pixel IS one transposed-read chunk: a lane addresses the chunk of tap
//           4 nt + lane % 4, so the 16 columns of an n-tile are 4 taps x 4 ci without an im2col image.  6 x v_mfma_f32_16x16x16_bf16
//           and 5 LDS reads per 16 pixels (the f32 kernel: 16 x v_mfma_f32_16x16x4_f32, 32 scalar LDS reads).
//   d input of the perception frames: dx[ci][px] = sum_tap W[ci][tap][0..31] . dv[px - tap][0..31] -- one k-step of 32 per tap
//           (24 channels + a zero vector), the B fragment is ONE ds_read_b128 of a neighbour pixel, the nine weight fragments
//           live in registers (the layout of head_fwd_mfma_kernel); 9 MFMA per 16 pixels against 648 FMA per pixel on the VALU.
// x (the normalised input image) and w_t are rounded to bf16 for the products; sums stay f32.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;

__device__ __forceinline__ f32x4_t mfma32b(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int TT>
__global__ __launch_bounds__(NTHR, TT == 3 ? 2 : 1) void stem_bwd_wx_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ w_t, const bf16_t* __restrict__ dv, float* __restrict__ dw_t,
    float* __restrict__ dP, const Geom g, const int t_first, const int n_frames, const int per_sample) {
  constexpr int NF = TT - 2 > 0 ? TT - 2 : 1;        // perception frames a launch may carry (the host checks n_frames <= NF)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  bf16_t* dt_ = reinterpret_cast<bf16_t*>(sm);       // [T][IH][IW][24] raw dv with halo
  bf16_t* xb = dt_ + (size_t)TT * IHW * SC + 8;      // [T][IH][IW][4]: ci 0..2 + a zero  (the 16 bytes in front: m-tile 1 of the
                                                     // last pixel reads "channels 24..31" there -- rows of D nobody uses)
  bf16_t* zs = xb + (size_t)TT * IHW * 4;            // 16 bytes of zeros: k-group 3 of the input-gradient B fragment
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, li = lane & 15;
  const int tiles_x = (g.W + TW - 1) / TW;
  for (int i = tid; i < TT * IHW + 2 + 2; i += NTHR)   // xb (its zero column is never written again), the 8 elements in front, zs
    *reinterpret_cast<uint2*>(xb - 8 + (size_t)i * 4) = make_uint2(0, 0);
  // input gradient: A[m = ci][k = channel] of tap sp (rows ci >= 3 and channels >= 24 are zero)
  uint4 wA[9];
#pragma unroll
  for (int sp = 0; sp < 9; ++sp) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * g4 + j;
      f[j] = (li < SCI && c < SC) ? w_t[c * 27 + li * 9 + sp] : 0.f;
    }
    wA[sp] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
  // weight gradient: D[mt][nt], m = channel 16 mt + .., n = 4 (tap - 4 nt) + ci.  Lane offsets of the transposed reads:
  //   A chunk (pixel 4 g4 + li / 4, channels c0 + 4 (li % 4)) of dv;  B chunk = the pixel of tap 4 nt + li % 4 in xb
  const int pxl = 4 * g4 + (li >> 2);
  int boff[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    int tap = 4 * nt + (li & 3);
    if (tap > 8) tap = 8;                             // columns nobody reads: any valid address
    boff[nt] = ((tap / 3) * IW + pxl + tap % 3) * 4;
  }
  const int aoff = (IW + 1 + pxl) * SC + 4 * (li & 3);
  f32x4_t dwacc[2][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) dwacc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t dxa[NF][4];
#pragma unroll
  for (int k = 0; k < NF; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) dxa[k][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int tl = blockIdx.x;
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  for (int b = blockIdx.y; b < g.B; b += gridDim.y) {
    // the staging addresses do not depend on the sample: left alone, the compiler keeps all ~50 of them (offsets, masks, LDS
    // addresses) in registers across the walk -- 396 registers, one workgroup per CU.  An opaque copy of the thread index per
    // iteration makes it recompute them (a few dozen integer operations per tile).
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    XTile<TT> xq;
    xq.issue(x, g, b, y0, x0, tidv);
    constexpr int DITEMS = TT * IHW * 3, DNL = (DITEMS + NTHR - 1) / NTHR;
    uint4 dr[DNL];
#pragma unroll
    for (int j = 0; j < DNL; ++j) {
      const int i = tidv + j * NTHR;
      const int cvv = i % 3;
      int q = i / 3;
      const int ix = q % IW;
      q /= IW;
      const int iy = q % IH, t = q / IH;
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      dr[j] = make_uint4(0, 0, 0, 0);
      if (i < DITEMS && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        dr[j] = *reinterpret_cast<const uint4*>(dv + ((((size_t)b * TT + t) * g.H + gy) * g.W + gx) * SC + cvv * 8);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XTile<TT>::NL; ++j) {
      const int i = tidv + j * NTHR;
      if (i < XTile<TT>::ITEMS) {
        const int ix = i % IW;
        int q = i / IW;
        const int iy = q % IH;
        q /= IH;                                      // ci * T + t
        const int ci = q / TT, t = q - ci * TT;
        xb[((size_t)(t * IH + iy) * IW + ix) * 4 + ci] = f32_to_bf16(xq.r[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < DNL; ++j) {
      const int i = tidv + j * NTHR;
      if (i < DITEMS) *reinterpret_cast<uint4*>(dt_ + (size_t)i * 8) = dr[j];
    }
    __syncthreads();
    // ---- d input for the perception frames: wave w owns tile rows {2w, 2w+1} x the two 16-pixel segments
    if (dP) {
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (k < n_frames) {
          const int t = t_first + k;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 2 * wave + (q >> 1), xs = (q & 1) * 16;
            f32x4_t acc = per_sample ? f32x4_t{0.f, 0.f, 0.f, 0.f} : dxa[k][q];
#pragma unroll
            for (int sp = 0; sp < 9; ++sp) {
              const int ky = sp / 3, kx = sp - ky * 3;
              const bf16_t* bp = g4 < 3 ? dt_ + ((size_t)(t * IH + r + 2 - ky) * IW + xs + li + 2 - kx) * SC + 8 * g4 : zs;
              acc = mfma32b(wA[sp], *reinterpret_cast<const uint4*>(bp), acc);
            }
            if (per_sample) {   // dP is a full NCDHW gradient [B][3][T][H][W]; lanes 0..15 hold ci = 0..2 of pixel xs + li
              const int gy = y0 + r, gx = x0 + xs + li;
              if (g4 == 0 && gy < g.H && gx < g.W) {
#pragma unroll
                for (int ci = 0; ci < SCI; ++ci) dP[((((size_t)b * SCI + ci) * TT + t) * g.H + gy) * g.W + gx] = acc[ci];
              }
            } else {
              dxa[k][q] = acc;
            }
          }
        }
      }
    }
    // ---- d w_t
#pragma unroll 1
    for (int t = 0; t < TT; ++t) {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const int r = 2 * wave + (q >> 1), xs = (q & 1) * 16;
        const bf16_t* ap = dt_ + (size_t)((t * IH + r) * IW + xs) * SC + aoff;
        const bf16_t* bp = xb + (size_t)((t * IH + r) * IW + xs) * 4;
        const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)ap);
        const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(ap + 16));
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          const s16x4_t bf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(bp + boff[nt]));
          dwacc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, bf, dwacc[0][nt], 0, 0, 0);
          dwacc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, bf, dwacc[1][nt], 0, 0, 0);
        }
      }
    }
  }
  if (dP && !per_sample) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (k < n_frames) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gy = y0 + 2 * wave + (q >> 1), gx = x0 + (q & 1) * 16 + li;
          if (g4 == 0 && gy < g.H && gx < g.W) {
#pragma unroll
            for (int ci = 0; ci < SCI; ++ci) {
              float* dst = dP + (((size_t)ci * n_frames + k) * g.H + gy) * g.W + gx;
              if (gridDim.y == 1) *dst += dxa[k][q][ci];     // sole owner of this pixel
              else atomicAdd(dst, dxa[k][q][ci]);
            }
          }
        }
      }
    }
  }
  // d w_t: the four waves' partial tiles through LDS (fixed order), then one atomic per weight and workgroup
  __syncthreads();
  float* wred = reinterpret_cast<float*>(dt_);      // [4 waves][32 ch][32 k]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int tap = 4 * nt + (li >> 2), ci = li & 3;
      if (tap < 9 && ci < SCI) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wred[(wave * 32 + mt * 16 + 4 * g4 + i) * 32 + ci * 9 + tap] = dwacc[mt][nt][i];
      }
    }
  __syncthreads();
  for (int i = tid; i < SC * 27; i += NTHR) {
    const int c = i / 27, k = i - c * 27;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += wred[(w * 32 + c) * 32 + k];
    atomicAdd(dw_t + (size_t)c * 27 + k, a);
  }
}

template <typename T, int TT>
int fwd_t(const float* x, const float* w_t, const float* w_xy, void* u, double* sums, const Geom& g, hipStream_t s) {
  const size_t lds = (2 * 4 * 2 * 32 + (size_t)SCI * TT * IHW) * sizeof(float);
  const int ntiles = ((g.W + TW - 1) / TW) * ((g.H + TH - 1) / TH);
  static const int env_tpw = c3d_env("C3D_STEM_FWD_TPW") ? atoi(c3d_env("C3D_STEM_FWD_TPW")) : 0;   // tuning knob
  int tpw = 8;
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * g.B < 1024) tpw >>= 1;   // keep ~4 workgroups per CU
  if (env_tpw > 0) tpw = env_tpw;
  dim3 grid((ntiles + tpw - 1) / tpw, g.B);
  stem_fwd_mfma_kernel<T, TT><<<grid, NTHR, lds, s>>>(x, w_t, w_xy, reinterpret_cast<T*>(u), sums, g, tpw);
  return 0;
}

template <typename T, int TT>
int dv_t(const float* x, const float* w_t, const float* w_xy, const void* g0, const void* u, const float* coef, void* dv,
         float* dw_xy, const Geom& g, hipStream_t s) {
  const size_t lds = (4 * 5 * 32 + (size_t)SCI * TT * IHW) * sizeof(float);
  const int ntiles = ((g.W + TW - 1) / TW) * ((g.H + TH - 1) / TH);
  static const int env_tpw = c3d_env("C3D_STEM_DV_TPW") ? atoi(c3d_env("C3D_STEM_DV_TPW")) : 0;   // tuning knob
  int tpw = 16;   // every workgroup ends with 120 same-address atomics
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * g.B < 2L * 256) tpw >>= 1;
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid((ntiles + tpw - 1) / tpw, g.B);
  stem_bwd_dv_mfma_kernel<T, TT><<<grid, NTHR, lds, s>>>(x, w_t, w_xy, reinterpret_cast<const T*>(g0),
                                                         reinterpret_cast<const T*>(u), coef, reinterpret_cast<T*>(dv),
                                                         dw_xy, g, tpw);
  return 0;
}

template <typename T, int TT>
int wx_t(const float* x, const float* w_t, const void* dv, float* dw_t, float* dP, const Geom& g, int t_first,
         int n_frames, int per_sample, hipStream_t s) {
  size_t lds = (27 * SC + (size_t)SCI * TT * IHW) * sizeof(float) + (size_t)TT * IHW * SC * sizeof(T);
  const size_t red = (27 * SC + (size_t)SCI * TT * IHW) * sizeof(float) + 4 * 32 * 32 * sizeof(float);
  if (red > lds) lds = red;
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  const int ntiles = ((g.W + TW - 1) / TW) * ((g.H + TH - 1) / TH);
  int bsplit = (2 * 256 + ntiles - 1) / ntiles;   // split the batch only when there are too few tiles
  if (bsplit > g.B) bsplit = g.B;
  if (bsplit < 1) bsplit = 1;
  if (sizeof(T) == 2 && c3d_option_stem_mfma >= 2 && (!dP || n_frames <= (TT - 2 > 0 ? TT - 2 : 1))) {
    // bf16 storage: both gradients on the bf16 matrix cores (c3d_set_option(C3D_OPT_STEM_MFMA, 1) keeps the f32-MFMA kernel)
    size_t lb = ((size_t)TT * IHW * SC + 8 + (size_t)TT * IHW * 4 + 8) * sizeof(bf16_t);
    if (lb < 4 * 32 * 32 * sizeof(float)) lb = 4 * 32 * 32 * sizeof(float);
    static bool attr_b = false;
    if (!attr_b) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_wx_bf16_kernel<TT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_b = true;
    }
    stem_bwd_wx_bf16_kernel<TT><<<dim3(ntiles, bsplit), NTHR, lb, s>>>(x, w_t, reinterpret_cast<const bf16_t*>(dv), dw_t, dP, g,
                                                                      t_first, n_frames, per_sample);
    C3D_CHECK_LAUNCH();
    return 0;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_wx_mfma_kernel<T, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  stem_bwd_wx_mfma_kernel<T, TT><<<dim3(ntiles, bsplit), NTHR, lds, s>>>(x, w_t, reinterpret_cast<const T*>(dv), dw_t, dP,
                                                                        g, t_first, n_frames, per_sample);
  return 0;
}

#define STEM_DISPATCH(FN, ...)                                                         \
  if (dtype == C3D_DT_F32) {                                                           \
    if (T == 3) return FN<float, 3>(__VA_ARGS__);                                      \
    if (T == 4) return FN<float, 4>(__VA_ARGS__);                                      \
    if (T == 5) return FN<float, 5>(__VA_ARGS__);                                      \
  } else if (dtype == C3D_DT_BF16) {                                                   \
    if (T == 3) return FN<bf16_t, 3>(__VA_ARGS__);                                     \
    if (T == 4) return FN<bf16_t, 4>(__VA_ARGS__);                                     \
    if (T == 5) return FN<bf16_t, 5>(__VA_ARGS__);                                     \
  }                                                                                    \
  return C3D_E_UNSUPPORTED;

}  // namespace

bool c3d_stem_mfma_enabled() { return c3d_option_stem_mfma != 0; }

int c3d_stem_fwd_mfma(const float* x, const float* w_t, const float* w_xy, void* u, double* sums, int B, int T, int H,
                      int W, int dtype, hipStream_t s) {
  const Geom g{B, T, H, W};
  STEM_DISPATCH(fwd_t, x, w_t, w_xy, u, sums, g, s)
}

int c3d_stem_bwd_dv_mfma(const float* x, const float* w_t, const float* w_xy, const void* g0, const void* u,
                         const float* coef, void* dv, float* dw_xy, int B, int T, int H, int W, int dtype, hipStream_t s) {
  const Geom g{B, T, H, W};
  STEM_DISPATCH(dv_t, x, w_t, w_xy, g0, u, coef, dv, dw_xy, g, s)
}

int c3d_stem_bwd_wx_mfma(const float* x, const float* w_t, const void* dv, float* dw_t, float* dP, int B, int T, int H,
                         int W, int t_first, int n_frames, int per_sample, int dtype, hipStream_t s) {
  const Geom g{B, T, H, W};
  STEM_DISPATCH(wx_t, x, w_t, dv, dw_t, dP, g, t_first, n_frames, per_sample, s)
}
