// X3D stem (reference model/x3d.py:70-106): spatial Conv3d 1x3x3 (Cin=3 -> 24, pad (0,1,1),
// named `conv_t` upstream) followed by temporal depthwise Conv3d 5x1x1 (pad (2,0,0), named
// `conv_xy`), fused so the 24-channel intermediate never touches HBM.  Input is the logical
// NCDHW f32 clip [B][3][T][H][W] exactly as `torch.cat` builds it (reference
// model/trainer.py:158-162); output is the raw (pre-BN) channels-last tensor [B][T][H][W][24]
// plus per-channel sum / sum-of-squares for the train-mode BatchNorm that follows.
//
// Backward:  stem_bwd_dv   du (BN backward applied on load) -> dv = conv_xy^T(du), d w_xy
//            stem_bwd_wx   dv -> d w_t, and d input for the perception frame(s) summed over
//                          the batch (the only input frames that are parameters,
//                          reference model/trainer.py:51-54,155).
#include "common.h"
#include "stem_mfma.h"
#include <cstdlib>
#include "../../include/change3d_hip.h"

namespace {

constexpr int ST_C = 24;      // stem output channels (X3D-L as used by Change3D)
constexpr int ST_CI = 3;
constexpr int ST_TH = 8, ST_TW = 16;
constexpr int ST_MAXT = 5;

// unconverted 8-channel vectors (kept raw while the loads are in flight)
template <typename T> struct RawS;
template <> struct RawS<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RawS<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};
constexpr int ST_IH = ST_TH + 2, ST_IW = ST_TW + 2;

struct StemGeom { int B, T, H, W; };

__device__ __forceinline__ void load_x_tile(float* xt, const float* __restrict__ x, const StemGeom& g, int b,
                                            int y0, int x0, int tid, int nthr) {
  const int items = ST_CI * g.T * ST_IH * ST_IW;
  for (int i = tid; i < items; i += nthr) {
    const int ix = i % ST_IW;
    int q = i / ST_IW;
    const int iy = q % ST_IH;
    q /= ST_IH;  // q = ci*T + t
    const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
    float v = 0.f;
    if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
      v = x[(((size_t)b * ST_CI * g.T + q) * g.H + gy) * g.W + gx];
    xt[i] = v;
  }
}

// v[t][8] = spatial conv of this thread's pixel for channel vector cv (weights wt[27][24] in LDS)
__device__ __forceinline__ void spatial_conv(float (&v)[ST_MAXT][8], const float* xt, const float* wt, int T,
                                             int py, int px, int cv) {
#pragma unroll
  for (int t = 0; t < ST_MAXT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[t][j] = 0.f;
#pragma unroll
  for (int ci = 0; ci < ST_CI; ++ci) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* wp = wt + (ci * 9 + ky * 3 + kx) * ST_C + cv * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wp);
        const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
        const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int t = 0; t < ST_MAXT; ++t) {
          if (t < T) {
            const float in = xt[((ci * T + t) * ST_IH + py + ky) * ST_IW + px + kx];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[t][j] = fmaf(in, w8[j], v[t][j]);
          }
        }
      }
    }
  }
}

__device__ __forceinline__ void stage_weights(float* wt, float* wxy, const float* __restrict__ w_t,
                                              const float* __restrict__ w_xy, int tid, int nthr) {
  for (int i = tid; i < 27 * ST_C; i += nthr) {  // wt[tap][c] <- w_t[c][tap]
    const int tap = i / ST_C, c = i - tap * ST_C;
    wt[i] = w_t[c * 27 + tap];
  }
  for (int i = tid; i < 5 * ST_C; i += nthr) {  // wxy[dt][c] <- w_xy[c][dt]
    const int dt = i / ST_C, c = i - dt * ST_C;
    wxy[i] = w_xy[c * 5 + dt];
  }
}

constexpr int ST_THREADS = ST_TH * ST_TW * 3;  // (pixel, channel vector)

template <typename T>
__global__ __launch_bounds__(ST_THREADS) void stem_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ w_t,
                                                              const float* __restrict__ w_xy, T* __restrict__ u,
                                                              double* __restrict__ sums, const StemGeom g,
                                                              const int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wt = sm;                    // [27][24]
  float* wxy = wt + 27 * ST_C;       // [5][24]
  float* red = wxy + 5 * ST_C;       // [6 waves][3][16]
  float* xt = red + 6 * 3 * 16;      // [3][T][IH][IW]
  const int tid = threadIdx.x;
  const int tiles_x = (g.W + ST_TW - 1) / ST_TW, tiles_y = (g.H + ST_TH - 1) / ST_TH;
  const int ntiles = tiles_x * tiles_y;
  const int b = blockIdx.y;
  stage_weights(wt, wxy, w_t, w_xy, tid, ST_THREADS);
  const int cv = tid % 3, pix = tid / 3;
  const int px = pix % ST_TW, py = pix / ST_TW;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // a workgroup walks `tiles_per_wg` tiles and flushes the BN statistics once (one tile per workgroup
  // meant 16 k workgroups x 48 same-address f64 atomics per step)
  const int tl0 = blockIdx.x * tiles_per_wg;
  const int tl1 = tl0 + tiles_per_wg < ntiles ? tl0 + tiles_per_wg : ntiles;
  for (int tl = tl0; tl < tl1; ++tl) {
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  __syncthreads();   // the previous tile's x rows are no longer read (and the weights are staged)
  load_x_tile(xt, x, g, b, ty * ST_TH, tx * ST_TW, tid, ST_THREADS);
  __syncthreads();
  const int gy = ty * ST_TH + py, gx = tx * ST_TW + px;
  float v[ST_MAXT][8];
  spatial_conv(v, xt, wt, g.T, py, px, cv);
  const bool ok = gy < g.H && gx < g.W;
#pragma unroll
  for (int t = 0; t < ST_MAXT; ++t) {
    if (t < g.T) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
        const int ti = t + dt - 2;
        if (ti >= 0 && ti < ST_MAXT && ti < g.T) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = fmaf(wxy[dt * ST_C + cv * 8 + j], v[ti][j], o[j]);
        }
      }
      if (ok) {
        Vec8<T>::store(u + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * ST_C + cv * 8, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float r = round_as<T>(o[j]); s1[j] += r; s2[j] += r * r; }
      }
    }
  }
  }  // tile walk
  if (!sums) return;
  // cv = tid % 3 is not lane-periodic in a wave (64 % 3 != 0): reduce through LDS atomics.  The
  // partial sums go in as 2^-40 fixed point: integer adds are associative, so the result does not
  // depend on the order in which the waves arrive (f32 LDS atomics made the BN_stem statistics --
  // and, amplified by 55 train-mode BN layers, every gradient -- differ from run to run).
  unsigned long long* red64 = reinterpret_cast<unsigned long long*>(red);  // [3][16]
  for (int i = tid; i < 3 * 16; i += ST_THREADS) red64[i] = 0ull;
  __syncthreads();
  constexpr float FX = 1099511627776.f;  // 2^40: |partial| < 2^23 is exact down to 2^-40
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    atomicAdd(&red64[cv * 16 + j], (unsigned long long)(long long)(s1[j] * FX));
    atomicAdd(&red64[cv * 16 + 8 + j], (unsigned long long)(long long)(s2[j] * FX));
  }
  __syncthreads();
  if (tid < 48) {
    const int vv = tid / 16, k = tid & 15;
    atomicAdd(sums + (size_t)(k >> 3) * ST_C + vv * 8 + (k & 7), (double)(long long)red64[tid] * (1.0 / 1099511627776.0));
  }
}

// du = A*g0 + B + C*u on load; dv = conv_xy^T(du); d w_xy += sum du[t] * v[t+dt-2]
template <typename T>
__global__ __launch_bounds__(ST_THREADS) void stem_bwd_dv_kernel(
    const float* __restrict__ x, const float* __restrict__ w_t, const float* __restrict__ w_xy,
    const T* __restrict__ g0, const T* __restrict__ u, const float* __restrict__ coef, T* __restrict__ dv,
    float* __restrict__ dw_xy, const StemGeom g, const int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wt = sm;
  float* wxy = wt + 27 * ST_C;
  float* red = wxy + 5 * ST_C;       // [5][24]
  float* xt = red + 5 * ST_C;
  const int tid = threadIdx.x;
  const int tiles_x = (g.W + ST_TW - 1) / ST_TW, tiles_y = (g.H + ST_TH - 1) / ST_TH;
  const int ntiles = tiles_x * tiles_y;
  const int b = blockIdx.y;
  stage_weights(wt, wxy, w_t, w_xy, tid, ST_THREADS);
  for (int i = tid; i < 5 * ST_C; i += ST_THREADS) red[i] = 0.f;
  const int cv = tid % 3, pix = tid / 3;
  const int px = pix % ST_TW, py = pix / ST_TW;
  float cA[8], cB[8], cC[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cA[j] = coef[cv * 8 + j]; cB[j] = coef[ST_C + cv * 8 + j]; cC[j] = coef[2 * ST_C + cv * 8 + j];
  }
  float dwx[5][8];
#pragma unroll
  for (int dt = 0; dt < 5; ++dt)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwx[dt][j] = 0.f;

  int tl0 = blockIdx.x * tiles_per_wg, tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int gy = ty * ST_TH + py, gx = tx * ST_TW + px;
    const bool in_img = gy < g.H && gx < g.W;
    // this pixel's (g0, u) rows are requested BEFORE the x tile is staged: the two global round trips of a tile
    // used to run back to back (one workgroup of 6 waves per CU, nothing else to hide them)
    typename RawS<T>::type gr[ST_MAXT], ur[ST_MAXT];
    if (in_img) {
#pragma unroll
      for (int t = 0; t < ST_MAXT; ++t) {
        if (t < g.T) {
          const size_t off = ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * ST_C + cv * 8;
          gr[t] = RawS<T>::load(g0 + off);
          ur[t] = RawS<T>::load(u + off);
        }
      }
    }
    __syncthreads();
    load_x_tile(xt, x, g, b, ty * ST_TH, tx * ST_TW, tid, ST_THREADS);
    __syncthreads();
    if (!in_img) continue;
    float v[ST_MAXT][8];
    spatial_conv(v, xt, wt, g.T, py, px, cv);
    float du[ST_MAXT][8];
#pragma unroll
    for (int t = 0; t < ST_MAXT; ++t) {
      if (t < g.T) {
        float gg[8], uu[8];
        RawS<T>::cvt(gr[t], gg);
        RawS<T>::cvt(ur[t], uu);
#pragma unroll
        for (int j = 0; j < 8; ++j) du[t][j] = fmaf(cA[j], gg[j], fmaf(cC[j], uu[j], cB[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) du[t][j] = 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < ST_MAXT; ++t) {
      if (t < g.T) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
          // u[to] += wxy[dt]*v[to+dt-2]  =>  dv[t] += wxy[dt]*du[t-dt+2];  dwxy[dt] += du[to]*v[to+dt-2]
          const int to = t - dt + 2;
          if (to >= 0 && to < ST_MAXT && to < g.T) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf(wxy[dt * ST_C + cv * 8 + j], du[to][j], o[j]);
          }
          const int ti = t + dt - 2;
          if (ti >= 0 && ti < ST_MAXT && ti < g.T) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dwx[dt][j] = fmaf(du[t][j], v[ti][j], dwx[dt][j]);
          }
        }
        Vec8<T>::store(dv + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * ST_C + cv * 8, o);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 5; ++dt)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&red[dt * ST_C + cv * 8 + j], dwx[dt][j]);
  __syncthreads();
  for (int i = tid; i < 5 * ST_C; i += ST_THREADS) {
    const int dt = i / ST_C, c = i - dt * ST_C;
    atomicAdd(dw_xy + c * 5 + dt, red[i]);
  }
}

// dv -> d w_t[c][ci][ky][kx] and d(perception frames)[ci][k][y][x] (sum over batch)
// 512 threads: waves 0-1 = one pixel per thread (input gradient), waves 2-7 = four row-slices of
// 96 threads (81 active: tap x channel vector) accumulating d w_t concurrently.  (The first version
// ran the d w_t loop on 81 threads of a 128-thread workgroup: 1.9 ms of a 50 ms step.)
constexpr int SW_PIX = ST_TH * ST_TW;      // 128
constexpr int SW_SLICES = 4, SW_SLICE_T = 96;
constexpr int SW_THREADS = SW_PIX + SW_SLICES * SW_SLICE_T;
template <typename T>
__global__ __launch_bounds__(SW_THREADS) void stem_bwd_wx_kernel(
    const float* __restrict__ x, const float* __restrict__ w_t, const T* __restrict__ dv, float* __restrict__ dw_t,
    float* __restrict__ dP, const StemGeom g, const int t_first, const int n_frames, const int per_sample,
    const int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wt = sm;                               // [27][24]
  float* xt = wt + 27 * ST_C;                   // [3][T][IH][IW]
  float* dt_ = xt + ST_CI * g.T * ST_IH * ST_IW;  // [T][IH][IW][24]  dv with halo
  const int tid = threadIdx.x;
  const int tiles_x = (g.W + ST_TW - 1) / ST_TW, tiles_y = (g.H + ST_TH - 1) / ST_TH;
  const int ntiles = tiles_x * tiles_y;
  for (int i = tid; i < 27 * ST_C; i += SW_THREADS) {
    const int tap = i / ST_C, c = i - tap * ST_C;
    wt[i] = w_t[c * 27 + tap];
  }
  const int px = tid % ST_TW, py = (tid / ST_TW) % ST_TH;
  // weight-gradient ownership: slice thread l < 81 owns (tap = l / 3, channel vector = l % 3) for the
  // output rows [slice * ST_TH / SW_SLICES, ...) of every tile
  const int wl = tid >= SW_PIX ? (tid - SW_PIX) % SW_SLICE_T : SW_SLICE_T, wslice = tid >= SW_PIX ? (tid - SW_PIX) / SW_SLICE_T : 0;
  const bool w_role = wl < 81;
  const int wtap = w_role ? wl / 3 : 0, wcv = wl % 3;
  const int wci = wtap / 9, wky = (wtap % 9) / 3, wkx = wtap % 3;
  float dwacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dwacc[j] = 0.f;
  // A workgroup owns ONE spatial tile and walks the samples b = blockIdx.y, +gridDim.y, ...: the
  // batch-summed input gradient and d w_t stay in registers and are flushed once (the per-sample
  // version issued 6 M + 5 M same-address f32 atomics per step and was bound by them).
  float dxacc[ST_MAXT][ST_CI];
#pragma unroll
  for (int k = 0; k < ST_MAXT; ++k)
#pragma unroll
    for (int ci = 0; ci < ST_CI; ++ci) dxacc[k][ci] = 0.f;
  const int tl = blockIdx.x;
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  const int y0 = ty * ST_TH, x0 = tx * ST_TW;
  for (int b = blockIdx.y; b < g.B; b += gridDim.y) {
    __syncthreads();
    load_x_tile(xt, x, g, b, y0, x0, tid, SW_THREADS);
    for (int i = tid; i < g.T * ST_IH * ST_IW * 3; i += SW_THREADS) {
      const int cvv = i % 3;
      int q = i / 3;
      const int ix = q % ST_IW;
      q /= ST_IW;
      const int iy = q % ST_IH, t = q / ST_IH;
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      float f[8];
      if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
        Vec8<T>::load(dv + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * ST_C + cvv * 8, f);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
      }
      Vec8<float>::store(dt_ + ((size_t)(t * ST_IH + iy) * ST_IW + ix) * ST_C + cvv * 8, f);
    }
    __syncthreads();
    // ---- d input for the perception frames --------------------------------------------------
    const int gy = y0 + py, gx = x0 + px;
    if (dP && tid < SW_PIX && gy < g.H && gx < g.W) {
#pragma unroll 1
      for (int k = 0; k < n_frames; ++k) {
        const int t = t_first + k;
        float dx[ST_CI] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
          for (int kx = 0; kx < 3; ++kx) {
            // out pixel q = p - (k - 1)  =>  local tile coords (py + 1 - (ky - 1)) ...
            const float* dp = dt_ + ((size_t)(t * ST_IH + py + 2 - ky) * ST_IW + px + 2 - kx) * ST_C;
#pragma unroll
            for (int c = 0; c < ST_C; ++c) {
              const float d = dp[c];
#pragma unroll
              for (int ci = 0; ci < ST_CI; ++ci) dx[ci] = fmaf(d, wt[(ci * 9 + ky * 3 + kx) * ST_C + c], dx[ci]);
            }
          }
        }
#pragma unroll
        for (int ci = 0; ci < ST_CI; ++ci) {
          if (per_sample)  // dP is a full NCDHW gradient [B][3][T][H][W]
            dP[((((size_t)b * ST_CI + ci) * g.T + t) * g.H + gy) * g.W + gx] = dx[ci];
        }
        if (!per_sample) {  // dP is [3][n_frames][H][W], summed over the batch: flushed after the walk
#pragma unroll
          for (int kk = 0; kk < ST_MAXT; ++kk)  // static register indexing (k is a runtime value)
#pragma unroll
            for (int ci = 0; ci < ST_CI; ++ci) dxacc[kk][ci] += kk == k ? dx[ci] : 0.f;
        }
      }
    }
    // ---- d w_t ---------------------------------------------------------------------------------
    if (w_role) {
      constexpr int ROWS = ST_TH / SW_SLICES;
      for (int t = 0; t < g.T; ++t) {
        for (int qy = wslice * ROWS; qy < (wslice + 1) * ROWS; ++qy) {
          for (int qx = 0; qx < ST_TW; ++qx) {
            const float xin = xt[((wci * g.T + t) * ST_IH + qy + wky) * ST_IW + qx + wkx];
            const float* dp = dt_ + ((size_t)(t * ST_IH + qy + 1) * ST_IW + qx + 1) * ST_C + wcv * 8;
            const float4 d0 = *reinterpret_cast<const float4*>(dp);
            const float4 d1 = *reinterpret_cast<const float4*>(dp + 4);
            dwacc[0] = fmaf(d0.x, xin, dwacc[0]); dwacc[1] = fmaf(d0.y, xin, dwacc[1]);
            dwacc[2] = fmaf(d0.z, xin, dwacc[2]); dwacc[3] = fmaf(d0.w, xin, dwacc[3]);
            dwacc[4] = fmaf(d1.x, xin, dwacc[4]); dwacc[5] = fmaf(d1.y, xin, dwacc[5]);
            dwacc[6] = fmaf(d1.z, xin, dwacc[6]); dwacc[7] = fmaf(d1.w, xin, dwacc[7]);
          }
        }
      }
    }
  }
  if (dP && !per_sample && tid < SW_PIX) {
    const int gy = y0 + py, gx = x0 + px;
    if (gy < g.H && gx < g.W) {
#pragma unroll
      for (int k = 0; k < ST_MAXT; ++k) {
        if (k < n_frames) {
#pragma unroll
          for (int ci = 0; ci < ST_CI; ++ci) {
            float* dst = dP + (((size_t)ci * n_frames + k) * g.H + gy) * g.W + gx;
            if (gridDim.y == 1) *dst += dxacc[k][ci];     // sole owner of this pixel
            else atomicAdd(dst, dxacc[k][ci]);
          }
        }
      }
    }
  }
  // d w_t: add the four row-slices through LDS, then one atomic per weight and workgroup
  __syncthreads();
  float* wred = dt_;  // [SW_SLICES - 1][81][8]
  if (w_role && wslice > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wred[((wslice - 1) * 81 + wl) * 8 + j] = dwacc[j];
  }
  __syncthreads();
  if (w_role && wslice == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = dwacc[j];
      for (int sl = 0; sl < SW_SLICES - 1; ++sl) v += wred[(sl * 81 + wl) * 8 + j];
      atomicAdd(dw_t + (size_t)(wcv * 8 + j) * 27 + wtap, v);
    }
  }
}

}  // namespace

extern "C" int c3d_stem_fwd(const float* x, const float* w_t, const float* w_xy, void* u, double* sums, int32_t B,
                            int32_t T, int32_t H, int32_t W, int32_t dtype, void* stream) {
  if (!x || !w_t || !w_xy || !u || B <= 0 || T <= 0 || T > ST_MAXT || H <= 0 || W <= 0) return C3D_E_BADARG;
  if (c3d_stem_mfma_enabled()) {
    const int rc = c3d_stem_fwd_mfma(x, w_t, w_xy, u, sums, B, T, H, W, dtype, reinterpret_cast<hipStream_t>(stream));
    if (rc != C3D_E_UNSUPPORTED) { if (rc == 0) C3D_CHECK_LAUNCH(); return rc; }
  }
  StemGeom g{B, T, H, W};
  const size_t lds = (27 * ST_C + 5 * ST_C + 6 * 3 * 16 + (size_t)ST_CI * T * ST_IH * ST_IW) * sizeof(float);
  const int ntiles = ((W + ST_TW - 1) / ST_TW) * ((H + ST_TH - 1) / ST_TH);
  static const int env_tpw = c3d_env("C3D_STEM_FWD_TPW") ? atoi(c3d_env("C3D_STEM_FWD_TPW")) : 0;   // tuning knob
  int tpw = 16;
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * B < 1024) tpw >>= 1;   // keep ~4 workgroups per CU
  if (env_tpw > 0) tpw = env_tpw;
  dim3 grid((ntiles + tpw - 1) / tpw, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32)
    stem_fwd_kernel<float><<<grid, ST_THREADS, lds, s>>>(x, w_t, w_xy, (float*)u, sums, g, tpw);
  else if (dtype == C3D_DT_BF16)
    stem_fwd_kernel<bf16_t><<<grid, ST_THREADS, lds, s>>>(x, w_t, w_xy, (bf16_t*)u, sums, g, tpw);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_stem_bwd_dv(const float* x, const float* w_t, const float* w_xy, const void* g0, const void* u,
                               const float* coef, void* dv, float* dw_xy, int32_t B, int32_t T, int32_t H,
                               int32_t W, int32_t dtype, void* stream) {
  if (!x || !w_t || !w_xy || !g0 || !u || !coef || !dv || !dw_xy || B <= 0 || T <= 0 || T > ST_MAXT)
    return C3D_E_BADARG;
  if (c3d_stem_mfma_enabled()) {
    const int rc = c3d_stem_bwd_dv_mfma(x, w_t, w_xy, g0, u, coef, dv, dw_xy, B, T, H, W, dtype, reinterpret_cast<hipStream_t>(stream));
    if (rc != C3D_E_UNSUPPORTED) { if (rc == 0) C3D_CHECK_LAUNCH(); return rc; }
  }
  StemGeom g{B, T, H, W};
  const size_t lds = (27 * ST_C + 5 * ST_C + 5 * ST_C + (size_t)ST_CI * T * ST_IH * ST_IW) * sizeof(float);
  const int ntiles = ((W + ST_TW - 1) / ST_TW) * ((H + ST_TH - 1) / ST_TH);
  // one workgroup (6 waves, ~216 VGPRs) fits per CU and each ends with 120 same-address atomics: walks as long as
  // 2 workgroups per CU allow
  static const int env_tpw = c3d_env("C3D_STEM_DV_TPW") ? atoi(c3d_env("C3D_STEM_DV_TPW")) : 0;   // tuning knob
  int tpw = 64;
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * B < 2L * 256) tpw >>= 1;
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid((ntiles + tpw - 1) / tpw, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32)
    stem_bwd_dv_kernel<float><<<grid, ST_THREADS, lds, s>>>(x, w_t, w_xy, (const float*)g0, (const float*)u, coef,
                                                             (float*)dv, dw_xy, g, tpw);
  else if (dtype == C3D_DT_BF16)
    stem_bwd_dv_kernel<bf16_t><<<grid, ST_THREADS, lds, s>>>(x, w_t, w_xy, (const bf16_t*)g0, (const bf16_t*)u,
                                                              coef, (bf16_t*)dv, dw_xy, g, tpw);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_stem_bwd_wx(const float* x, const float* w_t, const void* dv, float* dw_t, float* dP, int32_t B,
                               int32_t T, int32_t H, int32_t W, int32_t t_first, int32_t n_frames, int32_t per_sample,
                               int32_t dtype, void* stream) {
  if (!x || !w_t || !dv || !dw_t || B <= 0 || T <= 0 || T > ST_MAXT) return C3D_E_BADARG;
  if (dP && (t_first < 0 || n_frames <= 0 || t_first + n_frames > T)) return C3D_E_BADARG;
  if (c3d_stem_mfma_enabled()) {
    const int rc = c3d_stem_bwd_wx_mfma(x, w_t, dv, dw_t, dP, B, T, H, W, t_first, n_frames, per_sample, dtype,
                                        reinterpret_cast<hipStream_t>(stream));
    if (rc != C3D_E_UNSUPPORTED) { if (rc == 0) C3D_CHECK_LAUNCH(); return rc; }
  }
  StemGeom g{B, T, H, W};
  const size_t lds = (27 * ST_C + (size_t)ST_CI * T * ST_IH * ST_IW + (size_t)T * ST_IH * ST_IW * ST_C) * sizeof(float);
  const int ntiles = ((W + ST_TW - 1) / ST_TW) * ((H + ST_TH - 1) / ST_TH);
  // one workgroup per tile walking the batch; split the batch only when there are too few tiles
  int bsplit = (2 * 256 + ntiles - 1) / ntiles;
  if (bsplit > B) bsplit = B;
  if (bsplit < 1) bsplit = 1;
  const int tpw = 1;
  dim3 grid(ntiles, bsplit);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_wx_kernel<float>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_wx_kernel<bf16_t>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    attr_set = true;
  }
  if (dtype == C3D_DT_F32)
    stem_bwd_wx_kernel<float><<<grid, SW_THREADS, lds, s>>>(x, w_t, (const float*)dv, dw_t, dP, g, t_first,
                                                             n_frames, per_sample, tpw);
  else if (dtype == C3D_DT_BF16)
    stem_bwd_wx_kernel<bf16_t><<<grid, SW_THREADS, lds, s>>>(x, w_t, (const bf16_t*)dv, dw_t, dP, g, t_first,
                                                              n_frames, per_sample, tpw);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}
