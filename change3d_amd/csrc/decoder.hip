// ChangeDecoder kernels (reference model/change_decoder.py:30-55, 68-81), channels-last:
//   * ConvTranspose2d k=4 s=2 p=1 (+bias) fused with the top-down skip add
//       out[b,oy,ox,co] = bias[co] + skip[b,oy,ox,co] + sum_{ci,ky,kx} in[b,iy,ix,ci] W[ci][co][ky][kx],
//       oy = 2 iy - 1 + ky.  Every output pixel sees exactly 2x2 taps, selected by its parity
//       class, so one workgroup handles ONE parity class: its 4 tap matrices sit in LDS and no
//       lane diverges.
//   * its data gradient (a stride-2 gather over all 16 taps, tap groups staged by parity class)
//   * final Conv2d 3x3 (24 -> num_class <= 8, no bias) + optional sigmoid, forward / backward.
// The weight gradient of the transposed conv runs on the MFMA weight-gradient kernel
// (c3d_pw_wgrad with C3D_ROWS_S2SHIFT addressing), the bias gradient on c3d_col_sum.
#include "common.h"
#include "launch_hints.h"
#include <cstdlib>
#include "../../include/change3d_hip.h"

namespace {

constexpr int CT_QH = 8, CT_QW = 16;  // tile of input-resolution positions; 2 adjacent qx per thread

// Parity class (py,px) -> its 2 ky taps: py=0: ky in {1,3} (iy = qy, qy-1); py=1: ky in {0,2} (iy = qy+1, qy)
__device__ __forceinline__ void tap_of(int par, int j, int& k, int& d) {
  // par: output parity; j in {0,1}; k: kernel index; d: input offset relative to q
  if (par == 0) { k = j ? 3 : 1; d = j ? -1 : 0; }
  else { k = j ? 2 : 0; d = j ? 0 : 1; }
}

template <typename T>
__global__ void convT_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                 const float* __restrict__ bias, const T* __restrict__ skip, int64_t skip_bstride,
                                 T* __restrict__ out, int B, int h, int wd, int C) {
  // grid: (tiles, 4 parity classes, B); block: (CT_QH*CT_QW/2) * G threads
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [4 taps][C ci][C co]
  const int G = C >> 3;
  const int nthr = blockDim.x;
  const int tid = threadIdx.x;
  const int par = blockIdx.y, py = par >> 1, pxp = par & 1;
  const int b = blockIdx.z;
  for (int i = tid; i < 4 * C * C; i += nthr) {
    const int co = i % C;
    int q = i / C;
    const int ci = q % C;
    const int tp = q / C;  // tp = jy*2 + jx
    int ky, kx, dy, dx;
    tap_of(py, tp >> 1, ky, dy);
    tap_of(pxp, tp & 1, kx, dx);
    wl[i] = w[(((size_t)ci * C + co) * 4 + ky) * 4 + kx];
  }
  __syncthreads();
  const int tiles_x = (wd + CT_QW - 1) / CT_QW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int cv = tid % G;
  const int pp = tid / G;                 // pixel pair index in tile
  const int qx0 = tx * CT_QW + (pp % (CT_QW / 2)) * 2;
  const int qy = ty * CT_QH + pp / (CT_QW / 2);
  if (qy >= h) return;
  const int H = 2 * h, W = 2 * wd;
  float acc[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[p][j] = bias[cv * 8 + j];
  for (int tp = 0; tp < 4; ++tp) {
    int ky, kx, dy, dx;
    tap_of(py, tp >> 1, ky, dy);
    tap_of(pxp, tp & 1, kx, dx);
    const int iy = qy + dy;
    if (iy < 0 || iy >= h) continue;
    const float* wt = wl + (size_t)tp * C * C + cv * 8;
    for (int cg = 0; cg < G; ++cg) {
      float v0[8], v1[8];
      const int ix0 = qx0 + dx, ix1 = qx0 + 1 + dx;
      const bool ok0 = ix0 >= 0 && ix0 < wd, ok1 = ix1 >= 0 && ix1 < wd;
      if (ok0) Vec8<T>::load(in + (((size_t)b * h + iy) * wd + ix0) * C + cg * 8, v0);
      if (ok1) Vec8<T>::load(in + (((size_t)b * h + iy) * wd + ix1) * C + cg * 8, v1);
#pragma unroll
      for (int j = 0; j < 8; ++j) { if (!ok0) v0[j] = 0.f; if (!ok1) v1[j] = 0.f; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 w0 = *reinterpret_cast<const float4*>(wt + (size_t)(cg * 8 + e) * C);
        const float4 w1 = *reinterpret_cast<const float4*>(wt + (size_t)(cg * 8 + e) * C + 4);
        const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[0][j] = fmaf(v0[e], w8[j], acc[0][j]);
          acc[1][j] = fmaf(v1[e], w8[j], acc[1][j]);
        }
      }
    }
  }
  const int oy = 2 * qy + py;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int qx = qx0 + p;
    if (qx >= wd) continue;
    const int ox = 2 * qx + pxp;
    if (skip) {
      float s[8];
      Vec8<T>::load(skip + (size_t)b * skip_bstride + ((size_t)oy * W + ox) * C + cv * 8, s);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[p][j] += s[j];
    }
    Vec8<T>::store(out + (((size_t)b * H + oy) * W + ox) * C + cv * 8, acc[p]);
  }
}

// din[b,iy,ix,ci] = sum_{co,ky,kx} dout[b, 2iy-1+ky, 2ix-1+kx, co] * W[ci][co][ky][kx]
template <typename T>
__global__ void convT_bwd_data_kernel(const T* __restrict__ dout, const float* __restrict__ w,
                                      T* __restrict__ din, int B, int h, int wd, int C) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [4 taps][C co][C ci]
  const int G = C >> 3;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const int b = blockIdx.y;
  const int tiles_x = (wd + CT_QW - 1) / CT_QW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int cv = tid % G, pix = tid / G;
  const int ix = tx * CT_QW + pix % CT_QW, iy = ty * CT_QH + pix / CT_QW;
  const int H = 2 * h, W = 2 * wd;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int grp = 0; grp < 4; ++grp) {  // tap group: ky in {gy, gy+2}, kx in {gx, gx+2}
    const int gy = grp >> 1, gx = grp & 1;
    __syncthreads();
    for (int i = tid; i < 4 * C * C; i += nthr) {
      const int ci = i % C;
      int q = i / C;
      const int co = q % C;
      const int tp = q / C;
      const int ky = gy + 2 * (tp >> 1), kx = gx + 2 * (tp & 1);
      wl[i] = w[(((size_t)ci * C + co) * 4 + ky) * 4 + kx];
    }
    __syncthreads();
    if (iy < h && ix < wd) {
      for (int tp = 0; tp < 4; ++tp) {
        const int ky = gy + 2 * (tp >> 1), kx = gx + 2 * (tp & 1);
        const int oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
        if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
        const float* wt = wl + (size_t)tp * C * C + cv * 8;
        const T* dp = dout + (((size_t)b * H + oy) * W + ox) * C;
        for (int cg = 0; cg < G; ++cg) {
          float v[8];
          Vec8<T>::load(dp + cg * 8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float4 w0 = *reinterpret_cast<const float4*>(wt + (size_t)(cg * 8 + e) * C);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + (size_t)(cg * 8 + e) * C + 4);
            acc[0] = fmaf(v[e], w0.x, acc[0]); acc[1] = fmaf(v[e], w0.y, acc[1]);
            acc[2] = fmaf(v[e], w0.z, acc[2]); acc[3] = fmaf(v[e], w0.w, acc[3]);
            acc[4] = fmaf(v[e], w1.x, acc[4]); acc[5] = fmaf(v[e], w1.y, acc[5]);
            acc[6] = fmaf(v[e], w1.z, acc[6]); acc[7] = fmaf(v[e], w1.w, acc[7]);
          }
        }
      }
    }
  }
  if (iy < h && ix < wd) Vec8<T>::store(din + (((size_t)b * h + iy) * wd + ix) * C + cv * 8, acc);
}

// column sums of a dense [M][Cp] tensor: out[c] += sum_m x[m][c]   (ConvTranspose bias gradient)
template <typename T>
__global__ void col_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t nvec, int G, int C) {
  extern __shared__ float red[];
  const int v = threadIdx.x % G;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float f[8];
    Vec8<T>::load(x + i * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = s[j];
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * 8; idx += blockDim.x) {
    const int vv = idx / 8, k = idx % 8;
    float acc = 0.f;
    for (int t = vv; t < (int)blockDim.x; t += G) acc += red[t * 8 + k];
    if (vv * 8 + k < C) atomicAdd(out + vv * 8 + k, acc);
  }
}

// ---- final 3x3 conv (C=24 -> NC<=8) + sigmoid --------------------------------------------------
constexpr int HD_C = 24, HD_MAXNC = 8;
constexpr int HD_TH = 8, HD_TW = 32;

template <typename T>
__global__ __launch_bounds__(HD_TH * HD_TW) void head_fwd_kernel(const T* __restrict__ x,
                                                                 const float* __restrict__ w,
                                                                 float* __restrict__ out, int B, int H, int W,
                                                                 int NC, int has_sigmoid) {
  __shared__ float wl[9 * HD_C * HD_MAXNC];  // [tap][c][n]
  const int tid = threadIdx.x;
  for (int i = tid; i < 9 * HD_C * HD_MAXNC; i += HD_TH * HD_TW) {
    const int n = i % HD_MAXNC;
    int q = i / HD_MAXNC;
    const int c = q % HD_C, tap = q / HD_C;
    wl[i] = (n < NC) ? w[((size_t)n * HD_C + c) * 9 + tap] : 0.f;
  }
  __syncthreads();
  const int tiles_x = (W + HD_TW - 1) / HD_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int ox = tx * HD_TW + tid % HD_TW, oy = ty * HD_TH + tid / HD_TW;
  if (ox >= W || oy >= H) return;
  float acc[HD_MAXNC];
#pragma unroll
  for (int n = 0; n < HD_MAXNC; ++n) acc[n] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy + ky - 1;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox + kx - 1;
      if (ix < 0 || ix >= W) continue;
      const T* xp = x + (((size_t)b * H + iy) * W + ix) * HD_C;
#pragma unroll
      for (int cg = 0; cg < 3; ++cg) {
        float v[8];
        Vec8<T>::load(xp + cg * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float* wp = wl + ((ky * 3 + kx) * HD_C + cg * 8 + e) * HD_MAXNC;
          if (NC == 1) acc[0] = fmaf(v[e], wp[0], acc[0]);
          else {
#pragma unroll
            for (int n = 0; n < HD_MAXNC; ++n) acc[n] = fmaf(v[e], wp[n], acc[n]);
          }
        }
      }
    }
  }
  for (int n = 0; n < NC; ++n) {
    float r = acc[n];
    if (has_sigmoid) r = 1.0f / (1.0f + expf(-r));
    out[(((size_t)b * NC + n) * H + oy) * W + ox] = r;
  }
}

// dlogit = has_sigmoid ? dout * p * (1-p) : dout ;  dx[b,y,x,c] = sum_{n,k} dlogit[b,n,y-ky+1,x-kx+1] W[n][c][k]
template <typename T>
__global__ __launch_bounds__(HD_TH * HD_TW) void head_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ prob, const T* __restrict__ x,
    const float* __restrict__ w, T* __restrict__ dx, float* __restrict__ dw, int B, int H, int W, int NC,
    int has_sigmoid, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) float wl[9 * HD_C * HD_MAXNC];
  __shared__ float dl[HD_MAXNC][HD_TH + 2][HD_TW + 2];
  __shared__ float xt[HD_TH + 2][HD_TW + 2][HD_C + 1];
  const int tid = threadIdx.x;
  constexpr int NTHR = HD_TH * HD_TW;
  // [n][tap][c]: the data-gradient loop below reads the 24 channel weights of a (class, tap) as six 16-byte vectors
  // (the [tap][c][n] layout of the forward kernel made them 24 scalar LDS reads per tap and pixel)
  for (int i = tid; i < 9 * HD_C * HD_MAXNC; i += NTHR) {
    const int c = i % HD_C;
    int q = i / HD_C;
    const int tap = q % 9, n = q / 9;
    wl[i] = (n < NC) ? w[((size_t)n * HD_C + c) * 9 + tap] : 0.f;
  }
  const int tiles_x = (W + HD_TW - 1) / HD_TW, tiles_y = (H + HD_TH - 1) / HD_TH;
  const int ntiles = tiles_x * tiles_y;
  const int b = blockIdx.y;
  // Weight gradient: thread = (pixel slice of 9, tap, channel vector) keeps dW[n][tap][8 channels] for every class n.
  // (It used to be 27*NC threads each walking all 256 pixels of a tile while the other waves idled, and 216 global
  // atomics per 4-tile workgroup onto the same 216 addresses.)
  constexpr int HD_NSL = 9;
  const int wown = tid % 27, wsl = tid / 27;      // wsl < 9 for tid < 243
  const int wtap = wown / 3, wcv = wown % 3;
  const int wky = wtap / 3, wkx = wtap % 3;
  float dwacc[HD_MAXNC][8];
#pragma unroll
  for (int n = 0; n < HD_MAXNC; ++n)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[n][j] = 0.f;
  int tl0 = blockIdx.x * tiles_per_wg, tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int y0 = ty * HD_TH, x0 = tx * HD_TW;
    __syncthreads();
    for (int i = tid; i < NC * (HD_TH + 2) * (HD_TW + 2); i += NTHR) {
      const int lx = i % (HD_TW + 2);
      int q = i / (HD_TW + 2);
      const int ly = q % (HD_TH + 2), n = q / (HD_TH + 2);
      const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const size_t o = (((size_t)b * NC + n) * H + gy) * W + gx;
        v = dout[o];
        if (has_sigmoid) { const float p = prob[o]; v *= p * (1.f - p); }
      }
      dl[n][ly][lx] = v;
    }
    for (int i = tid; i < (HD_TH + 2) * (HD_TW + 2) * 3; i += NTHR) {
      const int cg = i % 3;
      int q = i / 3;
      const int lx = q % (HD_TW + 2), ly = q / (HD_TW + 2);
      const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
      float f[8];
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) Vec8<T>::load(x + (((size_t)b * H + gy) * W + gx) * HD_C + cg * 8, f);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) xt[ly][lx][cg * 8 + j] = f[j];
    }
    __syncthreads();
    // ---- dx for this thread's pixel ---------------------------------------------------------
    const int px = tid % HD_TW, py = tid / HD_TW;
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
      float acc[HD_C];
#pragma unroll
      for (int c = 0; c < HD_C; ++c) acc[c] = 0.f;
      for (int n = 0; n < NC; ++n) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float d = dl[n][py + 2 - ky][px + 2 - kx];  // output pixel (y-ky+1, x-kx+1)
#pragma unroll
            for (int c4 = 0; c4 < HD_C / 4; ++c4) {
              const float4 wv = *reinterpret_cast<const float4*>(wl + (n * 9 + ky * 3 + kx) * HD_C + c4 * 4);
              acc[c4 * 4 + 0] = fmaf(d, wv.x, acc[c4 * 4 + 0]); acc[c4 * 4 + 1] = fmaf(d, wv.y, acc[c4 * 4 + 1]);
              acc[c4 * 4 + 2] = fmaf(d, wv.z, acc[c4 * 4 + 2]); acc[c4 * 4 + 3] = fmaf(d, wv.w, acc[c4 * 4 + 3]);
            }
          }
        }
      }
#pragma unroll
      for (int cg = 0; cg < 3; ++cg) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = acc[cg * 8 + j];
        Vec8<T>::store(dx + (((size_t)b * H + gy) * W + gx) * HD_C + cg * 8, o);
      }
    }
    // ---- dW ----------------------------------------------------------------------------------
    if (tid < 27 * HD_NSL) {
      for (int pi = wsl; pi < HD_TH * HD_TW; pi += HD_NSL) {
        const int qy = pi / HD_TW, qx = pi % HD_TW;
        const float* xp = &xt[qy + wky][qx + wkx][wcv * 8];
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xp[j];
#pragma unroll
        for (int n = 0; n < HD_MAXNC; ++n) {
          if (n < NC) {
            const float d = dl[n][qy + 1][qx + 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) dwacc[n][j] = fmaf(d, xv[j], dwacc[n][j]);
          }
        }
      }
    }
  }
  // slices -> LDS (the weight copy is dead now and has exactly 27 x 8 x 8 floats) -> one global atomic per value.  The nine
  // pixel slices add their partial sums one after the other with plain LDS read-modify-writes (the 27 owners of a slice hold
  // distinct addresses): fixed order, and no LDS float atomics -- ds_add_f32 costs ~190 clocks per wave-instruction on
  // gfx950 (tools/micro/lds_atomic_rate.hip), the 56 of them per thread were a third of this kernel on the 7-class heads
  __syncthreads();
  for (int i = tid; i < 27 * HD_MAXNC * 8; i += NTHR) wl[i] = 0.f;
  for (int sl = 0; sl < HD_NSL; ++sl) {
    __syncthreads();
    if (tid < 27 * HD_NSL && wsl == sl) {
#pragma unroll
      for (int n = 0; n < HD_MAXNC; ++n) {
        if (n < NC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wl[(wown * HD_MAXNC + n) * 8 + j] += dwacc[n][j];
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 27 * HD_MAXNC * 8; i += NTHR) {
    const int j = i & 7, n = (i >> 3) % HD_MAXNC, own = i / (8 * HD_MAXNC);
    if (n < NC) atomicAdd(dw + ((size_t)n * HD_C + (own % 3) * 8 + j) * 9 + own / 3, wl[i]);
  }
}

}  // namespace

// bf16 (throughput) path: MFMA kernels in convt_mfma.hip (c3d_set_option(C3D_OPT_CONVT_MFMA, 0) keeps the scalar kernels: parity tests)
int c3d_detail_convt_fwd_bf16(const void* in, const float* w, const float* bias, const void* skip, int64_t skip_bstride, void* out,
                              int B, int h, int wd, int C, hipStream_t st);
int c3d_detail_convt_bwd_data_bf16(const void* dout, const float* w, void* din, int B, int h, int wd, int C, hipStream_t st);
static bool convt_mfma_on() { return c3d_option_convt_mfma != 0; }
// head 3x3 on the matrix cores (head_mfma.hip; same option: the scalar kernels stay the parity reference)
int c3d_detail_head_fwd_bf16(const void* x, const float* w, float* out, int B, int H, int W, int NC, int has_sigmoid, hipStream_t s);
int c3d_detail_head_bwd_bf16(const float* dout, const float* prob, const void* x, const float* w, void* dx, float* dw, float* ws,
                             int B, int H, int W, int NC, int has_sigmoid, hipStream_t s);
int64_t c3d_detail_head_ws_floats(int B, int H, int W, int NC);

extern "C" int c3d_convT4s2_fwd(const void* in, const float* w, const float* bias, const void* skip,
                                int64_t skip_bstride, void* out, int32_t B, int32_t h, int32_t wd, int32_t C,
                                int32_t dtype, void* stream) {
  if (!in || !w || !bias || !out || B <= 0 || h <= 0 || wd <= 0 || (C & 7) || C > 96) return C3D_E_BADARG;
  if (dtype == C3D_DT_BF16 && convt_mfma_on()) {
    const int rc = c3d_detail_convt_fwd_bf16(in, w, bias, skip, skip_bstride, out, B, h, wd, C, reinterpret_cast<hipStream_t>(stream));
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  const int G = C / 8;
  const size_t lds = (size_t)4 * C * C * sizeof(float);
  const int nthr = (CT_QH * CT_QW / 2) * G;
  dim3 grid(((wd + CT_QW - 1) / CT_QW) * ((h + CT_QH - 1) / CT_QH), 4, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_fwd_kernel<float>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_fwd_kernel<bf16_t>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (dtype == C3D_DT_F32)
    convT_fwd_kernel<float><<<grid, nthr, lds, s>>>((const float*)in, w, bias, (const float*)skip, skip_bstride,
                                                     (float*)out, B, h, wd, C);
  else if (dtype == C3D_DT_BF16)
    convT_fwd_kernel<bf16_t><<<grid, nthr, lds, s>>>((const bf16_t*)in, w, bias, (const bf16_t*)skip, skip_bstride,
                                                      (bf16_t*)out, B, h, wd, C);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_convT4s2_bwd_data(const void* dout, const float* w, void* din, int32_t B, int32_t h, int32_t wd,
                                     int32_t C, int32_t dtype, void* stream) {
  if (!dout || !w || !din || B <= 0 || h <= 0 || wd <= 0 || (C & 7) || C > 96) return C3D_E_BADARG;
  if (dtype == C3D_DT_BF16 && convt_mfma_on()) {
    const int rc = c3d_detail_convt_bwd_data_bf16(dout, w, din, B, h, wd, C, reinterpret_cast<hipStream_t>(stream));
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  const int G = C / 8;
  const size_t lds = (size_t)4 * C * C * sizeof(float);
  const int nthr = CT_QH * CT_QW * G;
  if (nthr > 1024) return C3D_E_UNSUPPORTED;
  dim3 grid(((wd + CT_QW - 1) / CT_QW) * ((h + CT_QH - 1) / CT_QH), B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_bwd_data_kernel<float>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_bwd_data_kernel<bf16_t>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (dtype == C3D_DT_F32)
    convT_bwd_data_kernel<float><<<grid, nthr, lds, s>>>((const float*)dout, w, (float*)din, B, h, wd, C);
  else if (dtype == C3D_DT_BF16)
    convT_bwd_data_kernel<bf16_t><<<grid, nthr, lds, s>>>((const bf16_t*)dout, w, (bf16_t*)din, B, h, wd, C);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_col_sum(const void* x, float* out, int64_t M, int32_t C, int32_t Cp, int32_t dtype,
                           void* stream) {
  if (!x || !out || M <= 0 || (Cp & 7) || Cp > 1024 || C > Cp) return C3D_E_BADARG;   // (<= 1024: vocabulary logits)
  const int G = Cp / 8, blk = G * (256 / G);
  const int64_t nvec = M * G;
  int64_t grid = (nvec + blk - 1) / blk;
  if (grid > 512) grid = 512;
  const size_t lds = (size_t)blk * 8 * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32) col_sum_kernel<float><<<(int)grid, blk, lds, s>>>((const float*)x, out, nvec, G, C);
  else if (dtype == C3D_DT_BF16) col_sum_kernel<bf16_t><<<(int)grid, blk, lds, s>>>((const bf16_t*)x, out, nvec, G, C);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_head3x3_fwd(const void* x, const float* w, float* out, int32_t B, int32_t H, int32_t W,
                               int32_t C, int32_t NC, int32_t has_sigmoid, int32_t dtype, void* stream) {
  if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || C != HD_C || NC <= 0 || NC > HD_MAXNC) return C3D_E_BADARG;
  dim3 grid(((W + HD_TW - 1) / HD_TW) * ((H + HD_TH - 1) / HD_TH), B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_BF16 && convt_mfma_on()) return c3d_detail_head_fwd_bf16(x, w, out, B, H, W, NC, has_sigmoid, s);   // head_mfma.hip
  if (dtype == C3D_DT_F32)
    head_fwd_kernel<float><<<grid, HD_TH * HD_TW, 0, s>>>((const float*)x, w, out, B, H, W, NC, has_sigmoid);
  else if (dtype == C3D_DT_BF16)
    head_fwd_kernel<bf16_t><<<grid, HD_TH * HD_TW, 0, s>>>((const bf16_t*)x, w, out, B, H, W, NC, has_sigmoid);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t c3d_head3x3_bwd_ws_floats(int32_t B, int32_t H, int32_t W, int32_t NC) {
  return (B > 0 && H > 0 && W > 0 && NC > 0) ? c3d_detail_head_ws_floats(B, H, W, NC) : 0;
}

extern "C" int c3d_head3x3_bwd(const float* dout, const float* prob, const void* x, const float* w, void* dx,
                               float* dw, float* ws, int32_t B, int32_t H, int32_t W, int32_t C, int32_t NC,
                               int32_t has_sigmoid, int32_t dtype, void* stream) {
  if (!dout || !x || !w || !dx || !dw || B <= 0 || C != HD_C || NC <= 0 || NC > HD_MAXNC) return C3D_E_BADARG;
  if (has_sigmoid && !prob) return C3D_E_BADARG;
  if (dtype == C3D_DT_BF16 && convt_mfma_on())
    return c3d_detail_head_bwd_bf16(dout, prob, x, w, dx, dw, ws, B, H, W, NC, has_sigmoid, reinterpret_cast<hipStream_t>(stream));
  const int ntiles = ((W + HD_TW - 1) / HD_TW) * ((H + HD_TH - 1) / HD_TH);
  static const int env_tpw = c3d_env("C3D_HEAD_TPW") ? atoi(c3d_env("C3D_HEAD_TPW")) : 0;   // tuning knob
  int tpw = env_tpw > 0 ? env_tpw : 16;    // long walks: every workgroup ends with 216*NC same-address global atomics
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * B < 2L * 256) tpw >>= 1;            // ... but keep >= 2 per CU
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid((ntiles + tpw - 1) / tpw, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == C3D_DT_F32)
    head_bwd_kernel<float><<<grid, HD_TH * HD_TW, 0, s>>>(dout, prob, (const float*)x, w, (float*)dx, dw, B, H, W,
                                                           NC, has_sigmoid, tpw);
  else if (dtype == C3D_DT_BF16)
    head_bwd_kernel<bf16_t><<<grid, HD_TH * HD_TW, 0, s>>>(dout, prob, (const bf16_t*)x, w, (bf16_t*)dx, dw, B, H,
                                                            W, NC, has_sigmoid, tpw);
  else return C3D_E_BADARG;
  C3D_CHECK_LAUNCH();
  return 0;
}
