// ChangeDecoder head on the matrix cores (bf16 storage): Conv2d 3x3, 24 -> NC <= 8 classes, no bias, optional sigmoid
// (reference model/change_decoder.py:46-55 `up_c1`, called at :78-81), forward and backward.
//
// The scalar kernels in decoder.hip (kept for f32 storage and as the parity reference of these) run at 0.5-0.8 TB/s: one
// thread per pixel, 216 x NC FMAs behind 63 LDS reads, no overlap of loads and arithmetic.  Here the convolution is an implicit
// GEMM over 16-pixel row segments:
//   forward   out[px][n]  = sum_tap  X[px + tap][0..31] . W[n][0..31][tap]          9 x v_mfma_f32_16x16x32_bf16 per 16 pixels
//             (channels 24..31 are zero columns of the LDS tile: one tap = one k-step, the A fragment of a lane is ONE
//             ds_read_b128 of a neighbour pixel, the nine weight fragments live in registers for the whole kernel)
//   dx        dx[c][px]   = sum_ks   Wd[c][(tap, n)] . DL[px - tap][n]              k = 4 taps x 8 classes per step, 3 steps
//             (roles swapped: a lane ends up with 4 consecutive channels of one pixel -> 8-byte stores)
//   dW        dW[n][c][tap] += sum_px DL[px][n] . X[px + tap][c]                    pixels are the contraction index:
//             both operands come out of the same LDS tiles through ds_read_b64_tr_b16 (rows = pixels along x)
// DL = dout * p (1 - p) (sigmoid head) or dout, rounded to bf16 for the matrix cores (dx is stored in bf16 anyway; dW sums in f32).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "launch_hints.h"

namespace {

constexpr int HM_C = 24, HM_MAXNC = 8;
constexpr int HM_TH = 8, HM_TW = 32;                 // output tile: 8 rows x 32 columns = 16 segments of 16 pixels, 4 per wave
constexpr int HM_IH = HM_TH + 2, HM_IW = HM_TW + 2;
constexpr int HM_XS = 32;                            // bf16 elements per pixel in the X tile (24 channels + 8 zero columns)
constexpr int HM_NTHR = 256;

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;

__device__ __forceinline__ f32x4_t mfma32(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// raw rows of the X tile of output tile (ty, tx): item i -> (pixel p = i / 3, channel vector v = i % 3), 4 slots per thread
struct XTile {
  uint4 raw[4];
  unsigned mask;
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ x, int b, int H, int W, int y0, int x0, int tid) {
    mask = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int i = tid + s * HM_NTHR;
      raw[s] = make_uint4(0, 0, 0, 0);
      if (i < HM_IH * HM_IW * 3) {
        const int p = i / 3, v = i - p * 3;
        const int ly = p / HM_IW, lx = p - ly * HM_IW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          raw[s] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + gy) * W + gx) * HM_C + v * 8);
          mask |= 1u << s;
        }
      }
    }
  }
  __device__ __forceinline__ void store(bf16_t* xt, int tid) const {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int i = tid + s * HM_NTHR;
      if (i < HM_IH * HM_IW * 3) {
        const int p = i / 3, v = i - p * 3;
        *reinterpret_cast<uint4*>(xt + p * HM_XS + v * 8) = raw[s];
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(HM_NTHR) void head_fwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ out, int B, int H, int W, int NC,
                                                                int has_sigmoid, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) bf16_t xt[HM_IH * HM_IW * HM_XS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, li = lane & 15;
  const int b = blockIdx.y;
  const int tiles_x = (W + HM_TW - 1) / HM_TW, tiles_y = (H + HM_TH - 1) / HM_TH, ntiles = tiles_x * tiles_y;
  // weight fragments (B operand): lane = (class n = li, channels 8 g4 + j) of tap t
  uint4 wf[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * g4 + j;
      f[j] = (li < NC && c < HM_C) ? w[((size_t)li * HM_C + c) * 9 + t] : 0.f;
    }
    wf[t] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
  for (int i = tid; i < HM_IH * HM_IW; i += HM_NTHR) *reinterpret_cast<uint4*>(xt + i * HM_XS + 24) = make_uint4(0, 0, 0, 0);
  int tl0 = blockIdx.x * tiles_per_wg, tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  XTile cur;
  if (tl0 < tl1) cur.issue(x, b, H, W, (tl0 / tiles_x) * HM_TH, (tl0 % tiles_x) * HM_TW, tid);
  for (int tl = tl0; tl < tl1; ++tl) {
    const int y0 = (tl / tiles_x) * HM_TH, x0 = (tl % tiles_x) * HM_TW;
    __syncthreads();                         // the previous tile's fragment reads are done
    cur.store(xt, tid);
    if (tl + 1 < tl1) cur.issue(x, b, H, W, ((tl + 1) / tiles_x) * HM_TH, ((tl + 1) % tiles_x) * HM_TW, tid);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int r = 2 * wave + (mt >> 1), xs = (mt & 1) * 16;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        const uint4 a = *reinterpret_cast<const uint4*>(xt + ((r + ky) * HM_IW + xs + li + kx) * HM_XS + 8 * g4);
        acc = mfma32(a, wf[t], acc);
      }
      // D: lane = (pixels xs + 4 g4 + r', class li)
      const int gy = y0 + r, gx = x0 + xs + 4 * g4;
      if (li < NC && gy < H) {
        float* dst = out + (((size_t)b * NC + li) * H + gy) * W + gx;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[q];
          if (has_sigmoid) v = 1.0f / (1.0f + expf(-v));
          if (gx + q < W) dst[q] = v;
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(HM_NTHR) __attribute__((amdgpu_waves_per_eu(2, 3))) void head_bwd_mfma_kernel(const float* __restrict__ dout, const float* __restrict__ prob,
                                                                const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                bf16_t* __restrict__ dx, float* __restrict__ dw,
                                                                float* __restrict__ ws, int B, int H,
                                                                int W, int NC, int has_sigmoid, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) bf16_t xt[HM_IH * HM_IW * HM_XS];
  __shared__ __attribute__((aligned(16))) bf16_t dl[(HM_IH * HM_IW + 1) * 8];   // [pixel][8 classes]; the last 16 bytes stay zero
  __shared__ float red[9 * 2 * 8 * 16];                                          // dW partial of the workgroup [tap][ct][class][ch]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, li = lane & 15;
  const int b = blockIdx.y;
  const int tiles_x = (W + HM_TW - 1) / HM_TW, tiles_y = (H + HM_TH - 1) / HM_TH, ntiles = tiles_x * tiles_y;
  // data-gradient weight fragments (A operand): lane = (channel c = 16 mt + li, k = (tap 4 ks + g4, class j))
  uint4 wd[3][2];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int c = 16 * mt + li, t = 4 * ks + g4;
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (c < HM_C && t < 9 && j < NC) ? w[((size_t)j * HM_C + c) * 9 + t] : 0.f;
      wd[ks][mt] = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
  for (int i = tid; i < HM_IH * HM_IW; i += HM_NTHR) *reinterpret_cast<uint4*>(xt + i * HM_XS + 24) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < HM_IH * HM_IW + 1; i += HM_NTHR) *reinterpret_cast<uint4*>(dl + i * 8) = make_uint4(0, 0, 0, 0);
  f32x4_t dwa[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { dwa[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dwa[t][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  int tl0 = blockIdx.x * tiles_per_wg, tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  XTile cur;
  if (tl0 < tl1) cur.issue(x, b, H, W, (tl0 / tiles_x) * HM_TH, (tl0 % tiles_x) * HM_TW, tid);
  for (int tl = tl0; tl < tl1; ++tl) {
    const int y0 = (tl / tiles_x) * HM_TH, x0 = (tl % tiles_x) * HM_TW;
    __syncthreads();
    cur.store(xt, tid);
    // d logit tile [pixel][class] (zero outside the image; classes >= NC were zeroed once)
    for (int i = tid; i < NC * HM_IH * HM_IW; i += HM_NTHR) {
      const int lx = i % HM_IW;
      const int q = i / HM_IW;
      const int ly = q % HM_IH, n = q / HM_IH;
      const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const size_t o = (((size_t)b * NC + n) * H + gy) * W + gx;
        v = dout[o];
        if (has_sigmoid) { const float p = prob[o]; v *= p * (1.f - p); }
      }
      dl[(ly * HM_IW + lx) * 8 + n] = f32_to_bf16(v);
    }
    if (tl + 1 < tl1) cur.issue(x, b, H, W, ((tl + 1) / tiles_x) * HM_TH, ((tl + 1) % tiles_x) * HM_TW, tid);
    __syncthreads();
#pragma unroll 1
    for (int mt = 0; mt < 4; ++mt) {   // (a real loop: unrolled, the kernel took 284 registers = one workgroup per CU)
      const int r = 2 * wave + (mt >> 1), xs = (mt & 1) * 16;
      // ---- data gradient: D[c][px] = sum_ks Wd[c][(tap, n)] DL[px - tap][n]
      f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int t = 4 * ks + g4;
        const int ky = t / 3, kx = t - ky * 3;
        // output pixel (y - ky + 1, x - kx + 1): tile-local (r + 2 - ky, xs + li + 2 - kx); taps >= 9 read the zero slot
        const int off = t < 9 ? ((r + 2 - ky) * HM_IW + xs + li + 2 - kx) : HM_IH * HM_IW;
        const uint4 bq = *reinterpret_cast<const uint4*>(dl + off * 8);
        a0 = mfma32(wd[ks][0], bq, a0);
        a1 = mfma32(wd[ks][1], bq, a1);
      }
      const int gy = y0 + r, gx = x0 + xs + li;
      if (gy < H && gx < W) {
        bf16_t* dst = dx + (((size_t)b * H + gy) * W + gx) * HM_C;
        *reinterpret_cast<uint2*>(dst + 4 * g4) = make_uint2(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]));
        if (g4 < 2) *reinterpret_cast<uint2*>(dst + 16 + 4 * g4) = make_uint2(pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3]));
      }
      // ---- weight gradient: dW[n][c][tap] += sum_px DL[px][n] X[px + tap][c]  (16 pixels of this segment as k)
      // A = DL^T: lane (class li, pixels 4 g4 + j); lanes addressing the class chunks 8..15 read the zero slot
      const int arow = (r + 1) * HM_IW + xs + 1 + 4 * g4 + (li >> 2);
      const s16x4_t af = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4_ptr_t)(dl + ((li & 3) < 2 ? arow * 8 + 4 * (li & 3) : HM_IH * HM_IW * 8)));
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        const bf16_t* xb = xt + ((r + ky) * HM_IW + xs + 4 * g4 + (li >> 2) + kx) * HM_XS + 4 * (li & 3);
        const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)xb);
        const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(xb + 16));
        dwa[t][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, b0, dwa[t][0], 0, 0, 0);
        dwa[t][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, b1, dwa[t][1], 0, 0, 0);
      }
    }
  }
  // dW: lane holds classes 4 g4 + q (g4 < 2), channel 16 ct + li; the four waves add their partials one after the other
  // (fixed order, plain LDS read-modify-writes), then one global atomic per value
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv && g4 < 2) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* d = &red[((t * 2 + ct) * 8 + 4 * g4 + q) * 16 + li];
            *d = wv == 0 ? dwa[t][ct][q] : *d + dwa[t][ct][q];
          }
    }
  }
  __syncthreads();
  // one partial per workgroup into ws [workgroups][NC * 216] (summed in fixed order by head_dw_reduce_kernel), or -- without a
  // workspace -- one global atomic per value: ~1000 workgroups then queue on the same 216 NC addresses, which was 3/4 of this
  // kernel's time (263 -> 70 us on the BCD head)
  float* part = ws ? ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NC * HM_C * 9 : nullptr;
  for (int i = tid; i < 9 * 2 * 8 * 16; i += HM_NTHR) {
    const int ch = i & 15, n = (i >> 4) & 7, ct = (i >> 7) & 1, t = i >> 8;
    const int c = 16 * ct + ch;
    if (n < NC && c < HM_C) {
      if (part) part[((size_t)n * HM_C + c) * 9 + t] = red[i];
      else atomicAdd(dw + ((size_t)n * HM_C + c) * 9 + t, red[i]);
    }
  }
}

// dw[i] += sum over the workgroups' partials, 32 lanes per value, four loads in flight per lane (fixed order).  (8 lanes with one
// load -> add chain each: ~128 dependent memory round trips, 40 us for 432 values.)
__global__ __launch_bounds__(256) void head_dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nvals, int parts) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int i = gid >> 5, q = gid & 31;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < nvals) {
    int p = q;
    for (; p + 96 < parts; p += 128) {
      s0 += ws[(size_t)p * nvals + i]; s1 += ws[(size_t)(p + 32) * nvals + i];
      s2 += ws[(size_t)(p + 64) * nvals + i]; s3 += ws[(size_t)(p + 96) * nvals + i];
    }
    for (; p < parts; p += 32) s0 += ws[(size_t)p * nvals + i];
  }
  float s = (s0 + s1) + (s2 + s3);
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
  if (i < nvals && q == 0) dw[i] += s;
}

int head_tiles_per_wg(int ntiles, int B, int dflt) {
  int tpw = dflt;
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * B < 3L * 256) tpw >>= 1;   // >= ~3 workgroups per CU
  return tpw > ntiles ? ntiles : tpw;
}

}  // namespace

int c3d_detail_head_fwd_bf16(const void* x, const float* w, float* out, int B, int H, int W, int NC, int has_sigmoid, hipStream_t s) {
  const int ntiles = ((W + HM_TW - 1) / HM_TW) * ((H + HM_TH - 1) / HM_TH);
  const int tpw = head_tiles_per_wg(ntiles, B, 4);
  head_fwd_mfma_kernel<<<dim3((ntiles + tpw - 1) / tpw, B), HM_NTHR, 0, s>>>((const bf16_t*)x, w, out, B, H, W, NC, has_sigmoid, tpw);
  C3D_CHECK_LAUNCH();
  return 0;
}

int64_t c3d_detail_head_ws_floats(int B, int H, int W, int NC) {
  const int ntiles = ((W + HM_TW - 1) / HM_TW) * ((H + HM_TH - 1) / HM_TH);
  const int tpw = head_tiles_per_wg(ntiles, B, 8);
  return (int64_t)((ntiles + tpw - 1) / tpw) * B * NC * HM_C * 9;
}

int c3d_detail_head_bwd_bf16(const float* dout, const float* prob, const void* x, const float* w, void* dx, float* dw, float* ws,
                             int B, int H, int W, int NC, int has_sigmoid, hipStream_t s) {
  const int ntiles = ((W + HM_TW - 1) / HM_TW) * ((H + HM_TH - 1) / HM_TH);
  const int tpw = head_tiles_per_wg(ntiles, B, 8);
  const dim3 grid((ntiles + tpw - 1) / tpw, B);
  head_bwd_mfma_kernel<<<grid, HM_NTHR, 0, s>>>(dout, prob, (const bf16_t*)x, w, (bf16_t*)dx, dw, ws, B, H, W, NC, has_sigmoid, tpw);
  C3D_CHECK_LAUNCH();
  if (ws) {
    const int nvals = NC * HM_C * 9;
    head_dw_reduce_kernel<<<dim3((nvals * 32 + 255) / 256), 256, 0, s>>>(ws, dw, nvals, (int)(grid.x * grid.y));
    C3D_CHECK_LAUNCH();
  }
  return 0;
}
