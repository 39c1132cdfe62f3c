// ConvTranspose2d k=4 s=2 p=1 of the ChangeDecoder (reference model/change_decoder.py:30-45: `up_c4/up_c3/up_c2[1]`,
// C -> C channels, C = 48 / 24 / 24) on MFMA for the bf16 (throughput) path: forward, data gradient and weight
// gradient.  The round-1 kernels were scalar-FMA stencils (decoder.hip; kept for the f32 parity path): at the top
// decoder level (24 channels, 128x128 -> 256x256, B=32) they ran at 0.24 / 0.28 ms (forward / data gradient) and the
// weight gradient -- 16 shifted passes of the generic pointwise weight-gradient kernel -- at 0.75 ms, against ~50 us of
// HBM time each.  A transposed convolution is 2304 MAC per output pixel at C=24: matrix-core work.
//
//   forward   out[b, 2qy+py, 2qx+px, co] = bias[co] + skip[...] + sum_{jy,jx,ci} in[b, qy+dy, qx+dx, ci] W[ci][co][ky][kx]
//             -> per output parity class (py,px) a GEMM  [16 pixels] x [K = 4 taps x C] x [C]:  the data operand of a
//             lane is 8 consecutive channels of ONE neighbour pixel = one 16-byte global load (no LDS staging: the 3x3
//             neighbourhood re-reads hit L1/L2), the weight operand is a ready-made fragment image in LDS.
//   bwd data  din[b, iy, ix, ci] = sum_{ky,kx,co} dout[b, 2iy-1+ky, 2ix-1+kx, co] W[ci][co][ky][kx]
//             -> one GEMM [16 pixels] x [K = 16 taps x C] x [C], same operand scheme.
//   wgrad     dW[ci][co][ky][kx] = sum_{b,i,j} t[b,i,j,ci] dcur[b, 2i-1+ky, 2j-1+kx, co]: the contraction runs over
//             PIXELS, so both operands are read "transposed" (one channel, 8 consecutive pixels) from an LDS copy of a
//             4x32-pixel tile of t and its (10 x 66)-pixel patch of dcur, staged ONCE for all 16 taps (the generic
//             kernel re-read dcur 16 times); wave w owns the 4 taps with ky = w; per-workgroup partial sums go to a
//             workspace and a small reducer adds them into dW (fixed order: deterministic).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include <cstdlib>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4_t mfma_bf16(const u32x4 a, const u32x4 b, const f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// output parity (0|1), tap index j (0|1) -> kernel index k and input offset d (out = 2*in - 1 + k)
__device__ __forceinline__ void par_tap(const int par, const int j, int& k, int& d) {
  if (par == 0) { k = j ? 3 : 1; d = j ? -1 : 0; }
  else { k = j ? 2 : 0; d = j ? 0 : 1; }
}

__device__ __forceinline__ u32x4 ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }

// ------------------------------------------------------------------------------------------------ forward
template <int C>
__global__ __launch_bounds__(256) void convt_fwd_mfma_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const bf16_t* __restrict__ skip,
                                                             const int64_t skip_bstride, bf16_t* __restrict__ out, const int B,
                                                             const int h, const int wd) {
  constexpr int NT = (C + 15) / 16, KS = 4 * C / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Wl = reinterpret_cast<u32x4*>(smem);   // [4 parities][NT][KS][64 lanes] weight fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the f32 weights go to LDS in one coalesced pass (into the region the output staging uses later): the fragment image below
  // was gathered element by element from global memory -- 37 k dependent-latency 4-byte loads per workgroup at C = 48, 45 us
  // for a launch that moves 15 MB
  constexpr int WRS = C * 16 + 2;                                     // ci rows padded by one dword (LDS banks)
  bf16_t* wraw = reinterpret_cast<bf16_t*>(Wl + 4 * NT * KS * 64);   // (bf16: 72 KB at C = 48, beside the 72 KB image)
  for (int i = tid; i < C * C * 16; i += 256) wraw[(i / (C * 16)) * WRS + i % (C * 16)] = f32_to_bf16(w[i]);
  __syncthreads();
  // ---- weight fragment image: element (f, l, e): co = nt*16 + (l&15), k = ks*32 + (l>>4)*8 + e = tap*C + ci
  for (int i = tid; i < 4 * NT * KS * 64; i += 256) {
    const int l = i & 63, f = i >> 6;
    const int ks = f % KS, nt = (f / KS) % NT, par = f / (KS * NT);
    const int co = nt * 16 + (l & 15);
    uint32_t pk[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      uint32_t v[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = ks * 32 + (l >> 4) * 8 + e2 * 2 + hh;
        const int tap = k / C, ci = k - tap * C;
        int ky, kx, dd;
        par_tap(par >> 1, tap >> 1, ky, dd);
        par_tap(par & 1, tap & 1, kx, dd);
        v[hh] = co < C ? wraw[ci * WRS + (co * 4 + ky) * 4 + kx] : 0u;
      }
      pk[e2] = v[0] | (v[1] << 16);
    }
    Wl[i] = u32x4{pk[0], pk[1], pk[2], pk[3]};
  }
  __syncthreads();
  const int p = lane & 15, g = lane >> 4;
  const int tiles_x = (wd + 15) >> 4;
  const int64_t ntile = (int64_t)B * h * tiles_x;
  const int H = 2 * h, W = 2 * wd;
  int tapv[KS], civ[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { const int k0 = ks * 32 + g * 8; tapv[ks] = k0 / C; civ[ks] = k0 - tapv[ks] * C; }
  float bv[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int co = nt * 16 + g * 4 + r; bv[nt][r] = co < C ? bias[co] : 0.f; }
  // Output staging (round 4): the four parity classes of a wave's 16 input pixels are the two output rows 2qy, 2qy+1 x 32
  // pixels = 2 x 1536 contiguous bytes.  Stored class by class they were 8-byte pieces at a 96-byte stride (and the skip rows
  // were read the same way, after the MFMAs); now the f32 results go through a per-wave LDS tile [2 rows][32 px][C], the skip
  // rows are requested as 16-byte vectors BEFORE the MFMAs and the tile leaves as 16-byte vectors of whole rows.
  constexpr int CH16 = C / 8;                         // 16-byte chunks per output pixel
  constexpr int NCH = 2 * 32 * CH16;                  // chunks of a wave tile
  constexpr int NJ = (NCH + 63) / 64;
  float* Ot = reinterpret_cast<float*>(Wl + 4 * NT * KS * 64) + (size_t)wave * 2 * 32 * C;
  for (int64_t t = ((int64_t)blockIdx.x * 4 + wave); t < ntile; t += (int64_t)gridDim.x * 4) {
    const int tx = (int)(t % tiles_x);
    const int64_t r_ = t / tiles_x;
    const int qy = (int)(r_ % h), b = (int)(r_ / h);
    const int qx = tx * 16 + p;
    const bf16_t* inb = in + (size_t)b * h * wd * C;
    u32x4 sk[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      const int row = c / (32 * CH16), cc = c - row * (32 * CH16);
      const int opx = cc / CH16;
      sk[j] = u32x4{0u, 0u, 0u, 0u};
      if (skip && c < NCH && tx * 32 + opx < W)
        sk[j] = ld16(skip + (size_t)b * skip_bstride + ((size_t)(2 * qy + row) * W + tx * 32) * C + (size_t)cc * 8);
    }
#pragma unroll
    for (int par = 0; par < 4; ++par) {
      const int py = par >> 1, px = par & 1;
      u32x4 xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        int ky, kx, dy, dx;
        par_tap(py, tapv[ks] >> 1, ky, dy);
        par_tap(px, tapv[ks] & 1, kx, dx);
        const int yy = qy + dy, xx = qx + dx;
        const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < wd;
        xf[ks] = ok ? ld16(inb + ((size_t)yy * wd + xx) * C + civ[ks]) : u32x4{0u, 0u, 0u, 0u};
      }
      f32x4_t acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{bv[nt][0], bv[nt][1], bv[nt][2], bv[nt][3]};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_bf16(Wl[((par * NT + nt) * KS + ks) * 64 + lane], xf[ks], acc[nt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + g * 4;
        if (co < C) *reinterpret_cast<float4*>(Ot + (size_t)(py * 32 + 2 * p + px) * C + co) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      }
    }
    __builtin_amdgcn_wave_barrier();                  // (LDS operations of one wave complete in order)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      const int row = c / (32 * CH16), cc = c - row * (32 * CH16);
      const int opx = cc / CH16;
      if (c < NCH && tx * 32 + opx < W) {
        const float4 lo = *reinterpret_cast<const float4*>(Ot + (size_t)row * 32 * C + (size_t)cc * 8);
        const float4 hi = *reinterpret_cast<const float4*>(Ot + (size_t)row * 32 * C + (size_t)cc * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (skip) {
          const uint32_t sw[4] = {sk[j][0], sk[j][1], sk[j][2], sk[j][3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(sw[e] << 16); v[2 * e + 1] += __uint_as_float(sw[e] & 0xffff0000u); }
        }
        *reinterpret_cast<u32x4*>(out + (size_t)b * H * W * C + ((size_t)(2 * qy + row) * W + tx * 32) * C + (size_t)cc * 8) =
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
template <int C>
__global__ __launch_bounds__(256) void convt_bwd_data_mfma_kernel(const bf16_t* __restrict__ dout, const float* __restrict__ w,
                                                                  bf16_t* __restrict__ din, const int B, const int h,
                                                                  const int wd) {
  constexpr int NT = (C + 15) / 16, KS = 16 * C / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Wl = reinterpret_cast<u32x4*>(smem);   // [NT][KS][64]: A[i = ci][k = tap16*C + co] = W[ci][co][ky][kx]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (see the forward kernel; aliases the patch staging.  Rows of ci padded by one dword: the 16 lanes of a fragment read 16
  // different ci at the same (co, tap) -- 768 bytes apart, one bank)
  constexpr int WRS = C * 16 + 2;
  bf16_t* wraw = reinterpret_cast<bf16_t*>(Wl + NT * KS * 64);
  for (int i = tid; i < C * C * 16; i += 256) wraw[(i / (C * 16)) * WRS + i % (C * 16)] = f32_to_bf16(w[i]);
  __syncthreads();
  for (int i = tid; i < NT * KS * 64; i += 256) {
    const int l = i & 63, f = i >> 6;
    const int ks = f % KS, nt = f / KS;
    const int ci = nt * 16 + (l & 15);
    uint32_t pk[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      uint32_t v[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = ks * 32 + (l >> 4) * 8 + e2 * 2 + hh;
        const int tap = k / C, co = k - tap * C;
        v[hh] = ci < C ? wraw[ci * WRS + co * 16 + tap] : 0u;
      }
      pk[e2] = v[0] | (v[1] << 16);
    }
    Wl[i] = u32x4{pk[0], pk[1], pk[2], pk[3]};
  }
  __syncthreads();
  const int p = lane & 15, g = lane >> 4;
  const int tiles_x = (wd + 15) >> 4;
  const int64_t ntile = (int64_t)B * h * tiles_x;
  const int H = 2 * h, W = 2 * wd;
  // dout patch of a wave's 16 input pixels: output rows 2iy-1 .. 2iy+2 x columns 2x0-1 .. 2x0+32 = 4 rows of 34 pixels,
  // each row contiguous in memory.  Round 4: the patch is staged in LDS by 16-byte row vectors (next tile's vectors are in
  // registers while this tile multiplies) and the 16 x C/8 operand vectors of a lane are ds_read_b128 -- the gathers used to be
  // 16-byte global loads at a 2-pixel stride, every dout pixel fetched by four lanes through L1.
  constexpr int CH16 = C / 8, PW_ = 34, PCH = 4 * PW_ * CH16, NJ = (PCH + 63) / 64;
  bf16_t* Pl = reinterpret_cast<bf16_t*>(Wl + NT * KS * 64) + (size_t)wave * (4 * PW_ * C + 8);
  auto issue = [&](const int64_t t, u32x4 (&r)[NJ]) {
    const int tx = (int)(t % tiles_x);
    const int64_t r_ = t / tiles_x;
    const int iy = (int)(r_ % h), b = (int)(r_ / h);
    const bf16_t* db = dout + (size_t)b * H * W * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      const int row = c / (PW_ * CH16), cc = c - row * (PW_ * CH16);
      const int oy = 2 * iy - 1 + row, ox = 2 * tx * 16 - 1 + cc / CH16;
      r[j] = u32x4{0u, 0u, 0u, 0u};
      if (c < PCH && oy >= 0 && oy < H && ox >= 0 && ox < W) r[j] = ld16(db + ((size_t)oy * W + ox) * C + (cc % CH16) * 8);
    }
  };
  u32x4 nx[NJ];
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t < ntile) issue(t, nx);
  for (; t < ntile; t += (int64_t)gridDim.x * 4) {
    const int tx = (int)(t % tiles_x);
    const int64_t r_ = t / tiles_x;
    const int iy = (int)(r_ % h), b = (int)(r_ / h);
    const int ix = tx * 16 + p;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < PCH) *reinterpret_cast<u32x4*>(Pl + (size_t)c * 8) = nx[j];
    }
    if (t + (int64_t)gridDim.x * 4 < ntile) issue(t + (int64_t)gridDim.x * 4, nx);
    __builtin_amdgcn_wave_barrier();
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const int k0 = ks * 32 + g * 8;
      const int tap = k0 / C, co0 = k0 - tap * C;
      const u32x4 xf = *reinterpret_cast<const u32x4*>(Pl + (size_t)((tap >> 2) * PW_ + 2 * p + (tap & 3)) * C + co0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_bf16(Wl[(nt * KS + ks) * 64 + lane], xf, acc[nt]);
    }
    __builtin_amdgcn_wave_barrier();
    if (ix < wd) {
      bf16_t* o = din + (((size_t)b * h + iy) * wd + ix) * C;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ci = nt * 16 + g * 4;
        if (ci < C)
          *reinterpret_cast<uint2*>(o + ci) = make_uint2(pack_bf16x2(acc[nt][0], acc[nt][1]), pack_bf16x2(acc[nt][2], acc[nt][3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int WG_TH = 4, WG_TW = 32;                       // tile of input-resolution pixels (one k-step = one row)
constexpr int WG_PH = 2 * WG_TH + 2, WG_PW = 2 * WG_TW + 2;  // its dcur patch

template <int C>
__global__ __launch_bounds__(256) void convt_wgrad_mfma_kernel(const bf16_t* __restrict__ tin, const bf16_t* __restrict__ dcur,
                                                               float* __restrict__ ws, const int B, const int h, const int wd) {
  constexpr int NT = (C + 15) / 16;
  constexpr int CP = C + 2;            // LDS pixel stride in bf16 (odd dword count: spreads the 8-pixel column reads over banks)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Tl = reinterpret_cast<bf16_t*>(smem);                       // [WG_TH][WG_TW][CP]
  bf16_t* Dl = Tl + WG_TH * WG_TW * CP;                               // [WG_PH][WG_PW][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // wave = ky
  const int i15 = lane & 15, g = lane >> 4;
  const int tiles_x = (wd + WG_TW - 1) / WG_TW, tiles_y = (h + WG_TH - 1) / WG_TH;
  const int64_t ntile = (int64_t)B * tiles_y * tiles_x;
  const int H = 2 * h, W = 2 * wd;
  f32x4_t acc[4][NT][NT];   // [kx][ci tile][co tile]
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[kx][a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int VPP = C / 8;   // 16-byte vectors per pixel
  for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int tx = (int)(t % tiles_x);
    const int64_t r_ = t / tiles_x;
    const int ty = (int)(r_ % tiles_y), b = (int)(r_ / tiles_y);
    const int iy0 = ty * WG_TH, ix0 = tx * WG_TW;
    __syncthreads();   // previous tile's reads are done
    // ---- stage t tile and dcur patch (zero outside the image); 8 bf16 per 16-byte load, written as 4 dwords (CP is even)
    for (int i = tid; i < WG_TH * WG_TW * VPP; i += 256) {
      const int v = i % VPP, px = (i / VPP) % WG_TW, py = i / (VPP * WG_TW);
      const int iy = iy0 + py, ix = ix0 + px;
      u32x4 x = u32x4{0u, 0u, 0u, 0u};
      if (iy < h && ix < wd) x = ld16(tin + (((size_t)b * h + iy) * wd + ix) * C + v * 8);
      uint32_t* d = reinterpret_cast<uint32_t*>(Tl + (py * WG_TW + px) * CP + v * 8);
      d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; d[3] = x[3];
    }
    for (int i = tid; i < WG_PH * WG_PW * VPP; i += 256) {
      const int v = i % VPP, px = (i / VPP) % WG_PW, py = i / (VPP * WG_PW);
      const int oy = 2 * iy0 - 1 + py, ox = 2 * ix0 - 1 + px;
      u32x4 x = u32x4{0u, 0u, 0u, 0u};
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) x = ld16(dcur + (((size_t)b * H + oy) * W + ox) * C + v * 8);
      uint32_t* d = reinterpret_cast<uint32_t*>(Dl + (py * WG_PW + px) * CP + v * 8);
      d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; d[3] = x[3];
    }
    __syncthreads();
    // ---- one k-step per tile row: K = 32 pixels (j = g*8 + e)
#pragma unroll 1
    for (int r = 0; r < WG_TH; ++r) {
      u32x4 af[NT];   // A[i = ci][k = pixel] = t[r][j][ci]
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int ci = a * 16 + i15;
        uint32_t pk[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const bf16_t* s = Tl + (r * WG_TW + g * 8 + e2 * 2) * CP + ci;
          const uint32_t lo = ci < C ? s[0] : 0, hi = ci < C ? s[CP] : 0;
          pk[e2] = lo | (hi << 16);
        }
        af[a] = u32x4{pk[0], pk[1], pk[2], pk[3]};
      }
      const int prow = 2 * r + wave;        // patch row of tap ky = wave: oy - (2*iy0 - 1) = 2r + ky
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          const int co = c * 16 + i15;
          uint32_t pk[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const bf16_t* s = Dl + (prow * WG_PW + 2 * (g * 8 + e2 * 2) + kx) * CP + co;   // B[k = pixel][j = co]
            const uint32_t lo = co < C ? s[0] : 0, hi = co < C ? s[2 * CP] : 0;
            pk[e2] = lo | (hi << 16);
          }
          const u32x4 bfrag = u32x4{pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
          for (int a = 0; a < NT; ++a) acc[kx][a][c] = mfma_bf16(af[a], bfrag, acc[kx][a][c]);
        }
      }
    }
  }
  // ---- per-workgroup partial: ws[wg][tap = ky*4+kx][ci][co] (C x C each); D[i = ci = g*4 + r][j = co = i15]
  float* dst = ws + (size_t)blockIdx.x * 16 * C * C;
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int co = c * 16 + i15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = a * 16 + g * 4 + r;
          if (ci < C && co < C) dst[((size_t)(wave * 4 + kx) * C + ci) * C + co] = acc[kx][a][c][r];
        }
      }
}

// dw[ci][co][ky][kx] += sum over workgroups of ws[wg][tap][ci][co].  32 outputs per block x 8 part-groups: a thread adds
// at most nwg/8 coalesced values with four loads in flight, the groups are combined through LDS in fixed order (one
// thread per output walking all nwg partials took 100 us per launch: a serial chain of ~1000 dependent-latency loads).
__global__ __launch_bounds__(256) void convt_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, const int nwg,
                                                                 const int C) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  const int n = 16 * C * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int g = pg;
    for (; g + 24 < nwg; g += 32) {
      s0 += ws[(size_t)g * n + i];
      s1 += ws[(size_t)(g + 8) * n + i];
      s2 += ws[(size_t)(g + 16) * n + i];
      s3 += ws[(size_t)(g + 24) * n + i];
    }
    for (; g < nwg; g += 8) s0 += ws[(size_t)g * n + i];
  }
  red[pg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pg == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][e];
    const int co = i % C, ci = (i / C) % C, tap = i / (C * C);
    dw[((size_t)ci * C + co) * 16 + tap] += s;
  }
}

int persistent_grid(const int64_t work_items, const int per_cu) {
  int64_t gmax = (int64_t)device_cus() * per_cu;
  if (gmax > work_items) gmax = work_items;
  return (int)(gmax < 1 ? 1 : gmax);
}

template <int C>
int launch_fwd(const void* in, const float* w, const float* bias, const void* skip, int64_t skip_bstride, void* out, int B,
               int h, int wd, hipStream_t st) {
  constexpr int NT = (C + 15) / 16, KS = 4 * C / 32;
  size_t stage = (size_t)4 * 2 * 32 * C * sizeof(float);
  if (stage < (size_t)C * (C * 16 + 2) * sizeof(bf16_t)) stage = (size_t)C * (C * 16 + 2) * sizeof(bf16_t);   // the raw weights, first
  const size_t lds = (size_t)4 * NT * KS * 64 * 16 + stage;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_fwd_mfma_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return C3D_E_UNSUPPORTED;
    attr = true;
  }
  const int64_t wave_tiles = (int64_t)B * h * ((wd + 15) / 16);
  const int grid = persistent_grid((wave_tiles + 3) / 4, C <= 24 ? 6 : 2);
  convt_fwd_mfma_kernel<C><<<grid, 256, lds, st>>>((const bf16_t*)in, w, bias, (const bf16_t*)skip, skip_bstride, (bf16_t*)out, B, h, wd);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int C>
int launch_bwd_data(const void* dout, const float* w, void* din, int B, int h, int wd, hipStream_t st) {
  constexpr int NT = (C + 15) / 16, KS = 16 * C / 32;
  size_t stage = (size_t)4 * (4 * 34 * C + 8) * sizeof(bf16_t);
  if (stage < (size_t)C * (C * 16 + 2) * sizeof(bf16_t)) stage = (size_t)C * (C * 16 + 2) * sizeof(bf16_t);
  const size_t lds = (size_t)NT * KS * 64 * 16 + stage;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_bwd_data_mfma_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return C3D_E_UNSUPPORTED;
    attr = true;
  }
  const int64_t wave_tiles = (int64_t)B * h * ((wd + 15) / 16);
  const int grid = persistent_grid((wave_tiles + 3) / 4, C <= 24 ? 6 : 2);
  convt_bwd_data_mfma_kernel<C><<<grid, 256, lds, st>>>((const bf16_t*)dout, w, (bf16_t*)din, B, h, wd);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <int C>
int wgrad_grid(int B, int h, int wd) {
  const int64_t tiles = (int64_t)B * ((h + WG_TH - 1) / WG_TH) * ((wd + WG_TW - 1) / WG_TW);
  return persistent_grid(tiles, 2);
}

template <int C>
int launch_wgrad(const void* t, const void* dcur, float* dw, float* ws, int B, int h, int wd, hipStream_t st) {
  constexpr int CP = C + 2;
  const size_t lds = (size_t)(WG_TH * WG_TW + WG_PH * WG_PW) * CP * 2;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_wgrad_mfma_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return C3D_E_UNSUPPORTED;
    attr = true;
  }
  const int grid = wgrad_grid<C>(B, h, wd);
  convt_wgrad_mfma_kernel<C><<<grid, 256, lds, st>>>((const bf16_t*)t, (const bf16_t*)dcur, ws, B, h, wd);
  C3D_CHECK_LAUNCH();
  const int n = 16 * C * C;
  convt_wgrad_reduce_kernel<<<(n + 31) / 32, 256, 0, st>>>(ws, dw, grid, C);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// bf16 dispatch used by c3d_convT4s2_fwd / c3d_convT4s2_bwd_data (decoder.hip); returns C3D_E_UNSUPPORTED for shapes
// the MFMA kernels do not cover (the caller then falls back to the scalar kernels).
int c3d_detail_convt_fwd_bf16(const void* in, const float* w, const float* bias, const void* skip, int64_t skip_bstride, void* out,
                              int B, int h, int wd, int C, hipStream_t st) {
  if ((skip_bstride & 3) != 0) return C3D_E_UNSUPPORTED;
  if (C == 24) return launch_fwd<24>(in, w, bias, skip, skip_bstride, out, B, h, wd, st);
  if (C == 48) return launch_fwd<48>(in, w, bias, skip, skip_bstride, out, B, h, wd, st);
  return C3D_E_UNSUPPORTED;
}

int c3d_detail_convt_bwd_data_bf16(const void* dout, const float* w, void* din, int B, int h, int wd, int C, hipStream_t st) {
  if (C == 24) return launch_bwd_data<24>(dout, w, din, B, h, wd, st);
  if (C == 48) return launch_bwd_data<48>(dout, w, din, B, h, wd, st);
  return C3D_E_UNSUPPORTED;
}

extern "C" int64_t c3d_convT4s2_wgrad_ws_floats(int32_t B, int32_t h, int32_t wd, int32_t C) {
  if (C == 24) return (int64_t)wgrad_grid<24>(B, h, wd) * 16 * C * C;
  if (C == 48) return (int64_t)wgrad_grid<48>(B, h, wd) * 16 * C * C;
  return -1;
}

extern "C" int c3d_convT4s2_wgrad(const void* t, const void* dcur, float* dw, float* ws, int32_t B, int32_t h, int32_t wd,
                                  int32_t C, int32_t dtype, void* stream) {
  if (!t || !dcur || !dw || !ws || B <= 0 || h <= 0 || wd <= 0) return C3D_E_BADARG;
  if (dtype != C3D_DT_BF16) return C3D_E_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C == 24) return launch_wgrad<24>(t, dcur, dw, ws, B, h, wd, st);
  if (C == 48) return launch_wgrad<48>(t, dcur, dw, ws, B, h, wd, st);
  return C3D_E_UNSUPPORTED;
}
