// conv_a DATA gradient of the X3D bottleneck WITH its weight gradient on the workgroup-cooperative tile loop (round 6): bf16
// storage, dense rows,
//
//     P[m, k]  = A[k] * t2[m, k] + B[k] + C[k] * a[m, k]                      (BatchNorm_a backward on load)
//     dx[m, n] = (sum_k P[m, k] * W_a[k, n] + res[m, n]) * (y_prev[m, n] > 0)  (+ the previous block's BatchNorm_c-backward sums)
//     dW_a[k, n] += sum_m P[m, k] * y_prev[m, n]
//
// (reference model/x3d.py:173-183: conv_a -> norm_a -> ReLU; one convolution_backward produces both gradients).  Same
// arguments as the C3D_PRO_AFFINE2 / C3D_EPI_ADD / C3D_WG_ROWS form of c3d_pw_gemm, which dispatches here (C3D_OPT_PW_CDG) and
// keeps everything this kernel does not take.
//
// Why.  The wave-private-tile kernel (csrc/pw_gemm_impl.h) holds its fused weight gradient as an f64 accumulator image in LDS
// (LDS atomics): K <= 112, so the 32 x 32 stage (K = 216) ran the data gradient (42 us) and a separate weight-gradient launch
// on the side stream (29 us) that reads t2 and a AGAIN; its res2 / res3 instantiations sit at 54 / 70 % of their byte floor.
// The forward kernels of this round (csrc/pw_cfwd.hip) showed that the cooperative loop streams at ~5 TB/s; here the same
// loop keeps BOTH products of a tile: the converted rows P (row-major LDS tile) are the B operand of the data gradient (rows
// as output index, packed weight image as A) and, read through gfx950's transposing LDS read, the A operand of the weight
// gradient (rows as contraction index) against the y_prev rows the mask needs anyway.  The weight-gradient accumulators
// (K x N / 8 waves: 48 registers at 216 x 96) live in registers for the whole walk; partial sums per workgroup, reduced in
// fixed order by the reducer launch of the first kernel's fused variant (c3d_detail_pw_wgrad_reduce).
//
// Results: operand conversion, weight fragments, k order of the MFMA chain, bf16 staging of the product and the epilogue
// arithmetic are the first kernel's -- dx is bit-identical to it; the BatchNorm sums group their f32 partial sums differently
// and the weight gradient is summed in f32 per workgroup instead of f64 per tile pair (agreement to f32 rounding).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "launch_hints.h"
#include "bn_fin.h"
#include <cstdlib>
#include <cstring>

thread_local int c3d_cdg_defer_reduce = 0, c3d_cdg_parts = 0;

#ifdef C3D_CD_CLOCK
// Debug build only (tools/r6/cdg_clock.py): s_memtime stamps of wave 0 per workgroup of the conv_c kernel -- [0] entry, [1]
// prologue done (weights, coefficients, first rows in), [2] first tile converted, [3] tile loop done, [4] last weight-gradient
// step + partials stored, [5] sums flushed
__device__ unsigned long long c3d_cd_clk[1024][8];
#define CDCLK(i) { if (threadIdx.x == 0) c3d_cd_clk[blockIdx.x & 1023][i] = __builtin_amdgcn_s_memtime(); }
extern "C" int c3d_debug_cd_clock(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_cd_clk), sizeof(c3d_cd_clk));
}
#else
#define CDCLK(i)
#endif

namespace {

constexpr int CD_THREADS = 512;
constexpr uint32_t CD_OOB = 0x80000000u;
constexpr int CD_MAX_PARTS = 512;     // = PW_WG_MAX_PARTS: the fused-variant workspace holds this many K x N partials

struct CdPlan {
  int MT, WR, WC;            // rows per tile = 16 WR; wave grid WR x WC = 8 (data gradient)
  int QG;                    // weight gradient: waves = PG x QG, wave (pg, qg) holds P tiles pg * NPW .., Q tiles qg * NQW ..
  int tiles_per_wg;
  int KL, QL;                // row strides (elements) of the converted P tile and of the y_prev tile
  int img_rows;              // rows of the packed weight image per 8-element k-chunk
  int w_off, a_off, a_bytes, q_off, q_bytes;   // weight image | two P tiles | two y_prev tiles
  int os_off, os_wave;       // per-wave result staging [16][NLw]
  int par_off, dump_off;     // A | B | C [Kp] each, mean | rstd [Np] each; dump
};

typedef uint32_t cd_u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* cd_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* cd_glb_ptr_t;
typedef short cd_s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) cd_s16x4_t* cd_lds_s16x4_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cd_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ uint4 cd_load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const cd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void cd_cvt(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void cd_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// KS: k-steps of 32 of the data gradient (Kpad / 32); NTW: its output tiles (16 channels) per wave; NPW x NQW: weight-
// gradient tiles (P channels x y_prev channels) per wave; RP / RQ: 16-byte items per thread and tile of the P streams / of
// the y_prev stream (prefetch registers); K2: k-steps of 32 rows of the weight gradient (MT / 32)
template <int KS, int NTW, int NPW, int NQW, int RP, int RQ, int K2>
__global__ __launch_bounds__(CD_THREADS) void pw_cdg_a_kernel(const c3d_pw_args a, const CdPlan L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kp = a.Kp, Np = a.Np, Gq = Kp >> 3, Gn = Np >> 3, MT = L.MT, KL = L.KL, QL = L.QL;
  bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem + L.w_off);
  float* const Pp = reinterpret_cast<float*>(smem + L.par_off);       // A | B | C
  float* const Ep = Pp + 3 * Kp;                                      // mean | rstd of the previous block's BatchNorm_c
  bf16_t* const dump = reinterpret_cast<bf16_t*>(smem + L.dump_off) + tid * 8;

  const int M32 = (int)a.M;
  const int tiles = (M32 + MT - 1) / MT;
  int t0 = (int)blockIdx.x * L.tiles_per_wg;
  if (t0 > tiles) t0 = tiles;
  int t1 = t0 + L.tiles_per_wg;
  if (t1 > tiles) t1 = tiles;
  const uint32_t row_hi = (uint32_t)(t1 * MT < M32 ? t1 * MT : M32);
  const __amdgpu_buffer_rsrc_t rX = cd_rsrc(a.x, row_hi * (uint32_t)Kp * 2u);
  const __amdgpu_buffer_rsrc_t rX2 = cd_rsrc(a.x2, row_hi * (uint32_t)Kp * 2u);
  const __amdgpu_buffer_rsrc_t rQ = cd_rsrc(a.wg_x3, row_hi * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rE1 = cd_rsrc(a.e1, (uint32_t)M32 * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rC1 = cd_rsrc(a.add_sums ? a.add_c : nullptr, (uint32_t)M32 * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rY = cd_rsrc(a.y, (uint32_t)M32 * (uint32_t)Np * 2u);

  // ---- item maps: item i = tid + 512 r of a tile <-> (row = i / G, vector = i % G); its bytes sit at tile base + 16 i
  int p_desc[RP], q_desc[RQ];
  uint32_t p_go[RP], q_go[RQ];
  {
    const float invG = 1.0f / (float)Gq, invN = 1.0f / (float)Gn;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const int i = tid + CD_THREADS * r;
      const int row = __float2int_rz(((float)i + 0.5f) * invG);
      const bool ok = i < MT * Gq;
      p_desc[r] = ok ? (row << 5) | (i - row * Gq) : 0;
      p_go[r] = ok ? (uint32_t)i * 16u : CD_OOB;
    }
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int i = tid + CD_THREADS * r;
      const int row = __float2int_rz(((float)i + 0.5f) * invN);
      const bool ok = i < MT * Gn;
      q_desc[r] = ok ? (row << 5) | (i - row * Gn) : 0;
      q_go[r] = ok ? (uint32_t)i * 16u : CD_OOB;
    }
  }
  const bool p_last = (wave * 64 + CD_THREADS * (RP - 1)) < MT * Gq;   // (wave-uniform: the last round has an item for this wave)
  const bool q_last = (wave * 64 + CD_THREADS * (RQ - 1)) < MT * Gn;
  uint4 rawp[RP], rawp2[RP], rawq[RQ];
  const uint32_t tbp = (uint32_t)(MT * Kp * 2), tbq = (uint32_t)(MT * Np * 2);
  {
    const uint32_t bp = (uint32_t)t0 * tbp, bq = (uint32_t)t0 * tbq;
#pragma unroll
    for (int r = 0; r < RP; ++r) { rawp[r] = cd_load(rX, p_go[r] + bp); rawp2[r] = cd_load(rX2, p_go[r] + bp); }
#pragma unroll
    for (int r = 0; r < RQ; ++r) rawq[r] = cd_load(rQ, q_go[r] + bq);
  }

  // ---- lane maps of the data gradient and of its epilogue (csrc/pw_cfwd.hip)
  const int wr = wave % L.WR, wc = wave / L.WR;
  const int nt0 = wc * NTW;                               // first output tile of this wave
  constexpr int NLW = NTW * 16 + 8;                       // row stride (elements) of the wave's result staging
  bf16_t* const Os = reinterpret_cast<bf16_t*>(smem + L.os_off + wave * L.os_wave);
  constexpr int GOW = NTW * 2, RPO = 64 / GOW, NPASS = (16 + RPO - 1) / RPO;
  const int rr_o = lane / GOW, v_o = lane - rr_o * GOW;
  const int cvec = nt0 * 2 + v_o;                         // 8-channel vector of the output row
  const bool act_o = lane < GOW * RPO && cvec * 8 < Np;
  // companion rows of the epilogue (res, and c of the previous block for its BatchNorm_c-backward sums): one request per pass
  // and lane, a tile ahead
  uint4 e1r[NPASS], c1r[NPASS];
#define CD_EOFF(TILE, P) ((act_o && (P) * RPO + rr_o < 16 && (TILE) * MT + wr * 16 + (P) * RPO + rr_o < M32)                        \
                              ? ((uint32_t)((TILE) * MT + wr * 16 + (P) * RPO + rr_o) * (uint32_t)Np + (uint32_t)cvec * 8u) * 2u : CD_OOB)
#pragma unroll
  for (int p = 0; p < NPASS; ++p) { const uint32_t o = t0 < t1 ? CD_EOFF(t0, p) : CD_OOB; e1r[p] = cd_load(rE1, o); c1r[p] = cd_load(rC1, o); }

  // ---- weight image -> LDS (LDS-DMA, 1 KB per wave instruction; the chunk order rotated by the workgroup index)
  {
    const int wbytes = (KS * 4) * L.img_rows * 16;
    const int nchunk = (wbytes + 1023) >> 10;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w_img);
    const int rot = (int)blockIdx.x % nchunk;
    for (int c = wave; c < nchunk; c += 8) {
      int r = c + rot;
      if (r >= nchunk) r -= nchunk;
      const int off = r * 1024 + lane * 16;
      if (off < wbytes)
        __builtin_amdgcn_global_load_lds((cd_glb_ptr_t)(src + off), (cd_lds_ptr_t)(smem + L.w_off + r * 1024), 16, 0, 0);
    }
  }
  // all four tiles zeroed once: the padding columns are never written again (k padding of P: 0 x weight row; channel padding
  // of y_prev: weight-gradient columns nobody stores)
  for (int i = tid * 16; i < 2 * L.a_bytes; i += CD_THREADS * 16) *reinterpret_cast<uint4*>(smem + L.a_off + i) = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid * 16; i < 2 * L.q_bytes; i += CD_THREADS * 16) *reinterpret_cast<uint4*>(smem + L.q_off + i) = make_uint4(0u, 0u, 0u, 0u);

  // ---- BatchNorm_a-backward coefficients: rebuilt from the producer's completed sums (csrc/bn_fin.h; workgroup 0 accumulates
  // d gamma / d beta and writes the vector for other readers) or read from memory -- the first kernel's prologue
  if (a.fin.sums) {
    for (int c = tid; c < Kp; c += CD_THREADS) {
      float cA, cB, cC;
      c3dfin::bn_bwd_coef_consume(a.fin, a.K, Kp, c, blockIdx.x == 0, cA, cB, cC);
      Pp[c] = cA; Pp[Kp + c] = cB; Pp[2 * Kp + c] = cC;
    }
  } else {
    for (int i = tid; i < 3 * Kp; i += CD_THREADS) Pp[i] = a.pro_p[i];
  }
  for (int i = tid; i < 2 * Np; i += CD_THREADS) Ep[i] = a.add_sums ? a.add_mr[i] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA chunks (and the first tile's rows)
  __syncthreads();

  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  const uint32_t mask_on = a.wg_mask_out ? 0xffffffffu : 0u;
  const uint32_t sums_on = a.add_sums ? 0xffffffffu : 0u;
  const bf16_t* const wfrag = Ws + ((size_t)(lane >> 4) * L.img_rows + nt0 * 16 + (lane & 15)) * 8;   // + (ks * 4 * img_rows + t * 16) * 8
  const int xfrag = (wr * 16 + (lane & 15)) * KL + (lane >> 4) * 8;                                  // + ks * 32

  // ---- weight gradient: wave (pg, qg); transposing read: lane l addresses the 8-byte chunk (row 4 (l / 16) + (l % 16) / 4,
  // channels 4 (l % 4) ..) of a 16 x 16 block and receives rows 4 (l / 16) .. + 3 of channel l % 16
  const int qg = wave % L.QG, pg = wave / L.QG;
  const int g4 = lane >> 4, li = lane & 15;
  const int pl = (4 * g4 + (li >> 2)) * KL + 4 * (li & 3) + pg * NPW * 16;
  const int ql = (4 * g4 + (li >> 2)) * QL + 4 * (li & 3) + qg * NQW * 16;
  f32x4_t dacc[NPW][NQW];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int j = 0; j < NQW; ++j) dacc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- one tile: conversion -> row-major LDS tiles.  Branch-free (an item a lane does not have is converted all the same --
  // its request answered zeros -- and written to the lane's 16 bytes of a dump region): ONE basic block the scheduler can
  // interleave with the matrix-core work of the previous tile.  A slot is requested again for the next tile as soon as it
  // is converted.
#define CD_CONVERT(TILE, CP, CQ)                                                                                    \
  {                                                                                                                 \
    const int rowg0_ = (TILE) * MT;                                                                                 \
    const uint32_t bpn_ = (uint32_t)((TILE) + 1) * tbp, bqn_ = (uint32_t)((TILE) + 1) * tbq;                        \
    _Pragma("unroll") for (int r = 0; r < RP; ++r) {                                                                \
      if (r == RP - 1 && !p_last) continue;                                                                         \
      const int row = p_desc[r] >> 5, v = p_desc[r] & 31;                                                           \
      bf16_t* dst = p_go[r] != CD_OOB ? (CP) + row * KL + v * 8 : dump;                                             \
      float f[8], f2[8], cA[8], cB[8], cC[8];                                                                       \
      cd_cvt(rawp[r], f); cd_cvt(rawp2[r], f2);                                                                     \
      cd_ld8(Pp + v * 8, cA); cd_ld8(Pp + Kp + v * 8, cB); cd_ld8(Pp + 2 * Kp + v * 8, cC);                         \
      /* a row past the tensor's end is zero x A + B: zeroed by a bit mask (`real ? fma : 0` compiles to a branch) */ \
      const uint32_t keep = rowg0_ + row < M32 ? 0xffffffffu : 0u;                                                  \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                 \
        f[e] = __uint_as_float(__float_as_uint(fmaf(cA[e], f[e], fmaf(cC[e], f2[e], cB[e]))) & keep);               \
      Vec8<bf16_t>::store(dst, f);                                                                                  \
      rawp[r] = cd_load(rX, p_go[r] + bpn_);   /* (past this workgroup's last row: zeros, no memory access) */       \
      rawp2[r] = cd_load(rX2, p_go[r] + bpn_);                                                                      \
    }                                                                                                               \
    _Pragma("unroll") for (int r = 0; r < RQ; ++r) {                                                                \
      if (r == RQ - 1 && !q_last) continue;                                                                         \
      const int row = q_desc[r] >> 5, v = q_desc[r] & 31;                                                           \
      bf16_t* dst = q_go[r] != CD_OOB ? (CQ) + row * QL + v * 8 : dump;                                             \
      *reinterpret_cast<uint4*>(dst) = rawq[r];                                                                     \
      rawq[r] = cd_load(rQ, q_go[r] + bqn_);                                                                        \
    }                                                                                                               \
  }
  // data gradient of one tile: A = weight fragment (packed image), B = data fragment (row-major P tile)
#define CD_MULT(CA)                                                                                                 \
  {                                                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                             \
      const uint4 xb = *reinterpret_cast<const uint4*>((CA) + xfrag + ks * 32);                                     \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                                             \
        const uint4 wa = *reinterpret_cast<const uint4*>(wfrag + ((size_t)ks * 4 * L.img_rows + t * 16) * 8);       \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0); \
      }                                                                                                             \
    }                                                                                                               \
  }
  // weight gradient of one tile: rows are the contraction index (k-steps of 32 rows)
#define CD_WGRAD(CA, CQ)                                                                                            \
  {                                                                                                                 \
    _Pragma("unroll") for (int k2 = 0; k2 < K2; ++k2) {                                                             \
      uint4 pa[NPW], qb[NQW];                                                                                       \
      const bf16_t* pb_ = (CA) + k2 * 32 * KL + pl;                                                                 \
      const bf16_t* qb_ = (CQ) + k2 * 32 * QL + ql;                                                                 \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                             \
        const cd_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(pb_ + i * 16));        \
        const cd_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(pb_ + 16 * KL + i * 16)); \
        pa[i] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));               \
      }                                                                                                             \
      _Pragma("unroll") for (int j = 0; j < NQW; ++j) {                                                             \
        const cd_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(qb_ + j * 16));        \
        const cd_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(qb_ + 16 * QL + j * 16)); \
        qb[j] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));               \
      }                                                                                                             \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NQW; ++j)                                                             \
          dacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[i]), __builtin_bit_cast(bf16x8_t, qb[j]), \
                                                               dacc[i][j], 0, 0, 0);                               \
    }                                                                                                               \
  }
  // result tile -> the wave's staging rows ([row = lane & 15][channel], bf16 as the first kernel stages it) -> 16-byte row
  // vectors: + res, ReLU mask of y_prev (read from its LDS tile), the previous block's BatchNorm_c-backward sums, store
#define CD_EPI(TILE, CQ, NEXT_ON)                                                                                   \
  {                                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                                 \
      *reinterpret_cast<uint2*>(Os + (lane & 15) * NLW + t * 16 + (lane >> 4) * 4) =                                \
          make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));                         \
    const int row0_ = (TILE) * MT + wr * 16;                                                                        \
    _Pragma("unroll") for (int p = 0; p < NPASS; ++p) {                                                             \
      const int row = p * RPO + rr_o;                                                                               \
      const int rowc = row < 16 ? row : 0;                                                                          \
      const int m = row0_ + row;                                                                                    \
      const bool ok = act_o && row < 16 && m < M32;                                                                 \
      const uint32_t keep = ok ? 0xffffffffu : 0u;                                                                  \
      const uint4 rawo = *reinterpret_cast<const uint4*>(Os + rowc * NLW + v_o * 8);                                \
      const uint4 x3 = *reinterpret_cast<const uint4*>((CQ) + (wr * 16 + rowc) * QL + (act_o ? cvec : 0) * 8);      \
      float f[8], rv[8], cv[8], eM[8], eR[8];                                                                       \
      cd_cvt(rawo, f); cd_cvt(e1r[p], rv); cd_cvt(c1r[p], cv);                                                      \
      cd_ld8(Ep + (act_o ? cvec : 0) * 8, eM); cd_ld8(Ep + Np + (act_o ? cvec : 0) * 8, eR);                        \
      const uint32_t xw[4] = {x3.x, x3.y, x3.z, x3.w};                                                              \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                               \
        const uint32_t hx = (xw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;                                               \
        /* y_prev > 0 (a ReLU output: nonzero magnitude, sign clear); without the mask every element passes */      \
        const uint32_t pass = ((hx & 0x7fffu) != 0u && !(hx & 0x8000u)) ? 0xffffffffu : ~mask_on;                   \
        const float d = __uint_as_float(__float_as_uint(f[j] + rv[j]) & pass);                                      \
        f[j] = d;                                                                                                   \
        const float gq = __uint_as_float(__float_as_uint(round_as<bf16_t>(d)) & keep & sums_on);                    \
        s0[j] += gq; s1[j] = fmaf(gq, (cv[j] - eM[j]) * eR[j], s1[j]);                                              \
      }                                                                                                             \
      const uint4 pk = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])); \
      __builtin_amdgcn_raw_buffer_store_b128(cd_u32x4_t{pk.x, pk.y, pk.z, pk.w}, rY,                                \
                                             ok ? ((uint32_t)m * (uint32_t)Np + (uint32_t)cvec * 8u) * 2u : CD_OOB, 0, 0); \
      const uint32_t no_ = (NEXT_ON) ? CD_EOFF((TILE) + 1, p) : CD_OOB;                                             \
      e1r[p] = cd_load(rE1, no_); c1r[p] = cd_load(rC1, no_);                                                       \
    }                                                                                                               \
  }

  bf16_t* const bufP0 = reinterpret_cast<bf16_t*>(smem + L.a_off);
  bf16_t* const bufP1 = reinterpret_cast<bf16_t*>(smem + L.a_off + L.a_bytes);
  bf16_t* const bufQ0 = reinterpret_cast<bf16_t*>(smem + L.q_off);
  bf16_t* const bufQ1 = reinterpret_cast<bf16_t*>(smem + L.q_off + L.q_bytes);
  if (t0 < t1) CD_CONVERT(t0, bufP0, bufQ0)
  int cur = 0;
  for (int tile = t0; tile < t1; ++tile, cur ^= 1) {
    bf16_t* const curP = cur ? bufP1 : bufP0;
    bf16_t* const nxtP = cur ? bufP0 : bufP1;
    bf16_t* const curQ = cur ? bufQ1 : bufQ0;
    bf16_t* const nxtQ = cur ? bufQ0 : bufQ1;
    __syncthreads();   // the only barrier per tile: tile `tile` is complete, every wave is past the products that read the other buffers
    f32x4_t acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (tile + 1 < t1) {
      // both products + store + sums of this tile and the conversion of the next one: ONE basic block
      CD_MULT(curP)
      CD_WGRAD(curP, curQ)
      CD_CONVERT(tile + 1, nxtP, nxtQ)
      CD_EPI(tile, curQ, true)
#ifndef CD_VPM
#define CD_VPM 10
#endif
#ifndef CD_SCHED
#define CD_SCHED 0   // (measured: the compiler's own order of this block is 0.09 ms per step better than the paced one of csrc/pw_cfwd.hip)
#endif
#pragma unroll
      for (int i = 0; i < (CD_SCHED ? NTW * KS + K2 * NPW * NQW : 0); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, CD_VPM, 0);   // VALU instructions of the conversion / the epilogue
        if (i & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a re-request
      }
    } else {
      CD_MULT(curP)
      CD_WGRAD(curP, curQ)
      CD_EPI(tile, curQ, false)
    }
  }
#undef CD_CONVERT
#undef CD_MULT
#undef CD_WGRAD
#undef CD_EPI
#undef CD_EOFF

  // ---- this workgroup's dW partial -> wg_ws[blockIdx.x][K][N] (the reducer launched behind this kernel adds the partials in
  // fixed order into wg_dw)
  {
    float* wsb = a.wg_ws + (size_t)blockIdx.x * a.K * a.N;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
#pragma unroll
      for (int j = 0; j < NQW; ++j) {
        const int n = (qg * NQW + j) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = (pg * NPW + i) * 16 + (lane >> 4) * 4 + r;
          if (k < a.K && n < a.N) wsb[(size_t)k * a.N + n] = dacc[i][j][r];
        }
      }
    }
  }

  // ---- BatchNorm_c-backward sums of the previous block (single set, f64 [2][N]: the layout c3d_block_out_bwd fills): lanes
  // -> LDS ([value][lane] per wave; the converted tiles are dead) -> one thread per (which, channel) adds the row-lanes of the
  // WR waves of its column group -> ONE f64 atomic per value and workgroup
  if (a.add_sums) {
    __syncthreads();
    float* mine = reinterpret_cast<float*>(smem + L.a_off) + (size_t)wave * 16 * 64;   // [16][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j * 64 + lane] = s0[j]; mine[(8 + j) * 64 + lane] = s1[j]; }
    __syncthreads();
    for (int i = tid; i < 2 * a.N; i += CD_THREADS) {
      const int which = i / a.N, c = i - which * a.N;
      const int cv = c >> 3, j = c & 7;                    // channel = vector cv, element j
      const int wcg = cv / GOW, vo = cv - wcg * GOW;       // column group of the vector, its index inside the group
      float accv = 0.f;
      for (int w_ = 0; w_ < L.WR; ++w_) {
        const float* base = reinterpret_cast<const float*>(smem + L.a_off) + (size_t)(wcg * L.WR + w_) * 16 * 64 + (which * 8 + j) * 64;
        for (int rr = 0; rr < RPO; ++rr) accv += base[rr * GOW + vo];
      }
      atomicAdd(a.add_sums + which * a.N + c, (double)accv);
    }
  }
}

template <int KS, int NTW, int NPW, int NQW, int RP, int RQ, int K2>
int cd_launch(const c3d_pw_args& a, const CdPlan& L, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_cdg_a_kernel<KS, NTW, NPW, NQW, RP, RQ, K2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  pw_cdg_a_kernel<KS, NTW, NPW, NQW, RP, RQ, K2><<<grid, dim3(CD_THREADS), lds, s>>>(a, L);
  return 0;
}

// shape -> instantiation: 0 = none.  1: 216 -> 96 (32 x 32 stage), 2: 108 -> 48, 3: 54 -> 24
int cd_variant(int Kp, int Np) {
  const int KS = (Kp + 31) / 32, ntn = (Np + 15) >> 4, ntp = (Kp + 15) >> 4;
  if (KS == 7 && ntn >= 5 && ntn <= 6 && ntp <= 16) return 1;
  if (KS == 4 && ntn == 3 && ntp <= 8) return 2;
  if (KS == 2 && ntn == 2 && ntp <= 4) return 3;
  return 0;
}

int cd_plan(const c3d_pw_args& a, CdPlan& L, int64_t& blocks, size_t& lds) {
  const int var = cd_variant(a.Kp, a.Np);
  if (!var) return 0;
  const int Kpad = (a.Kp + 31) / 32 * 32, KS = Kpad / 32, ntn = (a.Np + 15) >> 4;
  L.img_rows = (ntn <= 2 ? 2 : ntn <= 4 ? 4 : ntn <= 7 ? 7 : 14) * 16;
  int NTW;
  // (216 -> 96: 32-row tiles, 2 row slabs x 4 column groups of two output tiles -- the fourth group is padding: with 64-row tiles
  // the prefetch registers of three row streams beside 48 weight-gradient accumulators spilled, and LDS was full)
  if (var == 1) { L.WR = 2; L.WC = 4; NTW = 2; L.QG = 2; }
  else if (var == 2) { L.WR = 8; L.WC = 1; NTW = 3; L.QG = 1; }
  else { L.WR = 8; L.WC = 1; NTW = 2; L.QG = 2; }
  L.MT = 16 * L.WR;
  L.KL = Kpad + 8;
  L.QL = ntn * 16 + 8;
  const int64_t tiles = (a.M + L.MT - 1) / L.MT;
  blocks = device_cus();
  if (blocks > (tiles + 1) / 2) blocks = (tiles + 1) / 2;
  if (blocks > CD_MAX_PARTS) blocks = CD_MAX_PARTS;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  L.tiles_per_wg = tpw;
  auto al = [](size_t v) { return (v + 1023) / 1024 * 1024; };
  size_t off = 0;
  L.w_off = 0; off += al((size_t)KS * 4 * L.img_rows * 16);
  L.a_off = (int)off; L.a_bytes = (int)al((size_t)L.MT * L.KL * 2); off += 2 * (size_t)L.a_bytes;
  L.q_off = (int)off; L.q_bytes = (int)al((size_t)L.MT * L.QL * 2); off += 2 * (size_t)L.q_bytes;
  if (2 * (size_t)L.a_bytes + 2 * (size_t)L.q_bytes < (size_t)8 * 16 * 64 * 4) return 0;   // (the statistics dump at the end reuses the tiles)
  L.os_wave = 16 * (NTW * 16 + 8) * 2; L.os_off = (int)off; off += al((size_t)8 * L.os_wave);
  L.par_off = (int)off; off += al(((size_t)3 * a.Kp + 2 * a.Np) * 4);
  L.dump_off = (int)off; off += (size_t)CD_THREADS * 16;
  if (off > 160 * 1024) return 0;
  lds = off;
  return var;
}


// =====================================================================================================================
// conv_c DATA gradient with its weight gradient, same loop:
//
//     P[m, k]   = A[k] * g[m, k] + B[k] + C[k] * c[m, k]                      (BatchNorm_c backward on load)
//     d[m, n]   = sum_k P[m, k] * W_c[k, n]                                   (gradient at the Swish output)
//     pb = b * scale + shift, q = gate * pb, sg = sigmoid(q):   dq = d * sg * (1 + q (1 - sg)),   t1 = dq * gate  (stored)
//     per (sample, channel): sum dq * pb (d gate), sum t1, sum t1 * bhat       (SE / BatchNorm_b backward)
//     dW_c[k, n] += sum_m P[m, k] * (q sg)[m, n]                              (q sg = the forward operand of conv_c)
//
// (reference model/x3d.py:203-216 backward; the C3D_PRO_AFFINE2 / C3D_EPI_SWISH_SE_BWD / C3D_WG_SWISH form of c3d_pw_gemm).
// The weight gradient's second operand is a PRODUCT of the epilogue: the epilogue of tile t writes its q sg rows (bf16, as the
// first kernel's fused variant does) into a row-major LDS tile, and the weight-gradient MFMAs of tile t run in the block of
// tile t + 1 (behind that iteration's barrier) -- three P tiles, two q sg tiles.  A tile lies inside one sample
// (rows_per_sample is a multiple of the tile's rows: else the first kernel takes the call); the per-lane sums are combined by
// row-lane shuffles and one small LDS pass when the sample changes and at the end.
// Results: t1 equals the first kernel's bit for bit on the 48- and 24-channel layers; at 96 -> 216 one element in ~600 000 is
// one bf16 ulp away (same instruction chain in both disassemblies; deterministic; tests/test_pw_wg_gpu.py bounds it); sums and
// dW agree to f32 rounding.
struct CcPlan {
  int MT, WR, WC, QG;
  int tiles_per_wg;
  int KL, QL;
  int img_rows;
  int w_off, a_off, a_bytes, q_off, q_bytes, os_off, os_wave, par_off, red_off, dump_off;
};

template <int KS, int NPW, int NQW, int K2>
__global__ __launch_bounds__(CD_THREADS) void pw_cdg_c_kernel(const c3d_pw_args a, const CcPlan L) {
  constexpr int NTW = 4;                                  // output tiles per wave (all three widths): 8 vectors x 8 row-lanes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CDCLK(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kp = a.Kp, Np = a.Np, Gq = Kp >> 3, MT = L.MT, KL = L.KL, QL = L.QL;
  bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem + L.w_off);
  float* const Pp = reinterpret_cast<float*>(smem + L.par_off);       // A | B | C [Kp]
  float* const Ep = Pp + 3 * Kp;                                      // scale | shift | mean | rstd [Np] of BatchNorm_b
  float* const red = reinterpret_cast<float*>(smem + L.red_off);      // [8 waves][24][8]
  float* const Gs = Ep + 4 * Np;                                      // SE gate of the current sample [Np] (ones without SE)
  bf16_t* const dump = reinterpret_cast<bf16_t*>(smem + L.dump_off) + tid * 8;

  const int M32 = (int)a.M;
  const int tiles = (M32 + MT - 1) / MT;
  int t0 = (int)blockIdx.x * L.tiles_per_wg;
  if (t0 > tiles) t0 = tiles;
  int t1 = t0 + L.tiles_per_wg;
  if (t1 > tiles) t1 = tiles;
  const uint32_t row_hi = (uint32_t)(t1 * MT < M32 ? t1 * MT : M32);
  const __amdgpu_buffer_rsrc_t rX = cd_rsrc(a.x, row_hi * (uint32_t)Kp * 2u);
  const __amdgpu_buffer_rsrc_t rX2 = cd_rsrc(a.x2, row_hi * (uint32_t)Kp * 2u);
  const __amdgpu_buffer_rsrc_t rE1 = cd_rsrc(a.e1, (uint32_t)M32 * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rY = cd_rsrc(a.y, (uint32_t)M32 * (uint32_t)Np * 2u);
  const uint32_t rps = (uint32_t)a.rows_per_sample;

  // ---- item map of the P streams (one round: MT x Kp / 8 <= 512 items)
  int p_desc;
  uint32_t p_go;
  {
    const int row = __float2int_rz(((float)tid + 0.5f) * (1.0f / (float)Gq));
    const bool ok = tid < MT * Gq;
    p_desc = ok ? (row << 5) | (tid - row * Gq) : 0;
    p_go = ok ? (uint32_t)tid * 16u : CD_OOB;
  }
  const uint32_t tbp = (uint32_t)(MT * Kp * 2);
  // two tiles of the row streams in flight (the waves of these kernels are parked ~50 % of their cycles with one): register set
  // s holds tile t0 + s (+ 2, + 4 ..); the tile loop is unrolled by two so that the sets are named statically
  uint4 rawp[2], rawp2[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) { rawp[s_] = cd_load(rX, p_go + (uint32_t)(t0 + s_) * tbp); rawp2[s_] = cd_load(rX2, p_go + (uint32_t)(t0 + s_) * tbp); }

  // ---- lane maps of the data gradient and of its epilogue
  const int wr = wave % L.WR, wc = wave / L.WR;
  const int nt0 = wc * NTW;
  constexpr int NLW = NTW * 16 + 8;
  bf16_t* const Os = reinterpret_cast<bf16_t*>(smem + L.os_off + wave * L.os_wave);
  constexpr int GOW = NTW * 2, RPO = 64 / GOW, NPASS = 16 / RPO;   // 8 vectors x 8 row-lanes, two passes
  const int rr_o = lane >> 3, v_o = lane & 7;
  const int cvec = nt0 * 2 + v_o;
  const bool act_o = cvec * 8 < Np;
  const int cve = act_o ? cvec : 0;                       // (parameter / gate reads of lanes beyond the row stay inside the arrays)
  uint4 e1r[2][NPASS];
#define CC_EOFF(TILE, P) ((act_o && (TILE) < t1 && (TILE) * MT + wr * 16 + (P) * RPO + rr_o < M32)                                                \
                              ? ((uint32_t)((TILE) * MT + wr * 16 + (P) * RPO + rr_o) * (uint32_t)Np + (uint32_t)cvec * 8u) * 2u : CD_OOB)
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) e1r[s_][p] = cd_load(rE1, CC_EOFF(t0 + s_, p));

  // ---- weight image -> LDS
  {
    const int wbytes = (KS * 4) * L.img_rows * 16;
    const int nchunk = (wbytes + 1023) >> 10;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w_img);
    const int rot = (int)blockIdx.x % nchunk;
    for (int c = wave; c < nchunk; c += 8) {
      int r = c + rot;
      if (r >= nchunk) r -= nchunk;
      const int off = r * 1024 + lane * 16;
      if (off < wbytes)
        __builtin_amdgcn_global_load_lds((cd_glb_ptr_t)(src + off), (cd_lds_ptr_t)(smem + L.w_off + r * 1024), 16, 0, 0);
    }
  }
  // the five tiles zeroed once (k padding of P; columns of q sg nobody writes)
  for (int i = tid * 16; i < 3 * L.a_bytes; i += CD_THREADS * 16) *reinterpret_cast<uint4*>(smem + L.a_off + i) = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid * 16; i < 2 * L.q_bytes; i += CD_THREADS * 16) *reinterpret_cast<uint4*>(smem + L.q_off + i) = make_uint4(0u, 0u, 0u, 0u);
  // ---- BatchNorm_c-backward coefficients (csrc/bn_fin.h: workgroup 0 accumulates d gamma / d beta) and the epilogue's vectors
  if (a.fin.sums) {
    for (int c = tid; c < Kp; c += CD_THREADS) {
      float cA, cB, cC;
      c3dfin::bn_bwd_coef_consume(a.fin, a.K, Kp, c, blockIdx.x == 0, cA, cB, cC);
      Pp[c] = cA; Pp[Kp + c] = cB; Pp[2 * Kp + c] = cC;
    }
  } else {
    for (int i = tid; i < 3 * Kp; i += CD_THREADS) Pp[i] = a.pro_p[i];
  }
  for (int i = tid; i < 2 * Np; i += CD_THREADS) { Ep[i] = a.epi_p[i]; Ep[2 * Np + i] = a.epi_q[i]; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  CDCLK(1)

  float s0[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  int cur_n = -1;
  const bf16_t* const wfrag = Ws + ((size_t)(lane >> 4) * L.img_rows + nt0 * 16 + (lane & 15)) * 8;
  const int xfrag = (wr * 16 + (lane & 15)) * KL + (lane >> 4) * 8;
  const int qg = wave % L.QG, pg = wave / L.QG;
  const int g4 = lane >> 4, li = lane & 15;
  const int pl = (4 * g4 + (li >> 2)) * KL + 4 * (li & 3) + pg * NPW * 16;
  const int ql = (4 * g4 + (li >> 2)) * QL + 4 * (li & 3) + qg * NQW * 16;
  f32x4_t dacc[NPW][NQW];
#pragma unroll
  for (int i = 0; i < NPW; ++i)
#pragma unroll
    for (int j = 0; j < NQW; ++j) dacc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // per-(sample, channel) sums of the sample `cur_n`: row-lanes by shuffles, the WR waves of a column group through LDS, one f64
  // atomic per value and workgroup (every thread of the workgroup calls this)
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r0 = s0[j], r1 = s1[j], r2 = s2[j];
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { r0 += __shfl_xor(r0, o, 64); r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); }
      if (lane < 8) { red[(wave * 24 + j) * 8 + lane] = r0; red[(wave * 24 + 8 + j) * 8 + lane] = r1; red[(wave * 24 + 16 + j) * 8 + lane] = r2; }
      s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 3 * Np; i += CD_THREADS) {
      const int which = i / Np, c = i - which * Np;
      const int cv = c >> 3, j = c & 7;
      const int wcg = cv >> 3, vo = cv & 7;
      float accv = 0.f;
      for (int w_ = 0; w_ < L.WR; ++w_) accv += red[((wcg * L.WR + w_) * 24 + which * 8 + j) * 8 + vo];
      atomicAdd(a.stats + ((int64_t)cur_n * Np + c) * 3 + which, (double)accv);
    }
    __syncthreads();
  };

#define CC_CONVERT(TILE, CP, S)                                                                                      \
  {                                                                                                                 \
    const int rowg0_ = (TILE) * MT;                                                                                 \
    const uint32_t bpn_ = (uint32_t)((TILE) + 2) * tbp;   /* the slot is requested again for the tile after the next */  \
    const int row = p_desc >> 5, v = p_desc & 31;                                                                   \
    bf16_t* dst = p_go != CD_OOB ? (CP) + row * KL + v * 8 : dump;                                                  \
    float f[8], f2[8], cA[8], cB[8], cC[8];                                                                         \
    cd_cvt(rawp[S], f); cd_cvt(rawp2[S], f2);                                                                       \
    cd_ld8(Pp + v * 8, cA); cd_ld8(Pp + Kp + v * 8, cB); cd_ld8(Pp + 2 * Kp + v * 8, cC);                           \
    const uint32_t keep = rowg0_ + row < M32 ? 0xffffffffu : 0u;                                                    \
    _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                                   \
      f[e] = __uint_as_float(__float_as_uint(fmaf(cA[e], f[e], fmaf(cC[e], f2[e], cB[e]))) & keep);                 \
    Vec8<bf16_t>::store(dst, f);                                                                                    \
    rawp[S] = cd_load(rX, p_go + bpn_);                                                                             \
    rawp2[S] = cd_load(rX2, p_go + bpn_);                                                                           \
  }
#define CC_MULT(CA)                                                                                                 \
  {                                                                                                                 \
    /* two output tiles at a time, each pair staged to the wave's result rows at once (8 accumulator registers live, not 16) */ \
    _Pragma("unroll") for (int th = 0; th < NTW; th += 2) {                                                         \
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};                                  \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                           \
        const uint4 xb = *reinterpret_cast<const uint4*>((CA) + xfrag + ks * 32);                                   \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                             \
          const uint4 wa = *reinterpret_cast<const uint4*>(wfrag + ((size_t)ks * 4 * L.img_rows + (th + t) * 16) * 8); \
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0); \
        }                                                                                                           \
      }                                                                                                             \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                 \
        *reinterpret_cast<uint2*>(Os + (lane & 15) * NLW + (th + t) * 16 + (lane >> 4) * 4) =                       \
            make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));                       \
    }                                                                                                               \
  }
#define CC_WGRAD(CA, CQ)                                                                                            \
  {                                                                                                                 \
    _Pragma("unroll") for (int k2 = 0; k2 < K2; ++k2) {                                                             \
      uint4 pa[NPW], qb[NQW];                                                                                       \
      const bf16_t* pb_ = (CA) + k2 * 32 * KL + pl;                                                                 \
      const bf16_t* qb_ = (CQ) + k2 * 32 * QL + ql;                                                                 \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                             \
        const cd_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(pb_ + i * 16));        \
        const cd_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(pb_ + 16 * KL + i * 16)); \
        pa[i] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));               \
      }                                                                                                             \
      _Pragma("unroll") for (int j = 0; j < NQW; ++j) {                                                             \
        const cd_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(qb_ + j * 16));        \
        const cd_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cd_lds_s16x4_ptr_t)(qb_ + 16 * QL + j * 16)); \
        qb[j] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));               \
      }                                                                                                             \
      _Pragma("unroll") for (int i = 0; i < NPW; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NQW; ++j)                                                             \
          dacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[i]), __builtin_bit_cast(bf16x8_t, qb[j]), \
                                                               dacc[i][j], 0, 0, 0);                               \
    }                                                                                                               \
  }
  // result tile -> staging rows (bf16) -> row vectors: Swish / SE backward (the first kernel's arithmetic), t1 stored, q sg
  // left in the LDS tile CQ for the weight gradient
#define CC_EPI(TILE, CQ, S)                                                                                         \
  {                                                                                                                 \
    const int row0_ = (TILE) * MT + wr * 16;                                                                        \
    _Pragma("unroll") for (int p = 0; p < NPASS; ++p) {                                                             \
      const int row = p * RPO + rr_o;                                                                               \
      const int m = row0_ + row;                                                                                    \
      const bool ok = act_o && m < M32;                                                                             \
      const uint32_t keep = ok ? 0xffffffffu : 0u;                                                                  \
      const uint4 rawo = *reinterpret_cast<const uint4*>(Os + row * NLW + v_o * 8);                                 \
      float f[8], bv[8], qs[8];                                                                                     \
      cd_cvt(rawo, f); cd_cvt(e1r[S][p], bv);                                                                       \
      /* four channels at a time, the halves and the passes in program order (sched_barrier): the 7 x 8 temporaries of a whole   \
         vector beside the weight-gradient accumulators spilled 77 registers at 96 -> 216 */                       \
      _Pragma("unroll") for (int h = 0; h < 8; h += 4) {                                                            \
        const float4 eS = *reinterpret_cast<const float4*>(Ep + cve * 8 + h), eB = *reinterpret_cast<const float4*>(Ep + Np + cve * 8 + h); \
        const float4 eM = *reinterpret_cast<const float4*>(Ep + 2 * Np + cve * 8 + h), eR = *reinterpret_cast<const float4*>(Ep + 3 * Np + cve * 8 + h); \
        const float eS_[4] = {eS.x, eS.y, eS.z, eS.w}, eB_[4] = {eB.x, eB.y, eB.z, eB.w};                           \
        const float eM_[4] = {eM.x, eM.y, eM.z, eM.w}, eR_[4] = {eR.x, eR.y, eR.z, eR.w};                           \
        const float4 eGv = *reinterpret_cast<const float4*>(Gs + cve * 8 + h);                                      \
        const float eG_[4] = {eGv.x, eGv.y, eGv.z, eGv.w};                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                             \
          const int j = h + u;                                                                                      \
          const float pb = fmaf(bv[j], eS_[u], eB_[u]);                                                             \
          const float q = eG_[u] * pb;                                                                              \
          const float sg = sigmoid_t<bf16_t>(q);                                                                    \
          const float dq = __uint_as_float(__float_as_uint(f[j] * sg * (1.f + q * (1.f - sg))) & keep);             \
          const float t = round_as<bf16_t>(dq * eG_[u]);                                                            \
          s0[j] += dq * pb;                                                                                         \
          s1[j] += t;                                                                                               \
          s2[j] += t * ((bv[j] - eM_[u]) * eR_[u]);                                                                 \
          f[j] = t;                                                                                                 \
          qs[j] = __uint_as_float(__float_as_uint(q * sg) & keep);                                                  \
        }                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
      }                                                                                                             \
      const uint4 pk = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])); \
      __builtin_amdgcn_raw_buffer_store_b128(cd_u32x4_t{pk.x, pk.y, pk.z, pk.w}, rY,                                \
                                             ok ? ((uint32_t)m * (uint32_t)Np + (uint32_t)cvec * 8u) * 2u : CD_OOB, 0, 0); \
      Vec8<bf16_t>::store((CQ) + (wr * 16 + row) * QL + cvec * 8, qs);                                              \
      e1r[S][p] = cd_load(rE1, CC_EOFF((TILE) + 2, p));                                                             \
    }                                                                                                               \
  }

#ifndef CC_SECTIONS
#define CC_SECTIONS 1
#endif
#if CC_SECTIONS
#define CC_SECTION __builtin_amdgcn_sched_barrier(0);
#else
#define CC_SECTION
#endif
  bf16_t* pP = reinterpret_cast<bf16_t*>(smem + L.a_off);                     // P of the tile before
  bf16_t* cP = reinterpret_cast<bf16_t*>(smem + L.a_off + L.a_bytes);         // P of this tile
  bf16_t* nP = reinterpret_cast<bf16_t*>(smem + L.a_off + 2 * L.a_bytes);     // P of the next tile
  bf16_t* const bufQ0 = reinterpret_cast<bf16_t*>(smem + L.q_off);
  bf16_t* const bufQ1 = reinterpret_cast<bf16_t*>(smem + L.q_off + L.q_bytes);
  bf16_t* lastQ = bufQ0;
  if (t0 < t1) CC_CONVERT(t0, cP, 0)
  CDCLK(2)
  // One iteration, the same for every tile: data gradient + epilogue of this tile, weight gradient of the tile before (the first
  // tile multiplies the zeroed buffers), conversion of the next (past the workgroup's last tile: rows the stream bounds answer
  // with zeros, written to a buffer nobody reads) -- straight-line code, no first / last copies.
#define CC_ITER(TILE, S, CURQ, PRVQ)                                                                                \
  {                                                                                                                 \
    const int n_tile = (int)((uint32_t)((TILE) * MT) / rps);                                                        \
    if (n_tile != cur_n) {   /* (workgroup-uniform; once per sample) */                                             \
      if (cur_n >= 0) flush();                                                                                      \
      cur_n = n_tile;                                                                                               \
      /* the sample's gate -> LDS (read per half vector by the epilogue: as 8 registers per lane it was the last 9 spilled ones) */ \
      for (int i = tid; i < Np; i += CD_THREADS) Gs[i] = a.epi_gate ? a.epi_gate[(int64_t)cur_n * Np + i] : 1.f;    \
    }                                                                                                               \
    __syncthreads();   /* tile TILE is converted, the q sg rows of the tile before are complete, every wave is past the products of the iteration before */ \
    CC_MULT(cP)                                                                                                     \
    CC_SECTION                                                                                                      \
    CC_WGRAD(pP, PRVQ)                                                                                              \
    CC_SECTION                                                                                                      \
    CC_CONVERT((TILE) + 1, nP, 1 - (S))                                                                             \
    CC_SECTION                                                                                                      \
    CC_EPI(TILE, CURQ, S)                                                                                           \
    bf16_t* const tmp = pP; pP = cP; cP = nP; nP = tmp;                                                             \
    lastQ = (CURQ);                                                                                                 \
  }
  for (int tile = t0; tile < t1; tile += 2) {
    CC_ITER(tile, 0, bufQ0, bufQ1)
    if (tile + 1 < t1) CC_ITER(tile + 1, 1, bufQ1, bufQ0)
  }
#undef CC_ITER
  CDCLK(3)
  if (t0 < t1) {
    __syncthreads();
    CC_WGRAD(pP, lastQ)   // the last tile's
  }
#undef CC_CONVERT
#undef CC_MULT
#undef CC_WGRAD
#undef CC_EPI
#undef CC_EOFF

  {
    float* wsb = a.wg_ws + (size_t)blockIdx.x * a.K * a.N;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
#pragma unroll
      for (int j = 0; j < NQW; ++j) {
        const int n = (qg * NQW + j) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = (pg * NPW + i) * 16 + (lane >> 4) * 4 + r;
          if (k < a.K && n < a.N) wsb[(size_t)k * a.N + n] = dacc[i][j][r];
        }
      }
    }
  }
  CDCLK(4)
  if (cur_n >= 0) flush();
  CDCLK(5)
}

template <int KS, int NPW, int NQW, int K2>
int cc_launch(const c3d_pw_args& a, const CcPlan& L, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_cdg_c_kernel<KS, NPW, NQW, K2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  pw_cdg_c_kernel<KS, NPW, NQW, K2><<<grid, dim3(CD_THREADS), lds, s>>>(a, L);
  return 0;
}

// 1: 96 -> 216 (32 x 32 stage), 2: 48 -> 108, 3: 24 -> 54
int cc_variant(int Kp, int Np) {
  const int KS = (Kp + 31) / 32, ntn = (Np + 15) >> 4, ntp = (Kp + 15) >> 4;
  if (KS == 3 && ntn >= 13 && ntn <= 14 && ntp <= 6) return 1;
  if (KS == 2 && ntn == 7 && ntp <= 3) return 2;
  if (KS == 1 && ntn == 4 && ntp <= 2) return 3;
  return 0;
}

int cc_plan(const c3d_pw_args& a, CcPlan& L, int64_t& blocks, size_t& lds) {
  const int var = cc_variant(a.Kp, a.Np);
  if (!var) return 0;
  const int Kpad = (a.Kp + 31) / 32 * 32, KS = Kpad / 32, ntn = (a.Np + 15) >> 4;
  L.img_rows = (ntn <= 2 ? 2 : ntn <= 4 ? 4 : ntn <= 7 ? 7 : 14) * 16;
  if (var == 1) { L.WR = 2; L.WC = 4; L.QG = 4; }
  else if (var == 2) { L.WR = 4; L.WC = 2; L.QG = 8; }
  else { L.WR = 8; L.WC = 1; L.QG = 4; }
  L.MT = 16 * L.WR;
  if ((L.MT * (a.Kp >> 3)) > CD_THREADS) return 0;
  if (a.rows_per_sample <= 0 || a.rows_per_sample % L.MT) return 0;   // a tile inside one sample
  L.KL = Kpad + 8;
  L.QL = L.WC * 4 * 16 + 8;
  const int64_t tiles = (a.M + L.MT - 1) / L.MT;
  blocks = device_cus();
  if (blocks > (tiles + 1) / 2) blocks = (tiles + 1) / 2;
  if (blocks > CD_MAX_PARTS) blocks = CD_MAX_PARTS;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  L.tiles_per_wg = tpw;
  auto al = [](size_t v) { return (v + 1023) / 1024 * 1024; };
  size_t off = 0;
  L.w_off = 0; off += al((size_t)KS * 4 * L.img_rows * 16);
  L.a_off = (int)off; L.a_bytes = (int)al((size_t)L.MT * L.KL * 2); off += 3 * (size_t)L.a_bytes;
  L.q_off = (int)off; L.q_bytes = (int)al((size_t)L.MT * L.QL * 2); off += 2 * (size_t)L.q_bytes;
  L.os_wave = 16 * (4 * 16 + 8) * 2; L.os_off = (int)off; off += al((size_t)8 * L.os_wave);
  L.par_off = (int)off; off += al(((size_t)3 * a.Kp + 5 * a.Np) * 4);
  L.red_off = (int)off; off += al((size_t)8 * 24 * 8 * 4);
  L.dump_off = (int)off; off += (size_t)CD_THREADS * 16;
  if (off > 160 * 1024) return 0;
  lds = off;
  return var;
}

}  // namespace

// Host-side check for the stage driver: would c3d_detail_pw_cdg_a take this layer (bf16, dense rows, res_mode 0)?
__attribute__((visibility("hidden"))) bool c3d_detail_pw_cdg_a_supported(int Kp, int Np, int64_t M) {
  if (Kp <= 0 || Np <= 0 || (Kp & 7) || (Np & 7)) return false;
  if (M < 1024 || (M + 512) * (int64_t)(Kp > Np ? Kp : Np) * 2 >= ((int64_t)1 << 31)) return false;
  c3d_pw_args a;
  std::memset(&a, 0, sizeof(a));
  a.M = M; a.K = a.Kp = Kp; a.N = a.Np = Np;
  CdPlan L;
  int64_t blocks = 0;
  size_t lds = 0;
  return cd_plan(a, L, blocks, lds) != 0;
}

// Returns C3D_E_UNSUPPORTED for what it does not take (c3d_pw_gemm then runs the first kernel's fused variant).
__attribute__((visibility("hidden"))) int c3d_detail_pw_cdg_a(const c3d_pw_args* args, void* stream) {
  const c3d_pw_args& a = *args;
  if (a.dtype != C3D_DT_BF16 || a.row_mode != C3D_ROWS_DENSE || a.pro_mode != C3D_PRO_AFFINE2 || a.epi_mode != C3D_EPI_ADD ||
      a.wg_mode != C3D_WG_ROWS || a.res_mode != 0)
    return C3D_E_UNSUPPORTED;
  if (!a.w_img || !a.x2 || !a.e1 || !a.wg_x3 || !a.wg_dw || !a.wg_ws || a.pro_out || a.bias || a.fin.ticket) return C3D_E_UNSUPPORTED;
  if (a.fin.sums ? (a.fin.training != 0 || !a.fin.mr || !a.fin.gamma) : !a.pro_p) return C3D_E_UNSUPPORTED;
  if (a.add_sums && (!a.add_c || !a.add_mr || !a.wg_mask_out)) return C3D_E_UNSUPPORTED;
  if (a.M < 1024 || (a.M + 512) * (int64_t)(a.Kp > a.Np ? a.Kp : a.Np) * 2 >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  CdPlan L;
  int64_t blocks = 0;
  size_t lds = 0;
  const int var = cd_plan(a, L, blocks, lds);
  if (!var) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks);
  int rc = C3D_E_UNSUPPORTED;
  if (var == 1) rc = cd_launch<7, 2, 4, 3, 2, 1, 1>(a, L, grid, lds, s);        // 216 -> 96: 14 x 6 weight-gradient tiles, 4 x 3 per wave
  else if (var == 2) rc = cd_launch<4, 3, 1, 3, 4, 2, 4>(a, L, grid, lds, s);   // 108 -> 48: 7 x 3, 1 x 3 per wave
  else if (var == 3) rc = cd_launch<2, 2, 1, 1, 2, 1, 4>(a, L, grid, lds, s);   // 54 -> 24: 4 x 2, 1 x 1 per wave
  if (rc != 0) return rc;
  C3D_CHECK_LAUNCH();
  if (c3d_cdg_defer_reduce) { c3d_cdg_parts = (int)blocks; return 0; }   // (the stage driver reduces on its side stream)
  return c3d_detail_pw_wgrad_reduce(a.wg_ws, a.wg_dw, a.K, a.N, (int)blocks, a.w_sk, a.w_sn, s);
}

// conv_c: host-side check for the stage driver, and the launch
__attribute__((visibility("hidden"))) bool c3d_detail_pw_cdg_c_supported(int Kp, int Np, int64_t M, int64_t rows_per_sample) {
  if (Kp <= 0 || Np <= 0 || (Kp & 7) || (Np & 7)) return false;
  if (M < 1024 || (M + 512) * (int64_t)(Kp > Np ? Kp : Np) * 2 >= ((int64_t)1 << 31)) return false;
  c3d_pw_args a;
  std::memset(&a, 0, sizeof(a));
  a.M = M; a.K = a.Kp = Kp; a.N = a.Np = Np; a.rows_per_sample = rows_per_sample;
  CcPlan L;
  int64_t blocks = 0;
  size_t lds = 0;
  return cc_plan(a, L, blocks, lds) != 0;
}

__attribute__((visibility("hidden"))) int c3d_detail_pw_cdg_c(const c3d_pw_args* args, void* stream) {
  const c3d_pw_args& a = *args;
  if (a.dtype != C3D_DT_BF16 || a.row_mode != C3D_ROWS_DENSE || a.pro_mode != C3D_PRO_AFFINE2 || a.epi_mode != C3D_EPI_SWISH_SE_BWD ||
      a.wg_mode != C3D_WG_SWISH)
    return C3D_E_UNSUPPORTED;
  if (!a.w_img || !a.x2 || !a.e1 || !a.epi_p || !a.epi_q || !a.stats || !a.wg_dw || !a.wg_ws || a.pro_out || a.bias || a.fin.ticket ||
      a.add_sums)
    return C3D_E_UNSUPPORTED;
  if (a.fin.sums ? (a.fin.training != 0 || !a.fin.mr || !a.fin.gamma) : !a.pro_p) return C3D_E_UNSUPPORTED;
  if (a.M < 1024 || (a.M + 512) * (int64_t)(a.Kp > a.Np ? a.Kp : a.Np) * 2 >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  CcPlan L;
  int64_t blocks = 0;
  size_t lds = 0;
  const int var = cc_plan(a, L, blocks, lds);
  if (!var) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks);
  int rc = C3D_E_UNSUPPORTED;
  if (var == 1) rc = cc_launch<3, 3, 4, 1>(a, L, grid, lds, s);        // 96 -> 216: 6 x 14 weight-gradient tiles, 3 x 4 per wave
  else if (var == 2) rc = cc_launch<2, 3, 1, 2>(a, L, grid, lds, s);   // 48 -> 108: 3 x 7, 3 x 1 per wave
  else if (var == 3) rc = cc_launch<1, 1, 1, 4>(a, L, grid, lds, s);   // 24 -> 54: 2 x 4, 1 x 1 per wave
  if (rc != 0) return rc;
  C3D_CHECK_LAUNCH();
  if (c3d_cdg_defer_reduce) { c3d_cdg_parts = (int)blocks; return 0; }
  return c3d_detail_pw_wgrad_reduce(a.wg_ws, a.wg_dw, a.K, a.N, (int)blocks, a.w_sk, a.w_sn, s);
}
