// Weight gradient of the pointwise convolutions, second kernel (round 6): bf16 storage, dense rows.
//
//   dW[n, k] += sum_m P(m, n) * Q(m, k)        (reference model/x3d.py:173-175, 203-216: convolution_backward's weight half)
//
// What the first kernel (pw_wgrad.hip) measured on the res4 layers (profiles/r05_pw_wgrad_phase_clock.txt): its eight waves are
// split by OPERAND -- four stage P, four stage Q -- so the four that recompute Swish on 216 channels run alone on their SIMDs at
// a single wave's issue rate (49 % of their time) while the other four wait at the tile barrier (47 %); a tile is 32 rows (one
// barrier, one exposed fragment round trip and 875 clocks of exec-masked load issue per 32 rows); 1.8-2.7 TB/s.
//
// This kernel:
//  * staging is FLAT: a dense 64..256-row tile of an operand is one contiguous span of 16-byte vectors, item i = vector i of
//    the span, thread t takes items t, t + 512, ...: every wave does the same share of P and of Q work (both waves of a SIMD
//    interleave their Swish arithmetic), the loads are whole 1 KB wave requests, bounds-checked buffer loads without an exec
//    mask (a lane without an item asks "nowhere": counted waits stay exact, csrc/pw_gemm_impl.h BufIO);
//  * the LDS tiles are ROW-major [row][channel] (one ds_write_b128 per item, no register transpose); the fragments with the
//    ROWS as contraction index come from gfx950's transposing read (ds_read_b64_tr_b16, two per 32-row k-step), as in the
//    weight gradient fused into c3d_pw_gemm;
//  * per-channel prologue parameters (AFFINE2 A|B|C rebuilt from the producer's sums, BatchNorm_b scale|shift, the SE gates of
//    the two samples a tile can touch) live in LDS;
//  * same per-wave TN x TK accumulator grid, per-workgroup partials and fixed-order reducer as the first kernel.
// Per-element operand arithmetic is the first kernel's (same fmaf association, same v_exp/v_rcp sigmoid); the products are
// summed in a different order inside a k-step, so results agree to f32 rounding, not bit for bit.
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "launch_hints.h"
#include "bn_fin.h"
#include <cstdlib>

int c3d_detail_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int parts, int sn, int sk, hipStream_t stream);   // pw_wgrad.hip
int c3d_detail_pw_wgrad_v2_flush(hipStream_t stream);

namespace {

constexpr int W2_THREADS = 512;
// 16-byte items per thread, operand and tile (prefetch registers): 4 for an operand on 4 channel tiles per wave (up to 224
// channels at 64 rows), 2 for the narrower side
constexpr int w2_rounds(int t) { return t >= 4 ? 4 : 2; }
constexpr int W2_MAX_PARTS = 512;  // = WGRAD_MAX_PARTS of pw_wgrad.hip (c3d_pw_wgrad_ws_floats)
constexpr uint32_t W2_OOB = 0x80000000u;

struct W2Plan {
  int MT, tiles_per_wg, WN, WK;
  int ldp, ldq;          // row strides (elements) of the P / Q tiles: 16 * odd, so that the 8 rows a half-wave of a transposing
                         // read touches (32 B each) fall on distinct bank octets
  int q_off, buf_bytes;  // byte offset of the Q tile inside a buffer; bytes per buffer
  int par_off;           // f32 parameters: A|B|C [Np] each, scale|shift [Kp] each, gates [ns][Kp]
  int dump_off;          // 8 KB: where a lane without an item writes
  int ns;                // samples a workgroup's rows can span (their SE gates are staged once, in the prologue)
};

// Partials of the PREVIOUS chained launch on this stream (c3d_pw_wgrad_args.chain): reduced in this launch's prologue instead
// of by a pw_wgrad_reduce_kernel launch of their own (the 49 reducers of a BCD step were 5.7 us each on the side queue, plus
// their launch boundaries, and waited behind whole-CU kernels for a CU)
struct W2Red {
  const float* ws;   // [parts][N][K], or NULL
  float* dw;
  int N, K, parts, sn, sk;
};

typedef short w2_s16x4_t __attribute__((ext_vector_type(4)));
typedef short w2_s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) w2_s16x4_t* w2_lds_s16x4_ptr_t;
typedef uint32_t w2_u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w2_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ uint4 w2_load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const w2_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void w2_cvt(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void w2_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// q = g * (f * sc + sh);  f = q * sigmoid(q), eight channels.  The same IEEE operations as sigmoid_t<bf16_t> (common.h: v_exp_f32
// of q x -log2(e), + 1, v_rcp_f32) -- bit-identical results -- with the five multiplies / adds issued as packed-f32 instructions
// (two channels each): at two waves per SIMD a wave's instruction COUNT is what the conversion pass costs.
#ifndef W2_PK
#define W2_PK 1
#endif
__device__ __forceinline__ void w2_swish8(float (&f)[8], const float (&sc)[8], const float (&sh)[8], const float (&g)[8]) {
#if W2_PK
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2_t q = f32x2_t{g[j], g[j + 1]} * __builtin_elementwise_fma(f32x2_t{f[j], f[j + 1]}, f32x2_t{sc[j], sc[j + 1]}, f32x2_t{sh[j], sh[j + 1]});
    const f32x2_t t = q * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2_t{1.0f, 1.0f};
    const f32x2_t r = q * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f[j] = r[0]; f[j + 1] = r[1];
  }
#else
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float qv = g[j] * fmaf(f[j], sc[j], sh[j]);
    f[j] = qv * sigmoid_t<bf16_t>(qv);
  }
#endif
}

template <bool HASP2, bool QSW, int TN, int TK, bool IL>
__global__ __launch_bounds__(W2_THREADS) void pw_wgrad_v2_kernel(const c3d_pw_wgrad_args a, const W2Plan L, const W2Red red) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Np = a.Np, Kp = a.Kp, Gp = Np >> 3, Gq = Kp >> 3, MT = L.MT;
  const int ldp = L.ldp, ldq = L.ldq;
  float* const Pp = reinterpret_cast<float*>(smem + L.par_off);   // A | B | C
  float* const Qs = Pp + 3 * Np;                                  // scale | shift
  float* const Gs = Qs + 2 * Kp;                                  // gates [L.ns][Kp]

  const int M32 = (int)a.M;
  const int tiles = (M32 + MT - 1) / MT;
  int t0 = (int)blockIdx.x * L.tiles_per_wg;
  if (t0 > tiles) t0 = tiles;
  int t1 = t0 + L.tiles_per_wg;
  if (t1 > tiles) t1 = tiles;
  // buffer resources bounded by THIS workgroup's last row: the ragged end of the last tile reads zeros
  const uint32_t row_hi = (uint32_t)(t1 * MT < M32 ? t1 * MT : M32);
  const __amdgpu_buffer_rsrc_t rP = w2_rsrc(a.p, row_hi * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rP2 = w2_rsrc(HASP2 ? a.p2 : nullptr, row_hi * (uint32_t)Np * 2u);
  const __amdgpu_buffer_rsrc_t rQ = w2_rsrc(a.q, row_hi * (uint32_t)Kp * 2u);

  // ---- item maps: item i = tid + 512 r of a tile <-> (row = i / G, vector = i % G); its bytes sit at tile base + 16 i
  constexpr int RP = w2_rounds(TN), RQ = w2_rounds(TK);
  int p_desc[RP], q_desc[RQ];        // row << 5 | vector (0 for a lane without the item)
  uint32_t p_go[RP], q_go[RQ];       // byte offset inside the tile, or "nowhere"
  {
    const float invGp = 1.0f / (float)Gp, invGq = 1.0f / (float)Gq;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const int i = tid + W2_THREADS * r;
      const int rowp = __float2int_rz(((float)i + 0.5f) * invGp);
      const bool okp = i < MT * Gp;
      p_desc[r] = okp ? (rowp << 5) | (i - rowp * Gp) : 0;
      p_go[r] = okp ? (uint32_t)i * 16u : W2_OOB;
    }
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int i = tid + W2_THREADS * r;
      const int rowq = __float2int_rz(((float)i + 0.5f) * invGq);
      const bool okq = i < MT * Gq;
      q_desc[r] = okq ? (rowq << 5) | (i - rowq * Gq) : 0;
      q_go[r] = okq ? (uint32_t)i * 16u : W2_OOB;
    }
  }
  uint4 rawp[RP], rawq[RQ];
  uint4 rawp2[HASP2 ? RP : 1];
  const uint32_t tbp = (uint32_t)(MT * Np * 2), tbq = (uint32_t)(MT * Kp * 2);
#define W2_ISSUE(TILE)                                                                   \
  {                                                                                      \
    const uint32_t bp_ = (uint32_t)(TILE) * tbp, bq_ = (uint32_t)(TILE) * tbq;           \
    _Pragma("unroll") for (int r = 0; r < RP; ++r) {                                     \
      rawp[r] = w2_load(rP, p_go[r] + bp_);                                              \
      if (HASP2) rawp2[HASP2 ? r : 0] = w2_load(rP2, p_go[r] + bp_);                     \
    }                                                                                    \
    _Pragma("unroll") for (int r = 0; r < RQ; ++r) rawq[r] = w2_load(rQ, q_go[r] + bq_);   \
  }

  // the first tile's rows are requested before anything else: the zero fill and the parameter round trips run under them
  if (t0 < t1) W2_ISSUE(t0)
  // both tile buffers zeroed once: channel padding (up to the per-wave tile grid) is never written again
  for (int i = tid * 16; i < 2 * L.buf_bytes; i += W2_THREADS * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0u, 0u, 0u, 0u);

  // ---- the previous chained launch's partials -> its dw: pw_wgrad_reduce_kernel's arithmetic (32 outputs x 8 part-groups per
  // 256 threads, four accumulators per thread over parts p, p + 8, p + 16, p + 24 (+ 32 ..), combined as (s0 + s1) + (s2 + s3),
  // the eight groups added in order: bit-identical dw), two of its blocks per pass of this workgroup, every load of a pass
  // issued before the first add; under the first tile's memory latency
  if (red.ws) {
    float* const rsm = reinterpret_cast<float*>(smem + L.dump_off);   // [2][8][32] (the dump region is not in use yet)
    const int half = tid >> 8, t = tid & 255, e = t & 31, pg = t >> 5;
    const int NK = red.N * red.K, nvb = (NK + 31) >> 5;
    constexpr int RL = 8;   // parts per accumulator a thread covers without a second pass: 8 groups x 4 x RL = 256 partials
    for (int vb0 = (int)blockIdx.x * 2; vb0 < nvb; vb0 += (int)gridDim.x * 2) {
      const int idx = (vb0 + half) * 32 + e;
      const bool live = vb0 + half < nvb && idx < NK;
      float v[4][RL];
#pragma unroll
      for (int u = 0; u < RL; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p_ = pg + 8 * k + 32 * u;
          v[k][u] = (live && p_ < red.parts) ? red.ws[(size_t)p_ * NK + idx] : 0.f;
        }
      // the reducer's loop: full groups of four first (s0..s3 advance together while p + 24 < parts), then s0 takes the rest
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int p_ = pg, u_ = 0;
#pragma unroll
      for (int u = 0; u < RL; ++u) {
        if (p_ + 24 < red.parts) { s0 += v[0][u]; s1 += v[1][u]; s2 += v[2][u]; s3 += v[3][u]; p_ += 32; u_ = u + 1; }
      }
#pragma unroll
      for (int u = 0; u < RL; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q_ = pg + 8 * k + 32 * u;
          if (u >= u_ && q_ >= p_ && q_ < red.parts) s0 += v[k][u];
        }
      rsm[(half * 8 + pg) * 32 + e] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (pg == 0 && live) {
        float sum = 0.f;
#pragma unroll
        for (int g_ = 0; g_ < 8; ++g_) sum += rsm[(half * 8 + g_) * 32 + e];
        const int n = idx / red.K, k = idx - n * red.K;
        red.dw[(size_t)n * red.sn + (size_t)k * red.sk] += sum;
      }
      __syncthreads();
    }
  }

  // ---- per-sample SE gates: the gates of every sample this workgroup's rows touch are staged once (the planner bounded their
  // number); a tile touches at most two of them (rows_per_sample >= MT)
  const bool gate_on = QSW && a.q_gate != nullptr;
  const uint32_t rps = a.rows_per_sample > 0 ? (uint32_t)a.rows_per_sample : 1u;
  const int nmax = (int)((uint32_t)(M32 - 1) / rps);
  const int n_first = (int)((uint32_t)(t0 * MT) / rps);
  // ---- prologue parameters -> LDS (after the zero fill: separate regions)
  if constexpr (HASP2) {
    for (int c = tid; c < Np; c += W2_THREADS) {
      float cA, cB, cC;
      if (a.p_fin.sums) {
        c3dfin::bn_bwd_coef_consume(a.p_fin, a.N, Np, c, false, cA, cB, cC);
      } else {
        cA = a.p_coef[c]; cB = a.p_coef[Np + c]; cC = a.p_coef[2 * Np + c];
      }
      Pp[c] = cA; Pp[Np + c] = cB; Pp[2 * Np + c] = cC;
    }
  }
  if constexpr (QSW) {
    for (int c = tid; c < 2 * Kp; c += W2_THREADS) Qs[c] = a.q_ss[c];
    if (gate_on) {
      for (int i = tid; i < L.ns * Kp; i += W2_THREADS) {
        const int s_ = i / Kp, c = i - s_ * Kp;
        const int n = n_first + s_ < nmax ? n_first + s_ : nmax;
        Gs[i] = a.q_gate[(size_t)n * Kp + c];
      }
    } else {
      for (int i = tid; i < Kp; i += W2_THREADS) Gs[i] = 1.f;   // a block without SqueezeExcitation: 1.0f x q is q, bit for bit
    }
  }
  const int wn_i = wave % L.WN, wk_i = wave / L.WN;
  f32x4_t acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // transposing read: lane l addresses the 8-byte chunk (row 4 (l / 16) + (l % 16) / 4, channels 4 (l % 4) ..) of a 16 x 16
  // block and receives rows 4 (l / 16) .. + 3 of channel l % 16
  const int g4 = lane >> 4, li = lane & 15;
  const int pl = (4 * g4 + (li >> 2)) * ldp + 4 * (li & 3) + wn_i * 16;
  const int ql = (4 * g4 + (li >> 2)) * ldq + 4 * (li & 3) + wk_i * 16;
  const int pstep = L.WN * 16, qstep = L.WK * 16;
  const int KS = MT >> 5;

  // ---- one tile: convert + prologue -> row-major LDS tiles.  Branch-free: an item a lane does not have is converted all the
  // same (its request answered zeros) and written to the lane's 16 bytes of a dump region, so that the pass is ONE basic block
  // the scheduler can interleave with the multiply of the previous tile.  The slot a round has just converted is requested
  // again at once for the next tile: every request then has a whole iteration to land.  (Requested in one burst after the
  // pass, as the first kernel does, the rows had only the barrier and the multiply to land in.)
  bf16_t* const dump = reinterpret_cast<bf16_t*>(smem + L.dump_off) + tid * 8;
  // The last round of an operand is the only one that can be empty for a whole wave (items are dealt 512 at a time): it is
  // skipped under a wave-uniform branch -- e.g. 64 rows x 27 vectors = 3.4 rounds: waves 3..7 convert three, not four
  const bool p_last = (wave * 64 + W2_THREADS * (RP - 1)) < MT * Gp;
  const bool q_last = (wave * 64 + W2_THREADS * (RQ - 1)) < MT * Gq;
#define W2_CONVERT(TILE, CP, CQ)                                                                                   \
  {                                                                                                                 \
    const int rowg0_ = (TILE) * MT;                                                                                 \
    const int n_lo_ = (int)((uint32_t)rowg0_ / rps);                                                                \
    const int bound_ = (n_lo_ + 1) * (int)rps;                /* first row of the tile's second sample */          \
    const float* const GsC_ = Gs + (gate_on ? n_lo_ - n_first : 0) * Kp;                                            \
    const uint32_t bpn_ = (uint32_t)((TILE) + 1) * tbp, bqn_ = (uint32_t)((TILE) + 1) * tbq;                        \
    _Pragma("unroll") for (int r = 0; r < RP; ++r) {                                                                \
      if (r == RP - 1 && !p_last) continue;   /* (wave-uniform: none of this wave's lanes has an item in the last round) */ \
      const int row = p_desc[r] >> 5, v = p_desc[r] & 31;                                                           \
      bf16_t* dst = p_go[r] != W2_OOB ? (CP) + row * ldp + v * 8 : dump;                                            \
      if constexpr (HASP2) {                                                                                        \
        float f[8], f2[8], cA[8], cB[8], cC[8];                                                                     \
        w2_cvt(rawp[r], f);                                                                                         \
        w2_cvt(rawp2[HASP2 ? r : 0], f2);                                                                           \
        w2_ld8(Pp + v * 8, cA); w2_ld8(Pp + Np + v * 8, cB); w2_ld8(Pp + 2 * Np + v * 8, cC);                       \
        /* a row past the tensor's end is zero x A + B: zeroed by a bit mask (as `real ? fma : 0` the compiler made it a branch) */ \
        const uint32_t keep = rowg0_ + row < M32 ? 0xffffffffu : 0u;                                                \
        _Pragma("unroll") for (int j = 0; j < 8; j += 2) {   /* (packed f32: the same two fused multiply-adds per channel) */ \
          const f32x2_t in_ = __builtin_elementwise_fma(f32x2_t{cC[j], cC[j + 1]}, f32x2_t{f2[j], f2[j + 1]}, f32x2_t{cB[j], cB[j + 1]}); \
          const f32x2_t o_ = __builtin_elementwise_fma(f32x2_t{cA[j], cA[j + 1]}, f32x2_t{f[j], f[j + 1]}, in_);     \
          f[j] = __uint_as_float(__float_as_uint(o_[0]) & keep); f[j + 1] = __uint_as_float(__float_as_uint(o_[1]) & keep); \
        }                                                                                                           \
        Vec8<bf16_t>::store(dst, f);                                                                                \
      } else {                                                                                                      \
        *reinterpret_cast<uint4*>(dst) = rawp[r];   /* no prologue: the raw bf16 vector IS the operand */            \
      }                                                                                                             \
      rawp[r] = w2_load(rP, p_go[r] + bpn_);   /* (past this workgroup's last row: zeros, no memory access) */       \
      if (HASP2) rawp2[HASP2 ? r : 0] = w2_load(rP2, p_go[r] + bpn_);                                               \
    }                                                                                                               \
    _Pragma("unroll") for (int r = 0; r < RQ; ++r) {                                                                \
      if (r == RQ - 1 && !q_last) continue;                                                                         \
      const int row = q_desc[r] >> 5, v = q_desc[r] & 31;                                                           \
      bf16_t* dst = q_go[r] != W2_OOB ? (CQ) + row * ldq + v * 8 : dump;                                            \
      if constexpr (QSW) {                                                                                          \
        float f[8], sc[8], sh[8], g[8];                                                                             \
        w2_cvt(rawq[r], f);                                                                                         \
        w2_ld8(Qs + v * 8, sc); w2_ld8(Qs + Kp + v * 8, sh);                                                        \
        w2_ld8(GsC_ + ((gate_on && rowg0_ + row >= bound_) ? Kp : 0) + v * 8, g);   /* (no gate: a row of ones) */    \
        w2_swish8(f, sc, sh, g);                                                                                    \
        Vec8<bf16_t>::store(dst, f);                                                                                \
      } else {                                                                                                      \
        *reinterpret_cast<uint4*>(dst) = rawq[r];                                                                   \
      }                                                                                                             \
      rawq[r] = w2_load(rQ, q_go[r] + bqn_);                                                                        \
    }                                                                                                               \
  }
  // ---- multiply one tile: rows are the contraction index
#define W2_FRAGS(BP, BQ, SLOT, KSI)                                                                                 \
  {                                                                                                                 \
    const bf16_t* pb_ = (BP) + (KSI) * 32 * ldp + pl;                                                               \
    const bf16_t* qb_ = (BQ) + (KSI) * 32 * ldq + ql;                                                               \
    _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                                \
      const w2_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4_ptr_t)(pb_ + i * pstep));        \
      const w2_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4_ptr_t)(pb_ + 16 * ldp + i * pstep)); \
      pa[SLOT][i] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));           \
    }                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < TK; ++j) {                                                                \
      const w2_s16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4_ptr_t)(qb_ + j * qstep));        \
      const w2_s16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4_ptr_t)(qb_ + 16 * ldq + j * qstep)); \
      qb[SLOT][j] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));           \
    }                                                                                                               \
  }
#define W2_MMA(SLOT)                                                                                                \
  {                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                                  \
      _Pragma("unroll") for (int j = 0; j < TK; ++j)                                                                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[SLOT][i]),              \
                                                            __builtin_bit_cast(bf16x8_t, qb[SLOT][j]), acc[i][j], 0, 0, 0); \
  }

  __syncthreads();   // zero fill, parameters, gates
  bf16_t* const buf0 = reinterpret_cast<bf16_t*>(smem);
  bf16_t* const buf1 = reinterpret_cast<bf16_t*>(smem + L.buf_bytes);
  const int qoe = L.q_off >> 1;   // Q tile offset in elements
  if (t0 < t1) W2_CONVERT(t0, buf0, buf0 + qoe)
  int cur = 0;
  for (int tile = t0; tile < t1; ++tile, cur ^= 1) {
    bf16_t* const bufP = cur ? buf1 : buf0;
    bf16_t* const bufQ = bufP + qoe;
    bf16_t* const nxtP = cur ? buf0 : buf1;
    bf16_t* const nxtQ = nxtP + qoe;
    __syncthreads();     // the only barrier per tile: tile `tile` is complete in this buffer, and every wave is past the multiply
                         // that read the other one
    uint4 pa[2][TN], qb[2][TK];
    if constexpr (IL) {
      // MT = 64: two k-steps, unrolled -- the multiply of this tile and the conversion of the next one are ONE basic block:
      // matrix-core and LDS-read work of the one beside VALU work of the other, in every wave
      if (tile + 1 < t1) {
        W2_FRAGS(bufP, bufQ, 0, 0)
        W2_FRAGS(bufP, bufQ, 1, 1)
        W2_MMA(0)
        W2_CONVERT(tile + 1, nxtP, nxtQ)
        W2_MMA(1)
        // (left to itself the scheduler keeps the source order -- 12 matrix-core instructions, the whole conversion, 12 more:
        // both waves of a SIMD then queue on the matrix core and on the VALU in turn)
#ifndef W2_VPM_Q
#define W2_VPM_Q 18
#endif
#ifndef W2_VPM_P
#define W2_VPM_P 11
#endif
#ifndef W2_SCHED
#define W2_SCHED 1
#endif
        constexpr int VPM = QSW ? W2_VPM_Q : (HASP2 ? W2_VPM_P : 2);
#pragma unroll
        for (int i = 0; i < (W2_SCHED ? 2 * TN * TK : 0); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VPM VALU instructions of the conversion
          if (i & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a re-request, as soon as its slot is free (else: all at the end)
        }
      } else {
        W2_FRAGS(bufP, bufQ, 0, 0)
        W2_FRAGS(bufP, bufQ, 1, 1)
        W2_MMA(0)
        W2_MMA(1)
      }
    } else {
      W2_FRAGS(bufP, bufQ, 0, 0)
      for (int ks = 0; ks < KS; ks += 2) {   // KS is even (MT is a multiple of 64)
        W2_FRAGS(bufP, bufQ, 1, ks + 1)
        W2_MMA(0)
        if (ks + 2 < KS) W2_FRAGS(bufP, bufQ, 0, ks + 2)
        W2_MMA(1)
      }
      if (tile + 1 < t1) W2_CONVERT(tile + 1, nxtP, nxtQ)
    }
  }
#undef W2_FRAGS
#undef W2_MMA
#undef W2_CONVERT
#undef W2_ISSUE

  // partials -> workspace [grid][N][K] (the reducer behind this launch adds them in fixed order)
  float* wsb = a.ws + (size_t)blockIdx.x * a.N * a.K;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int nt = wn_i + i * L.WN;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int kt = wk_i + j * L.WK;
      const int k = kt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + (lane >> 4) * 4 + r;
        if (n < a.N && k < a.K) wsb[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  }
}

template <bool HASP2, bool QSW, int TN, int TK, bool IL>
int w2_launch_il(const c3d_pw_wgrad_args& a, const W2Plan& L, const W2Red& red, dim3 grid, size_t lds, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_v2_kernel<HASP2, QSW, TN, TK, IL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  pw_wgrad_v2_kernel<HASP2, QSW, TN, TK, IL><<<grid, dim3(W2_THREADS), lds, stream>>>(a, L, red);
  return 0;
}
template <bool HASP2, bool QSW, int TN, int TK>
int w2_launch_inst(const c3d_pw_wgrad_args& a, const W2Plan& L, const W2Red& red, dim3 grid, size_t lds, hipStream_t stream) {
  return L.MT == 64 ? w2_launch_il<HASP2, QSW, TN, TK, true>(a, L, red, grid, lds, stream)
                    : w2_launch_il<HASP2, QSW, TN, TK, false>(a, L, red, grid, lds, stream);
}

struct W2Inst { int tn, tk; };
constexpr W2Inst W2_INSTS[] = {{1, 1}, {2, 2}, {3, 4}, {4, 3}, {4, 4}};

template <bool HASP2, bool QSW>
int w2_launch_pick(const c3d_pw_wgrad_args& a, int inst, const W2Plan& L, const W2Red& red, dim3 grid, size_t lds, hipStream_t s) {
  switch (inst) {
    case 0: return w2_launch_inst<HASP2, QSW, 1, 1>(a, L, red, grid, lds, s);
    case 1: return w2_launch_inst<HASP2, QSW, 2, 2>(a, L, red, grid, lds, s);
    case 2: return w2_launch_inst<HASP2, QSW, 3, 4>(a, L, red, grid, lds, s);
    case 3: return w2_launch_inst<HASP2, QSW, 4, 3>(a, L, red, grid, lds, s);
    default: return w2_launch_inst<HASP2, QSW, 4, 4>(a, L, red, grid, lds, s);
  }
}

// the partials of this thread's last chained launch that no launch has reduced yet
struct W2Pending { W2Red red; hipStream_t stream; };
thread_local W2Pending w2_pending = {{nullptr, nullptr, 0, 0, 0, 0, 0}, nullptr};

}  // namespace

// Reducer launch for the pending partials (c3d_pw_wgrad_flush, and whenever the next launch cannot take them over).
__attribute__((visibility("hidden"))) int c3d_detail_pw_wgrad_v2_flush(hipStream_t stream) {
  if (!w2_pending.red.ws) return 0;
  const W2Red r = w2_pending.red;
  w2_pending.red.ws = nullptr;
  (void)stream;   // the partials' own stream orders the reducer behind the launch that wrote them
  return c3d_detail_pw_wgrad_reduce(r.ws, r.dw, r.N, r.K, r.parts, r.sn, r.sk, w2_pending.stream);
}

void c3d_detail_pw_wgrad_v2_drop() { w2_pending.red.ws = nullptr; }

// Returns C3D_E_UNSUPPORTED for what it does not take (the caller then runs the first kernel).
__attribute__((visibility("hidden"))) int c3d_detail_pw_wgrad_v2(const c3d_pw_wgrad_args* args, hipStream_t stream) {
  const c3d_pw_wgrad_args& a = *args;
  if (a.dtype != C3D_DT_BF16 || a.row_mode != C3D_ROWS_DENSE || a.taps > 1) return C3D_E_UNSUPPORTED;
  if (a.Kp > 224 || a.Np > 224 || a.M <= 0) return C3D_E_UNSUPPORTED;
  const bool hasp2 = a.p_coef || a.p_fin.sums;
  const bool qsw = a.q_mode == C3D_PRO_BN_SE_SWISH;
  if (qsw && !hasp2) return C3D_E_UNSUPPORTED;   // (no layer has it)
  // 32-bit byte offsets into bounds-checked resources
  // (one tile past the end is addressed, and "nowhere" = 2^31 + a tile base must not wrap into the tensor)
  if ((a.M + 512) * (int64_t)(a.Np > a.Kp ? a.Np : a.Kp) * 2 >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  const int NT = (a.Np + 15) >> 4, KT = (a.Kp + 15) >> 4;
  // wave grid WN x WK = 8 and the instantiated per-wave tile grid, as in pw_wgrad.hip
  int WN = 0, WK = 0, tn_need = 0, tk_need = 0;
  const int cand[4][2] = {{8, 1}, {4, 2}, {2, 4}, {1, 8}};
  int best = 1 << 30;
  for (int c = 0; c < 4; ++c) {
    const int tn = (NT + cand[c][0] - 1) / cand[c][0], tk = (KT + cand[c][1] - 1) / cand[c][1];
    if (tn > 4 || tk > 4) continue;
    int ti = 4, tj = 4;
    for (int i = 0; i < 5; ++i)
      if (W2_INSTS[i].tn >= tn && W2_INSTS[i].tk >= tk) { ti = W2_INSTS[i].tn; tj = W2_INSTS[i].tk; break; }
    const int cost = ti * tj * 4 + ti + tj;
    if (cost < best) { best = cost; WN = cand[c][0]; WK = cand[c][1]; tn_need = tn; tk_need = tk; }
  }
  if (WN == 0) return C3D_E_UNSUPPORTED;
  int inst = 4;
  for (int i = 0; i < 5; ++i)
    if (W2_INSTS[i].tn >= tn_need && W2_INSTS[i].tk >= tk_need) { inst = i; break; }
  const int TNi = W2_INSTS[inst].tn, TKi = W2_INSTS[inst].tk;
  if (inst == 4 && hasp2) return C3D_E_UNSUPPORTED;   // 4 x 4 tiles per wave beside the two-tensor prefetch: scratch (no layer has it)
  W2Plan L;
  L.WN = WN; L.WK = WK;
  L.ldp = TNi * WN * 16; if (((L.ldp >> 4) & 1) == 0) L.ldp += 16;
  L.ldq = TKi * WK * 16; if (((L.ldq >> 4) & 1) == 0) L.ldq += 16;
  const int Gp = a.Np >> 3, Gq = a.Kp >> 3;
  const bool gate = qsw && a.q_gate;
  int64_t cap = device_cus() < W2_MAX_PARTS ? device_cus() : W2_MAX_PARTS;
  static const int cap_env = c3d_env("C3D_WG_BLOCKS") ? atoi(c3d_env("C3D_WG_BLOCKS")) : 0;   // tuning knobs of pw_wgrad.hip
  static const int side_env = c3d_env("C3D_PWWG_SIDE_WGS") ? atoi(c3d_env("C3D_PWWG_SIDE_WGS")) : 0;
  static const int mt_env = c3d_env("C3D_WG2_MT") ? atoi(c3d_env("C3D_WG2_MT")) : 0;
  if (cap_env > 0 && cap_env <= W2_MAX_PARTS) cap = cap_env;
  else if (c3d_side_launch) {   // beside the data-gradient chain: 7/8 of the CUs (launch_hints.h)
#ifndef W2_SIDE_EIGHTHS
#define W2_SIDE_EIGHTHS 7
#endif
    const int64_t side_cap = side_env > 0 ? side_env : (int64_t)device_cus() * W2_SIDE_EIGHTHS / 8;
    if (side_cap < cap) cap = side_cap;
  }
  // rows per tile: the tallest of 256 / 128 / 64 that fits the item budget, LDS and the two-samples-per-tile rule, and still
  // leaves every workgroup a walk of a few tiles
  constexpr int W2_NS_MAX = 16;
  int MT = 0, ns = 0;
  size_t lds = 0, par_bytes = 0;
  for (int mt = 256; mt >= 64; mt >>= 1) {
    if (mt_env && mt > mt_env) continue;
    // the workgroup grid this tile height gives, and the samples one workgroup's rows can then span
    {
      const int64_t tiles_ = (a.M + mt - 1) / mt;
      int64_t blocks_ = (tiles_ + 1) / 2;
      if (blocks_ > cap) blocks_ = cap;
      if (blocks_ < 1) blocks_ = 1;
      const int64_t rows_wg = (tiles_ + blocks_ - 1) / blocks_ * mt;
      ns = gate ? (int)((rows_wg + a.rows_per_sample - 2) / a.rows_per_sample + 1) : 1;
      if (ns > W2_NS_MAX) continue;
    }
    par_bytes = (size_t)(3 * a.Np + (2 + ns) * a.Kp) * sizeof(float) + W2_THREADS * 16;   // + the dump region
    if ((mt * Gp + W2_THREADS - 1) / W2_THREADS > w2_rounds(TNi) || (mt * Gq + W2_THREADS - 1) / W2_THREADS > w2_rounds(TKi)) continue;
    if (gate && a.rows_per_sample < mt) continue;
    const size_t buf = (size_t)mt * (L.ldp + L.ldq) * 2;
    if (2 * buf + par_bytes > 160 * 1024) continue;
    if (mt > 64 && a.M / mt < 3 * cap) continue;
    MT = mt; lds = 2 * buf + par_bytes;
    break;
  }
  if (MT == 0) return C3D_E_UNSUPPORTED;
  L.MT = MT;
  L.q_off = MT * L.ldp * 2;
  L.buf_bytes = MT * (L.ldp + L.ldq) * 2;
  L.par_off = 2 * L.buf_bytes;
  L.ns = ns;
  L.dump_off = L.par_off + (3 * a.Np + (2 + ns) * a.Kp) * (int)sizeof(float);
  const int64_t tiles = (a.M + MT - 1) / MT;
  int64_t blocks = (tiles + 1) / 2;   // >= 2 tiles per workgroup when there is enough work
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  L.tiles_per_wg = tpw;
  const dim3 grid((unsigned)blocks);
  // pending partials of the previous chained launch: this launch's prologue reduces them -- if they are on this stream, in
  // another workspace, and small enough for one register pass (256 partials); anything else gets its own reducer launch now
  W2Red red = {nullptr, nullptr, 0, 0, 0, 0, 0};
  if (w2_pending.red.ws) {
    if (w2_pending.stream == stream && w2_pending.red.ws != a.ws && w2_pending.red.parts <= 256) red = w2_pending.red;
    else { const int rcf = c3d_detail_pw_wgrad_v2_flush(w2_pending.stream); if (rcf != 0) return rcf; }
    w2_pending.red.ws = nullptr;
  }
  int rc;
  if (hasp2 && qsw) rc = w2_launch_pick<true, true>(a, inst, L, red, grid, lds, stream);
  else if (hasp2) rc = w2_launch_pick<true, false>(a, inst, L, red, grid, lds, stream);
  else rc = w2_launch_pick<false, false>(a, inst, L, red, grid, lds, stream);
  if (rc != 0) return rc;
  C3D_CHECK_LAUNCH();
  if (a.chain) {
    w2_pending.red = W2Red{a.ws, a.dw, a.N, a.K, (int)blocks, a.dw_sn, a.dw_sk};
    w2_pending.stream = stream;
    return 0;
  }
  return c3d_detail_pw_wgrad_reduce(a.ws, a.dw, a.N, a.K, (int)blocks, a.dw_sn, a.dw_sk, stream);
}
