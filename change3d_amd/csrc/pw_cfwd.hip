// conv_c FORWARD of the X3D bottleneck on a workgroup-cooperative tile loop (round 6): bf16 storage, dense rows,
//
//     c[m, n] = sum_k swish(gate[sample(m), k] * (b[m, k] * scale[k] + shift[k])) * W_c[n, k],   + BatchNorm_c statistics
//
// (reference model/x3d.py:203-216: norm_b -> SqueezeExcitation -> Swish -> conv_c -> norm_c).  Same arguments as the
// C3D_PRO_BN_SE_SWISH / C3D_EPI_STATS form of c3d_pw_gemm, which dispatches here (C3D_OPT_PW_CFWD) and keeps everything this
// kernel does not take.
//
// Why a second kernel.  In csrc/pw_gemm_impl.h every WAVE owns whole 16-row tiles end to end: request -> BatchNorm x SE x
// Swish on 216 channels (13 VALU instructions per element) -> 42 MFMAs behind 42 weight-fragment reads -> store + statistics,
// one phase after the other, three tiles per wave on the 32 x 32 maps; the round-6 counters show its waves parked 59 % of
// their cycles and the launch at 1.5 TB/s -- the slowest large shape of the step for three rounds.  The weight-gradient kernel
// of this round (csrc/pw_wgrad_v2.hip) showed what helps at two waves per SIMD: deal the conversion to all waves alike, and
// put the matrix-core phase of tile t into ONE basic block with the VALU phase of tile t + 1.  Here:
//  * a workgroup walks 64- or 128-row tiles; the 16-byte vectors of a tile are dealt 512 at a time (flat staging, bounds-
//    checked buffer loads, each slot requested again for the next tile as soon as it is converted);
//  * the converted rows go to a double-buffered row-major LDS tile; wave (r, c) multiplies row slab r (16 rows) by the
//    output-tile group c: the weight image is the packed LDS image of the first kernel (c3d_pw_pack_weights), copied once;
//  * multiply + store + statistics of tile t and the conversion of tile t + 1 are one basic block (two k-step counts
//    instantiated);
//  * BatchNorm_b scale / shift and the SE gate are rebuilt in the prologue from the per-sample sums by the SAME device
//    functions as the first kernel (csrc/bn_fin.h), under the first tile's requests.
// Results: the converted operand, the weight fragments and the k order of the MFMA chain are the first kernel's, so `c` is
// bit-identical to it; the statistics group their f32 partial sums differently (agreement to f32 rounding).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "launch_hints.h"
#include "bn_fin.h"
#include <cstdlib>

#ifdef C3D_CF_CLOCK
// Debug build only (tools/r6/cfwd_clock.py): s_memtime stamps of wave 0 per workgroup -- [0] entry, [6] requests issued, [7]
// BatchNorm_b scale / shift done, [5] SE gate done, [1] weights and first tile's rows in, [2] first tile converted, [3] tile
// loop done, [4] statistics flushed
__device__ unsigned long long c3d_cf_clk[1024][8];
#define CFCLK(i) { if (threadIdx.x == 0) c3d_cf_clk[blockIdx.x & 1023][i] = __builtin_amdgcn_s_memtime(); }
extern "C" int c3d_debug_cf_clock(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_cf_clk), sizeof(c3d_cf_clk));
}
#else
#define CFCLK(i)
#endif

namespace {

constexpr int CF_THREADS = 512;
constexpr int CF_SE_NS = 4;          // samples a workgroup's rows may span with the in-kernel SE gate (= PW_SE_NS)
constexpr int CF_SE_CR = 32;         // hidden units at most (= PW_SE_CR)
constexpr uint32_t CF_OOB = 0x80000000u;

struct CfPlan {
  int MT, WR, WC;            // rows per tile = 16 WR; wave grid WR x WC = 8
  int tiles_per_wg;
  int KL;                    // row stride (elements) of the converted tile = Kpad + 8
  int img_rows;              // rows of the packed weight image per 8-element k-chunk
  int w_off, a_off, a_bytes; // weight image | two converted tiles
  int os_off, os_wave;       // per-wave result staging [16][NLw]
  int par_off;               // scale | shift [Kp] each, gates [ns][Kp] (+ hidden units), dump
  int gate_off, dump_off;
  int ns;
};

typedef uint32_t cf_u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* cf_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* cf_glb_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cf_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ uint4 cf_load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const cf_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void cf_cvt(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void cf_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// PRO: C3D_PRO_BN_SE_SWISH (conv_c) or C3D_PRO_AFFINE2 with the fused residual output (conv_a: the operand is
// y = relu(bn_c(c) + shortcut) of the previous block, written out once -- c3d_pw_args.pro_out); NTW: output tiles (16 channels)
// per wave; KS: k-steps of 32 (Kpad / 32); STATS: BatchNorm statistics epilogue
template <int PRO, int NTW, int KS, bool STATS>
__global__ __launch_bounds__(CF_THREADS) void pw_cfwd_kernel(const c3d_pw_args a, const CfPlan L) {
  typedef Mma<bf16_t> MM;
  constexpr bool AFF = PRO == C3D_PRO_AFFINE2;
  // 16-byte items per thread and tile (prefetch registers).  conv_c: 128 rows x 7 vectors, 64 x 27, 128 x 14; conv_a: two tensors
  // of 64 x 12, 128 x 6, 128 x 3 vectors
  constexpr int CF_R = (AFF || KS <= 2) ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CFCLK(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kp = a.Kp, Np = a.Np, Gq = Kp >> 3, MT = L.MT, KL = L.KL;
  bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem + L.w_off);
  float* const Pp = reinterpret_cast<float*>(smem + L.par_off);       // scale | shift
  float* const Gs = reinterpret_cast<float*>(smem + L.gate_off);      // gates [ns][Kp] (+ [ns][32] hidden units behind them)
  bf16_t* const dump = reinterpret_cast<bf16_t*>(smem + L.dump_off) + tid * 8;

  const int M32 = (int)a.M;
  const int tiles = (M32 + MT - 1) / MT;
  int t0 = (int)blockIdx.x * L.tiles_per_wg;
  if (t0 > tiles) t0 = tiles;
  int t1 = t0 + L.tiles_per_wg;
  if (t1 > tiles) t1 = tiles;
  const uint32_t row_hi = (uint32_t)(t1 * MT < M32 ? t1 * MT : M32);
  const __amdgpu_buffer_rsrc_t rX = cf_rsrc(a.x, row_hi * (uint32_t)Kp * 2u);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rX2 = cf_rsrc(AFF ? a.x2 : nullptr, row_hi * (uint32_t)Kp * 2u);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rPO = cf_rsrc(AFF ? a.pro_out : nullptr, row_hi * (uint32_t)Kp * 2u);
  const __amdgpu_buffer_rsrc_t rY = cf_rsrc(a.y, (uint32_t)M32 * (uint32_t)Np * 2u);

  // ---- item map: item i = tid + 512 r of a tile <-> (row = i / Gq, vector = i % Gq); its bytes sit at tile base + 16 i
  int q_desc[CF_R];
  uint32_t q_go[CF_R];
  {
    const float invG = 1.0f / (float)Gq;
#pragma unroll
    for (int r = 0; r < CF_R; ++r) {
      const int i = tid + CF_THREADS * r;
      const int row = __float2int_rz(((float)i + 0.5f) * invG);
      const bool ok = i < MT * Gq;
      q_desc[r] = ok ? (row << 5) | (i - row * Gq) : 0;
      q_go[r] = ok ? (uint32_t)i * 16u : CF_OOB;
    }
  }
  const bool q_last = (wave * 64 + CF_THREADS * (CF_R - 1)) < MT * Gq;   // (wave-uniform: the last round has an item for this wave)
  uint4 raw[CF_R];
  uint4 raw2[AFF ? CF_R : 1];
  const uint32_t tbq = (uint32_t)(MT * Kp * 2);
  {
    const uint32_t b0 = (uint32_t)t0 * tbq;
#pragma unroll
    for (int r = 0; r < CF_R; ++r) {
      raw[r] = cf_load(rX, q_go[r] + b0);
      if (AFF) raw2[AFF ? r : 0] = cf_load(rX2, q_go[r] + b0);
    }
  }

  // ---- weight image -> LDS (LDS-DMA, 1 KB per wave instruction; the chunk order rotated by the workgroup index: every
  // workgroup of the launch reads the same image at the same moment)
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
        __builtin_amdgcn_global_load_lds((cf_glb_ptr_t)(src + off), (cf_lds_ptr_t)(smem + L.w_off + r * 1024), 16, 0, 0);
    }
  }
  const uint32_t rps = a.rows_per_sample > 0 ? (uint32_t)a.rows_per_sample : 1u;
  [[maybe_unused]] const int se_n_lo = (int)((uint32_t)(t0 * MT) / rps);
  [[maybe_unused]] const bool gate_on = !AFF && a.se_w1 != nullptr;
  // both converted tiles zeroed once: the k-padding columns [Kp, KL) are never written again
  for (int i = tid * 16; i < 2 * L.a_bytes; i += CF_THREADS * 16) *reinterpret_cast<uint4*>(smem + L.a_off + i) = make_uint4(0u, 0u, 0u, 0u);

  // ---- conv_c: BatchNorm_b scale / shift (and the SE gate of this workgroup's samples) from the per-sample sums; conv_a:
  // BatchNorm_c scale / shift of the PREVIOUS block from its completed statistics -- the first kernel's prologue
  // (csrc/pw_gemm_impl.h), same device functions, same owner rules
  CFCLK(6)
  if (blockIdx.x == 0 && tid == 0 && a.fin.nbt) *a.fin.nbt += 1;
  if constexpr (AFF) {
    c3dfin::bn_consume(a.fin, a.K, Kp, 0, Kp, blockIdx.x == 0, Pp, Pp + Kp, tid, CF_THREADS);
  } else {
    c3dfin::bn_consume_nc(a.fin, a.K, Kp, blockIdx.x == 0, Pp, Pp + Kp, tid, CF_THREADS);
    CFCLK(7)
    if (gate_on) {
      if (t0 < t1) {
        const int r0 = t0 * MT, r1 = (int)row_hi - 1;
        const int n_lo = (int)((uint32_t)r0 / rps), n_hi = (int)((uint32_t)r1 / rps);
        const int own_lo = (int)(((uint32_t)r0 + rps - 1) / rps);   // first sample whose row 0 is >= r0
        c3dfin::se_gate_consume(a.fin.sums, (double)a.rows_per_sample, a.K, Kp, a.se_w1, a.se_b1, a.se_w2, a.se_b2, a.se_cr, n_lo,
                                n_hi - n_lo + 1, own_lo, n_hi, Pp, Pp + Kp, Gs, Gs + L.ns * Kp, const_cast<float*>(a.pro_gate),
                                a.se_hid, tid, CF_THREADS);
      }
    } else {
      for (int i = tid; i < Kp; i += CF_THREADS) Gs[i] = 1.f;   // a block without SqueezeExcitation: 1.0f x q is q, bit for bit
    }
  }
  CFCLK(5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA chunks (and the first tile's rows)
  __syncthreads();
  CFCLK(1)

  // ---- lane maps of the multiply and of the epilogue
  const int wr = wave % L.WR, wc = wave / L.WR;
  const int nt0 = wc * NTW;                               // first output tile of this wave
  constexpr int NLW = NTW * 16 + 8;                       // row stride (elements) of the wave's result staging
  bf16_t* const Os = reinterpret_cast<bf16_t*>(smem + L.os_off + wave * L.os_wave);
  constexpr int GOW = NTW * 2, RPO = 64 / GOW, NPASS = (16 + RPO - 1) / RPO;
  const int rr_o = lane / GOW, v_o = lane - rr_o * GOW;
  const int cvec = nt0 * 2 + v_o;                         // 8-channel vector of the output row
  const bool act_o = lane < GOW * RPO && cvec * 8 < Np;
  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  const bf16_t* const wfrag = Ws + ((size_t)(lane >> 4) * L.img_rows + nt0 * 16 + (lane & 15)) * 8;   // + (ks * 4 * img_rows + nt * 16) * 8
  const int xfrag = (wr * 16 + (lane & 15)) * KL + (lane >> 4) * 8;                                  // + ks * 32

#define CF_CONVERT(TILE, CQ)                                                                                        \
  {                                                                                                                 \
    const int rowg0_ = (TILE) * MT;                                                                                 \
    [[maybe_unused]] const int n_lo_ = (int)((uint32_t)rowg0_ / rps);                                               \
    [[maybe_unused]] const int bound_ = (n_lo_ + 1) * (int)rps;   /* first row of the tile's second sample */       \
    [[maybe_unused]] const float* const GsC_ = Gs + (gate_on ? n_lo_ - se_n_lo : 0) * Kp;                           \
    const uint32_t bq_ = (uint32_t)(TILE) * tbq, bqn_ = bq_ + tbq;                                                  \
    _Pragma("unroll") for (int r = 0; r < CF_R; ++r) {                                                              \
      if (r == CF_R - 1 && !q_last) continue;                                                                       \
      const int row = q_desc[r] >> 5, v = q_desc[r] & 31;                                                           \
      bf16_t* dst = q_go[r] != CF_OOB ? (CQ) + row * KL + v * 8 : dump;                                             \
      float f[8], sc[8], sh[8];                                                                                     \
      cf_cvt(raw[r], f);                                                                                            \
      cf_ld8(Pp + v * 8, sc); cf_ld8(Pp + Kp + v * 8, sh);                                                          \
      if constexpr (AFF) {                                                                                          \
        /* y = relu(bn_c(c) + shortcut) in the association of c3d_block_out_fwd (fmaf(f2, 1, 0) is f2), zero past the  \
           tensor's end; the bf16 rounding of y that is stored is the operand the GEMM reads */                     \
        float f2[8];                                                                                                \
        cf_cvt(raw2[AFF ? r : 0], f2);                                                                              \
        const uint32_t keep = rowg0_ + row < M32 ? 0xffffffffu : 0u;                                                \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                               \
          f[e] = __uint_as_float(__float_as_uint(fmaxf(fmaf(f[e], sc[e], sh[e]) + fmaf(f2[e], 1.f, 0.f), 0.f)) & keep); \
        const uint4 pk = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])); \
        *reinterpret_cast<uint4*>(dst) = pk;                                                                        \
        __builtin_amdgcn_raw_buffer_store_b128(cf_u32x4_t{pk.x, pk.y, pk.z, pk.w}, rPO,                             \
                                               (q_go[r] != CF_OOB && keep) ? q_go[r] + bq_ : CF_OOB, 0, 0);         \
        raw2[AFF ? r : 0] = cf_load(rX2, q_go[r] + bqn_);                                                           \
      } else {                                                                                                      \
        float g[8];                                                                                                 \
        cf_ld8(GsC_ + ((gate_on && rowg0_ + row >= bound_) ? Kp : 0) + v * 8, g);                                   \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                             \
          const float q = g[e] * fmaf(f[e], sc[e], sh[e]);                                                          \
          f[e] = q * sigmoid_t<bf16_t>(q);                                                                          \
        }                                                                                                           \
        Vec8<bf16_t>::store(dst, f);                                                                                \
      }                                                                                                             \
      raw[r] = cf_load(rX, q_go[r] + bqn_);   /* (past this workgroup's last row: zeros, no memory access) */        \
    }                                                                                                               \
  }
#define CF_MULT(CA)                                                                                                 \
  {                                                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                             \
      const uint4 xb = *reinterpret_cast<const uint4*>((CA) + xfrag + ks * 32);                                     \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                                             \
        const uint4 wa = *reinterpret_cast<const uint4*>(wfrag + ((size_t)ks * 4 * L.img_rows + t * 16) * 8);       \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0); \
      }                                                                                                             \
    }                                                                                                               \
  }
  // result tile -> the wave's staging rows ([row = lane & 15][channel]) -> 16-byte row vectors: store + statistics
#define CF_EPI(TILE)                                                                                                \
  {                                                                                                                 \
    _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                                 \
      *reinterpret_cast<uint2*>(Os + (lane & 15) * NLW + t * 16 + (lane >> 4) * 4) =                                \
          make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));                         \
    const int row0_ = (TILE) * MT + wr * 16;                                                                        \
    _Pragma("unroll") for (int p = 0; p < NPASS; ++p) {                                                             \
      const int row = p * RPO + rr_o;                                                                               \
      const int m = row0_ + row;                                                                                    \
      const bool ok = act_o && row < 16 && m < M32;                                                                 \
      const uint4 rawo = *reinterpret_cast<const uint4*>(Os + (row < 16 ? row : 0) * NLW + v_o * 8);                \
      if constexpr (STATS) {                                                                                        \
        float fo[8];                                                                                                \
        cf_cvt(rawo, fo);                                                                                           \
        const uint32_t keep = ok ? 0xffffffffu : 0u;                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                             \
          const float fv = __uint_as_float(__float_as_uint(fo[j]) & keep);                                          \
          s0[j] += fv; s1[j] += fv * fv;                                                                            \
        }                                                                                                           \
      }                                                                                                             \
      __builtin_amdgcn_raw_buffer_store_b128(cf_u32x4_t{rawo.x, rawo.y, rawo.z, rawo.w}, rY,                        \
                                             ok ? ((uint32_t)m * (uint32_t)Np + (uint32_t)cvec * 8u) * 2u : CF_OOB, 0, 0); \
    }                                                                                                               \
  }

  bf16_t* const buf0 = reinterpret_cast<bf16_t*>(smem + L.a_off);
  bf16_t* const buf1 = reinterpret_cast<bf16_t*>(smem + L.a_off + L.a_bytes);
  if (t0 < t1) CF_CONVERT(t0, buf0)
  CFCLK(2)
  int cur = 0;
  for (int tile = t0; tile < t1; ++tile, cur ^= 1) {
    bf16_t* const bufA = cur ? buf1 : buf0;
    bf16_t* const nxtA = cur ? buf0 : buf1;
    __syncthreads();   // the only barrier per tile: tile `tile` is complete, every wave is past the multiply that read the other buffer
    f32x4_t acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (tile + 1 < t1) {
      // multiply + store + statistics of this tile and the conversion of the next one: ONE basic block
      CF_MULT(bufA)
      CF_CONVERT(tile + 1, nxtA)
      CF_EPI(tile)
      constexpr int VPM = 16;
#pragma unroll
      for (int i = 0; i < NTW * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VPM VALU instructions of the conversion
        if (i & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a re-request
      }
    } else {
      CF_MULT(bufA)
      CF_EPI(tile)
    }
  }
#undef CF_CONVERT
#undef CF_MULT
#undef CF_EPI
  CFCLK(3)

  // ---- BatchNorm_c statistics: lanes -> LDS ([value][lane] per wave; the converted tiles are dead) -> one thread per
  // (sum | sum of squares, channel) adds the row-lanes of the WR waves of its column group -> ONE f64 atomic per value and
  // workgroup, into one of C3D_STAT_STRIPES accumulator sets
  if constexpr (STATS) {
    __syncthreads();
    float* mine = reinterpret_cast<float*>(smem + L.a_off) + (size_t)wave * 16 * 64;   // [16][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j * 64 + lane] = s0[j]; mine[(8 + j) * 64 + lane] = s1[j]; }
    __syncthreads();
    double* dst = a.stats + (size_t)(blockIdx.x % C3D_STAT_STRIPES) * 2 * a.N;
    for (int i = tid; i < 2 * a.N; i += CF_THREADS) {
      const int which = i / a.N, c = i - which * a.N;
      const int cv = c >> 3, j = c & 7;                    // channel = vector cv, element j
      const int wcg = cv / GOW, vo = cv - wcg * GOW;       // column group of the vector, its index inside the group
      float accv = 0.f;
      for (int w_ = 0; w_ < L.WR; ++w_) {
        const float* base = reinterpret_cast<const float*>(smem + L.a_off) + (size_t)(wcg * L.WR + w_) * 16 * 64 + (which * 8 + j) * 64;
        for (int rr = 0; rr < RPO; ++rr) accv += base[rr * GOW + vo];
      }
      atomicAdd(dst + which * a.N + c, (double)accv);
    }
  }
  CFCLK(4)
}

template <int PRO, int NTW, int KS>
int cf_launch(const c3d_pw_args& a, const CfPlan& L, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_cfwd_kernel<PRO, NTW, KS, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  pw_cfwd_kernel<PRO, NTW, KS, true><<<grid, dim3(CF_THREADS), lds, s>>>(a, L);
  return 0;
}

}  // namespace

// Returns C3D_E_UNSUPPORTED for what it does not take (c3d_pw_gemm then runs the first kernel).
__attribute__((visibility("hidden"))) int c3d_detail_pw_cfwd(const c3d_pw_args* args, void* stream) {
  const c3d_pw_args& a = *args;
  if (a.dtype != C3D_DT_BF16 || a.row_mode != C3D_ROWS_DENSE || a.epi_mode != C3D_EPI_STATS) return C3D_E_UNSUPPORTED;
  const bool aff = a.pro_mode == C3D_PRO_AFFINE2;
  if (!aff && a.pro_mode != C3D_PRO_BN_SE_SWISH) return C3D_E_UNSUPPORTED;
  if (!a.w_img || !a.fin.sums || !a.fin.training || a.fin.ticket || a.wg_mode != C3D_WG_NONE || a.bias || !a.stats) return C3D_E_UNSUPPORTED;
  if (aff) {
    // conv_a with the previous block's residual add in its prologue: BatchNorm_c from the completed stripes, y written out
    if (!a.pro_out || !a.x2 || a.fin.batch > 0 || !a.fin.gamma || !a.fin.beta || !a.fin.ss) return C3D_E_UNSUPPORTED;
  } else {
    if (a.fin.batch <= 0 || a.rows_per_sample <= 0) return C3D_E_UNSUPPORTED;
    if (a.se_w1 && (!a.pro_gate || !a.se_b1 || !a.se_w2 || !a.se_b2 || !a.se_hid || a.se_cr <= 0 || a.se_cr > CF_SE_CR))
      return C3D_E_UNSUPPORTED;
    if (!a.se_w1 && a.pro_gate) return C3D_E_UNSUPPORTED;   // gates given in memory (a separate finalize launch made them): first kernel
  }
  if (a.Kp > 224 || a.Np > 224 || a.M < 1024) return C3D_E_UNSUPPORTED;
  if ((a.M + 512) * (int64_t)(a.Kp > a.Np ? a.Kp : a.Np) * 2 >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  const int Kpad = (a.Kp + 31) / 32 * 32, KS = Kpad / 32, Gq = a.Kp >> 3;
  const int ntn = (a.Np + 15) >> 4;
  CfPlan L;
  L.img_rows = (ntn <= 2 ? 2 : ntn <= 4 ? 4 : ntn <= 7 ? 7 : 14) * 16;
  // wave grid: row slabs x output-tile groups.  conv_c: 4 x 2 (3 tiles each) for 5..6 output tiles, else 8 x 1; conv_a: 4 x 2
  // (7 tiles each) for 8..14 tiles, else 8 x 1 with 7 / 4 tiles
  int NTW;
  if (aff) {
    if (ntn > 7) { L.WR = 4; L.WC = 2; NTW = 7; }
    else { L.WR = 8; L.WC = 1; NTW = ntn > 4 ? 7 : 4; }
  } else {
    if (ntn > 6) return C3D_E_UNSUPPORTED;
    if (ntn > 3) { L.WR = 4; L.WC = 2; NTW = 3; }
    else { L.WR = 8; L.WC = 1; NTW = ntn < 2 ? 2 : ntn; }
  }
  L.MT = 16 * L.WR;
  const int rounds = (aff || KS <= 2) ? 2 : 4;
  if ((L.MT * Gq + CF_THREADS - 1) / CF_THREADS > rounds) return C3D_E_UNSUPPORTED;
  if (!aff && a.rows_per_sample < L.MT) return C3D_E_UNSUPPORTED;   // a tile touches at most two samples
  L.KL = Kpad + 8;
  const int64_t tiles = (a.M + L.MT - 1) / L.MT;
  int64_t blocks = device_cus();
  // the narrowest layers (K <= 64, N <= 64: under 128 registers and 80 KB of LDS) run TWO workgroups per CU, out of phase with
  // each other: conv_c on the 128 x 128 maps 73.9 -> 59.7 us per launch (its byte floor: 50), conv_a there unchanged (at its floor)
  if (KS <= 2 && ntn <= 4) blocks *= 2;
  if (blocks > (tiles + 1) / 2) blocks = (tiles + 1) / 2;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  L.tiles_per_wg = tpw;
  const int64_t rows_wg = (int64_t)tpw * L.MT;
  L.ns = (!aff && a.se_w1) ? (int)((rows_wg + a.rows_per_sample - 2) / a.rows_per_sample + 1) : 1;
  if (L.ns > CF_SE_NS) return C3D_E_UNSUPPORTED;
  auto al = [](size_t v) { return (v + 1023) / 1024 * 1024; };
  size_t off = 0;
  L.w_off = 0; off += al((size_t)KS * 4 * L.img_rows * 16);
  L.a_off = (int)off; L.a_bytes = (int)al((size_t)L.MT * L.KL * 2); off += 2 * (size_t)L.a_bytes;
  L.os_wave = 16 * (NTW * 16 + 8) * 2; L.os_off = (int)off; off += al((size_t)8 * L.os_wave);
  // (the statistics dump at the end -- 8 waves x [16][64] f32 -- reuses the converted tiles and the staging rows behind them)
  if (off - L.a_off < (size_t)8 * 16 * 64 * 4) off = L.a_off + (size_t)8 * 16 * 64 * 4;
  L.par_off = (int)off; off += al((size_t)2 * a.Kp * 4);
  L.gate_off = (int)off; off += al((size_t)L.ns * (a.Kp + CF_SE_CR) * 4);
  L.dump_off = (int)off; off += (size_t)CF_THREADS * 16;
  if (off > 160 * 1024) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks);
  int rc = C3D_E_UNSUPPORTED;
  if (!aff) {
    if (NTW == 3 && KS == 7) rc = cf_launch<C3D_PRO_BN_SE_SWISH, 3, 7>(a, L, grid, off, s);
    else if (NTW == 3 && KS == 4) rc = cf_launch<C3D_PRO_BN_SE_SWISH, 3, 4>(a, L, grid, off, s);
    else if (NTW == 2 && KS == 2) rc = cf_launch<C3D_PRO_BN_SE_SWISH, 2, 2>(a, L, grid, off, s);
  } else {
    if (NTW == 7 && KS == 3) rc = cf_launch<C3D_PRO_AFFINE2, 7, 3>(a, L, grid, off, s);        // res4: 96 -> 216
    else if (NTW == 7 && KS == 2) rc = cf_launch<C3D_PRO_AFFINE2, 7, 2>(a, L, grid, off, s);   // res3: 48 -> 108
    else if (NTW == 4 && KS == 1) rc = cf_launch<C3D_PRO_AFFINE2, 4, 1>(a, L, grid, off, s);   // res2: 24 -> 54
  }
  if (rc != 0) return rc;
  C3D_CHECK_LAUNCH();
  return 0;
}
