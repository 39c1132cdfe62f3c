// Weight gradient of the pointwise-convolution family (see pw_gemm.hip for the forward/data path).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "launch_hints.h"
#include "bn_fin.h"
#include <cstdlib>

#ifdef C3D_PW_CLOCK
// Debug build only (tools/pw_phase_clock.py --wgrad): per-phase shader-clock sums, one slot per (workgroup, wave).
constexpr int WCLK_WAVES = 8192;
__device__ unsigned long long c3d_wg_clk[WCLK_WAVES][8];
#define WCLK_DECL unsigned long long wclk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wclk_last_ = __builtin_amdgcn_s_memtime();
#define WCLK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wclk_[i] += t_ - wclk_last_; wclk_last_ = t_; }
#define WCLK_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define WCLK_FLUSH if (lane == 0) { const int w_ = ((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) % WCLK_WAVES; for (int i_ = 0; i_ < 7; ++i_) c3d_wg_clk[w_][i_] += wclk_[i_]; c3d_wg_clk[w_][7] += 1ull; }
#else
#define WCLK_DECL
#define WCLK(i)
#define WCLK_WAITVM
#define WCLK_FLUSH
#endif

namespace {

// =============================================================================================
// Weight gradient:  dW[n, k] += sum_m P(m, n) * Q(m, k)
//
// The reduction runs over data rows, so both MFMA operands are needed "transposed"
// (8 consecutive rows of one channel per lane).  Each thread loads an 8-row x 8-channel
// register block (8 coalescing-friendly 16-byte loads), applies the operand prologue in f32
// and packs along the ROW axis - the transpose is pure register naming - then writes one
// 16-byte LDS vector per channel.  A workgroup shares 64-row tiles; its 4 waves split the
// (n-tile, k-tile) grid and keep <= 4x4 16x16 accumulators each across the whole row range;
// per-workgroup partials go to a workspace and a second tiny kernel reduces them into dW.
// =============================================================================================
template <typename T> struct MmaT;  // transposed-operand LDS tiles [channel][row]
template <> struct MmaT<bf16_t> {
  typedef bf16_t lds_t;
  static constexpr int KSTEP = 32;
  // tile rows of MT + 16 elements = 32 B mod 64 B: the stride class on which the four 16-lane groups of a ds_read_b128 A / B
  // fragment read (lane = channel row lane % 16, 8-row chunk lane / 16) fall on distinct bank quads; + 8 (16 B mod 64 B) made
  // every fragment read 2-way conflicted (tools/lds_bank_model.py: conflict-free row strides are 16 B and 32 B mod 64 B)
  static constexpr int MPAD = 16;
  typedef uint4 frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int ch, int ks, int ml, int lane) {
    return *reinterpret_cast<const uint4*>(base + ch * ml + ks * 32 + (lane >> 4) * 8);
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<bf16_t>::store(p, f); }
};
template <> struct MmaT<float> {
  typedef float lds_t;
  static constexpr int KSTEP = 4;
  static constexpr int MPAD = 4;
  typedef float frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int ch, int ks, int ml, int lane) {
    return base[ch * ml + ks * 4 + (lane >> 4)];
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<float>::store(p, f); }
};

constexpr int WG_THREADS = 512;  // 8 waves: threads 0-255 stage P, 256-511 stage Q
constexpr int WG_MAXMT = 128;    // rows per tile (multiple of 32; taller for narrow channel counts)

__device__ __forceinline__ int64_t q_row_offset(const c3d_pw_wgrad_args& a, int64_t m) {
  if (a.row_mode == C3D_ROWS_DENSE) return m * a.Kp;
  if (a.row_mode == C3D_ROWS_FRAME) {
    const uint32_t g = (uint32_t)m / (uint32_t)a.rpg;
    const uint32_t r = (uint32_t)m - g * (uint32_t)a.rpg;
    return (int64_t)g * a.gstride + (int64_t)r * a.Kp;
  }
  const uint32_t Wo = (uint32_t)a.W >> 1, Ho = (uint32_t)a.H >> 1;
  const uint32_t um = (uint32_t)m;
  const uint32_t wo = um % Wo;
  const uint32_t t = um / Wo;
  const uint32_t ho = t % Ho;
  const uint32_t bt = t / Ho;
  if (a.row_mode == C3D_ROWS_STRIDE2) return (((int64_t)bt * a.H + 2 * ho) * a.W + 2 * wo) * a.Kp;
  // S2SHIFT: pixel (2ho + dy, 2wo + dx), rows outside the image read as zero (offset -1)
  const int yy = 2 * (int)ho + a.dy, xx = 2 * (int)wo + a.dx;
  if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) return -1;
  return (((int64_t)bt * a.H + yy) * a.W + xx) * a.Kp;
}

template <typename T> struct RawW;
template <> struct RawW<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ type zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RawW<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ type zero() { type t; t.a = make_float4(0, 0, 0, 0); t.b = t.a; return t; }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};

// Transpose buffer: WG_RPT rows x 8 channels per thread, written out as one row-vector per channel
// (the transpose is pure register naming).
constexpr int WG_RPT = 4;
template <typename T> struct ColBuf;
template <> struct ColBuf<bf16_t> {
  uint32_t w[8][2];
  __device__ __forceinline__ void put2(int rp, const float (&f0)[8], const float (&f1)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j][rp] = pack_bf16x2(f0[j], f1[j]);
  }
  __device__ __forceinline__ void store(bf16_t* dst, int j) const {
    *reinterpret_cast<uint2*>(dst) = make_uint2(w[j][0], w[j][1]);
  }
};
template <> struct ColBuf<float> {
  float w[8][4];
  __device__ __forceinline__ void put2(int rp, const float (&f0)[8], const float (&f1)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j][2 * rp] = f0[j]; w[j][2 * rp + 1] = f1[j]; }
  }
  __device__ __forceinline__ void store(float* dst, int j) const {
    *reinterpret_cast<float4*>(dst) = make_float4(w[j][0], w[j][1], w[j][2], w[j][3]);
  }
};

// HASP2: the P operand carries the AFFINE2 prologue (second tensor + coefficients)
// QD: the Q rows are dense and unshifted (row m at m * Kp) -- tile base in scalar registers + a 32-bit lane offset
// instead of q_row_offset()'s mode switch and 64-bit multiplies per row ("issue next tile" was 18-25 % of a wave's
// time in this VALU-bound kernel).
// TN x TK: 16x16 output tiles per wave (compile time).  The LDS tiles are padded with zero channel rows up to TN * WN and
// TK * WK tiles, so the multiply phase has no conditions: every fragment read and MFMA is unconditional (an MFMA on zero
// rows costs nothing here -- the matrix cores are ~2 % busy), fully unrolled, and the fragments of k-step s+1 are read
// while step s multiplies.  (Round 2's phase clocks: the runtime-bounded, branchy multiply phase took 2 700 clocks per
// 128-row tile for FOUR MFMAs per wave on the 24x54 layers.)
template <typename T, bool HASP2, bool QD, int TN, int TK>
__global__ __launch_bounds__(WG_THREADS) void pw_wgrad_kernel(const c3d_pw_wgrad_args a_in, const int tiles_per_wg,
                                                              const int WN, const int WK, const int MT) {
  c3d_pw_wgrad_args a = a_in;
  if (a.taps > 1) {   // batched ConvTranspose taps: blockIdx.y selects the (dy, dx) shift of the q rows
    a.dy = (int)blockIdx.y / 4 - 1;
    a.dx = (int)blockIdx.y % 4 - 1;
  }
  typedef MmaT<T> MM;
  typedef typename MM::lds_t lds_t;
  typedef RawW<T> RW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WCLK_DECL
  const int Kp = a.Kp, Np = a.Np;
  const int NT = TN * WN;   // padded tile counts (>= ceil(Np / 16), ceil(Kp / 16))
  const int ML = MT + MM::MPAD;
  const size_t buf_elems = (size_t)(TN * WN + TK * WK) * 16 * ML;
  lds_t* base = reinterpret_cast<lds_t*>(smem);

  // zero both buffers once (covers the channel-padding rows, which are never written again)
  for (size_t i = (size_t)tid * 8; i < 2 * buf_elems; i += (size_t)WG_THREADS * 8) {
    float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MM::store8(base + i, z);
  }

  const int Gp = Np >> 3, Gq = Kp >> 3;
  const int half = tid >> 8;                 // 0: stages the P tile, 1: stages the Q tile
  const int t256 = tid & 255;
  const int nrg = MT / WG_RPT;               // row groups (WG_RPT rows each) per tile
  const int vv = t256 / nrg, rg = t256 - vv * nrg;
  const bool p_act = half == 0 && vv < Gp, q_act = half == 1 && vv < Gq;
  float cA[8], cB[8], cC[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cA[j] = 1.f; cB[j] = 0.f; cC[j] = 0.f; }
  if (HASP2 && p_act) {
    if (a.p_fin.sums) {   // coefficients rebuilt from the producer's sums (csrc/bn_fin.h); nothing is accumulated here
      c3dfin::bn_bwd_coef_consume8(a.p_fin, a.N, Np, vv * 8, cA, cB, cC);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cA[j] = a.p_coef[vv * 8 + j]; cB[j] = a.p_coef[Np + vv * 8 + j]; cC[j] = a.p_coef[2 * Np + vv * 8 + j];
      }
    }
  }
  if (q_act && a.q_mode == C3D_PRO_BN_SE_SWISH) {  // Q side reuses cA/cB as scale/shift
#pragma unroll
    for (int j = 0; j < 8; ++j) { cA[j] = a.q_ss[vv * 8 + j]; cB[j] = a.q_ss[Kp + vv * 8 + j]; }
  }

  const int wn_i = wave % WN, wk_i = wave / WN;
  f32x4_t acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const T* P = reinterpret_cast<const T*>(a.p);
  const T* P2 = reinterpret_cast<const T*>(a.p2);
  const T* Q = reinterpret_cast<const T*>(a.q);
  const int64_t tiles = (a.M + MT - 1) / MT;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wg;
  int64_t t1 = t0 + tiles_per_wg;
  if (t1 > tiles) t1 = tiles;

  typename RW::type r1[WG_RPT];
  typename RW::type r2[HASP2 ? WG_RPT : 1];
  unsigned vmask = 0;  // which of the WG_RPT rows were real

  const int lane_row0 = rg * WG_RPT;                              // first tile row of this thread
  const int p_lane = lane_row0 * Np + vv * 8, q_lane = lane_row0 * Kp + vv * 8;
#define WG_ISSUE(TILE)                                                                          \
  {                                                                                             \
    const int64_t tb_ = (int64_t)(TILE) * MT;                     /* wave-uniform */             \
    const int64_t left64_ = a.M - tb_;                                                          \
    const int left_ = left64_ > MT ? MT : (int)left64_;           /* rows of this tile that exist */ \
    const T* Pt_ = P + tb_ * Np;                                                                \
    const T* P2t_ = HASP2 ? P2 + tb_ * Np : P;                                                  \
    const T* Qt_ = Q + tb_ * Kp;                                                                \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int r = 0; r < WG_RPT; ++r) {                                        \
      r1[r] = RW::zero();                                                                       \
      if (HASP2) r2[HASP2 ? r : 0] = RW::zero();                                                \
      if (lane_row0 + r < left_) {                                                              \
        if (p_act) {                                                                            \
          r1[r] = RW::load(Pt_ + (p_lane + r * Np));                                            \
          if (HASP2) r2[HASP2 ? r : 0] = RW::load(P2t_ + (p_lane + r * Np));                    \
          vmask |= 1u << r;                                                                     \
        } else if (q_act) {                                                                     \
          if constexpr (QD) {                                                                   \
            r1[r] = RW::load(Qt_ + (q_lane + r * Kp)); vmask |= 1u << r;                        \
          } else {                                                                              \
            const int64_t qo_ = q_row_offset(a, tb_ + lane_row0 + r);                           \
            if (qo_ >= 0) { r1[r] = RW::load(Q + qo_ + vv * 8); vmask |= 1u << r; }             \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
  }

  // Swish/SE gate of the sample this thread is in: cached across tiles (it used to be re-read, behind an
  // s_waitcnt vmcnt(0), for every tile -- a global round trip on the critical Q-staging waves)
  const bool swish = q_act && a.q_mode == C3D_PRO_BN_SE_SWISH;
  float g[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = 1.f;
  int gn = -1;

  __syncthreads();  // zero fill complete
  if (t0 < t1) WG_ISSUE(t0)
  int cur = 0;
  WCLK(0)
  for (int64_t tile = t0; tile < t1; ++tile, cur ^= 1) {
    WCLK_WAITVM
    WCLK(1)
    lds_t* PT = base + (size_t)cur * buf_elems;
    lds_t* QT = PT + (size_t)NT * 16 * ML;
    // ---- convert + prologue + transpose (register naming) -> LDS -------------------------------
    if (p_act || q_act) {
      ColBuf<T> cb;
      const bool all_real = vmask == (1u << WG_RPT) - 1u;
      int rows_n[WG_RPT];    // sample of each row (swish gate); the thread's rows are consecutive
      bool one_sample = true;
      if (swish && a.q_gate && vmask) {   // (a thread without a real row must not index the gate: m0 / rps may be >= B)
        const int64_t m0 = tile * MT + rg * WG_RPT;
        const int64_t rps = a.rows_per_sample;
        if (gn < 0 || m0 < (int64_t)gn * rps || m0 >= (int64_t)(gn + 1) * rps) {   // rarely: the cached sample moved on
          const int n0 = (int)((uint32_t)m0 / (uint32_t)rps);
          one_sample = m0 + WG_RPT - 1 < (int64_t)(n0 + 1) * rps;
          if (one_sample) {
            const float* gp = a.q_gate + (int64_t)n0 * Kp + vv * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = gp[j];
            gn = n0;
          }
        } else {
          one_sample = m0 + WG_RPT - 1 < (int64_t)(gn + 1) * rps;
        }
        if (!one_sample) {
#pragma unroll
          for (int r = 0; r < WG_RPT; ++r) rows_n[r] = (int)((uint32_t)(m0 + r) / (uint32_t)rps);
        }
      }
#pragma unroll
      for (int rp = 0; rp < WG_RPT / 2; ++rp) {
        float fr[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 2 * rp + h;
          float (&f)[8] = fr[h];
          RW::cvt(r1[r], f);
          if (HASP2 && p_act) {
            float f2[8];
            RW::cvt(r2[HASP2 ? r : 0], f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
          } else if (swish) {
            if (!one_sample && ((vmask >> r) & 1u) && rows_n[r] != gn) {   // a row group that straddles two samples
              const float* gp = a.q_gate + (int64_t)rows_n[r] * Kp + vv * 8;
#pragma unroll
              for (int j = 0; j < 8; ++j) g[j] = gp[j];
              gn = rows_n[r];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float qv = g[j] * fmaf(f[j], cA[j], cB[j]);
              f[j] = qv * sigmoid_t<T>(qv);
            }
          }
          if (!all_real && !((vmask >> r) & 1u)) {   // rows past the end of the tensor (last tile) / outside the image
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
          }
        }
        cb.put2(rp, fr[0], fr[1]);
      }
      lds_t* dst = (p_act ? PT : QT) + (size_t)(vv * 8) * ML + rg * WG_RPT;
#pragma unroll
      for (int j = 0; j < 8; ++j) cb.store(dst + (size_t)j * ML, j);
    }
    WCLK(2)
    // ---- prefetch the next tile while this one is multiplied -----------------------------------
    if (tile + 1 < t1) WG_ISSUE(tile + 1)
    WCLK(3)
    __syncthreads();  // the only barrier per tile (LDS tiles are double buffered)
    WCLK(4)
    {
      const int KS = MT / MM::KSTEP;
      const lds_t* pbase = PT + (size_t)(wn_i * 16 + (lane & 15)) * ML;   // tile (wn_i + i * WN) is i * WN * 16 rows on
      const lds_t* qbase = QT + (size_t)(wk_i * 16 + (lane & 15)) * ML;
      const int pstep = WN * 16 * ML, qstep = WK * 16 * ML;
      typename MM::frag_t pa[2][TN], qb[2][TK];
#define WG_FRAGS(SLOT, KSI)                                                                                  \
  {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TN; ++i) pa[SLOT][i] = MM::load(pbase + i * pstep, 0, (KSI), ML, lane); \
    _Pragma("unroll") for (int j = 0; j < TK; ++j) qb[SLOT][j] = MM::load(qbase + j * qstep, 0, (KSI), ML, lane); \
  }
#define WG_MMA(SLOT)                                                                                         \
  {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < TK; ++j) acc[i][j] = MM::mma(pa[SLOT][i], qb[SLOT][j], acc[i][j]); \
  }
      WG_FRAGS(0, 0)
      for (int ks = 0; ks < KS; ks += 2) {
        if (ks + 1 < KS) WG_FRAGS(1, ks + 1)
        WG_MMA(0)
        if (ks + 1 < KS) {
          if (ks + 2 < KS) WG_FRAGS(0, ks + 2)
          WG_MMA(1)
        }
      }
#undef WG_FRAGS
#undef WG_MMA
    }
    WCLK(5)
  }
#undef WG_ISSUE

  // partials -> workspace [grid][N][K]
  float* wsb = a.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.N * a.K;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int nt = wn_i + i * WN;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int kt = wk_i + j * WK;
      const int k = kt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + (lane >> 4) * 4 + r;
        if (n < a.N && k < a.K) wsb[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  }
  WCLK(6)
  WCLK_FLUSH
}

// dW += sum over per-workgroup partials.  32 outputs per block x 8 part-groups: each thread adds at
// most parts/8 coalesced values with 4 loads in flight (the serial 256-deep walk cost 55 us/call).
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ ws, float* dw, int N, int K,
                                                              int parts, int sn, int sk, int tap_stride) {
  __shared__ float red[8][32];
  ws += (size_t)blockIdx.y * parts * N * K;   // batched taps: one slab of partials per tap
  dw += (size_t)blockIdx.y * tap_stride;
  const int e = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + e;
  const int NK = N * K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < NK) {
    int p = pg;
    for (; p + 24 < parts; p += 32) {
      s0 += ws[(size_t)p * NK + idx];
      s1 += ws[(size_t)(p + 8) * NK + idx];
      s2 += ws[(size_t)(p + 16) * NK + idx];
      s3 += ws[(size_t)(p + 24) * NK + idx];
    }
    for (; p < parts; p += 8) s0 += ws[(size_t)p * NK + idx];
  }
  red[pg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pg == 0 && idx < NK) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][e];
    const int n = idx / K, k = idx - n * K;
    dw[(size_t)n * sn + (size_t)k * sk] += s;
  }
}

constexpr int WGRAD_MAX_PARTS = 512;

// One instantiation: sets the LDS attribute once, launches.
template <typename T, bool HASP2, bool QD, int TN, int TK>
int launch_wgrad_inst(const c3d_pw_wgrad_args& a, dim3 grid, size_t lds, int tpw, int WN, int WK, int MT, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_kernel<T, HASP2, QD, TN, TK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  pw_wgrad_kernel<T, HASP2, QD, TN, TK><<<grid, dim3(WG_THREADS), lds, stream>>>(a, tpw, WN, WK, MT);
  return 0;
}

// Instantiated per-wave tile grids (TN x TK): a launch takes the smallest that covers its ceil(NT / WN) x ceil(KT / WK).
struct WgInst { int tn, tk; };
constexpr WgInst WG_INSTS[] = {{1, 1}, {2, 2}, {3, 4}, {4, 3}, {4, 4}};

template <typename T, bool HASP2, bool QD>
int launch_wgrad_pick2(const c3d_pw_wgrad_args& a, int inst, dim3 grid, size_t lds, int tpw, int WN, int WK, int MT, hipStream_t s) {
  switch (inst) {
    case 0: return launch_wgrad_inst<T, HASP2, QD, 1, 1>(a, grid, lds, tpw, WN, WK, MT, s);
    case 1: return launch_wgrad_inst<T, HASP2, QD, 2, 2>(a, grid, lds, tpw, WN, WK, MT, s);
    case 2: return launch_wgrad_inst<T, HASP2, QD, 3, 4>(a, grid, lds, tpw, WN, WK, MT, s);
    case 3: return launch_wgrad_inst<T, HASP2, QD, 4, 3>(a, grid, lds, tpw, WN, WK, MT, s);
    default: return launch_wgrad_inst<T, HASP2, QD, 4, 4>(a, grid, lds, tpw, WN, WK, MT, s);
  }
}
template <typename T, bool HASP2>
int launch_wgrad_pick(const c3d_pw_wgrad_args& a, bool qd, int inst, dim3 grid, size_t lds, int tpw, int WN, int WK, int MT,
                      hipStream_t s) {
  return qd ? launch_wgrad_pick2<T, HASP2, true>(a, inst, grid, lds, tpw, WN, WK, MT, s)
            : launch_wgrad_pick2<T, HASP2, false>(a, inst, grid, lds, tpw, WN, WK, MT, s);
}

template <typename T>
int launch_wgrad(const c3d_pw_wgrad_args& a, hipStream_t stream) {
  typedef MmaT<T> MM;
  const int NT = (a.Np + 15) >> 4, KT = (a.Kp + 15) >> 4;
  const int taps = a.taps > 1 ? a.taps : 1;
  const bool qd = a.row_mode == C3D_ROWS_DENSE && taps == 1;
  // wave grid: WN*WK = 8 with ceil(NT/WN) <= 4 and ceil(KT/WK) <= 4
  int WN = 0, WK = 0, tn_need = 0, tk_need = 0;
  const int cand[4][2] = {{8, 1}, {4, 2}, {2, 4}, {1, 8}};
  int best = 1 << 30;
  for (int c = 0; c < 4; ++c) {
    const int tn = (NT + cand[c][0] - 1) / cand[c][0], tk = (KT + cand[c][1] - 1) / cand[c][1];
    if (tn > 4 || tk > 4) continue;
    // the instantiated grid that will run (zero-padded) is what costs: MFMAs dominate, then fragment loads
    int ti = 4, tj = 4;
    for (int i = 0; i < 5; ++i)
      if (WG_INSTS[i].tn >= tn && WG_INSTS[i].tk >= tk) { ti = WG_INSTS[i].tn; tj = WG_INSTS[i].tk; break; }
    const int cost = ti * tj * 4 + ti + tj;
    if (cost < best) { best = cost; WN = cand[c][0]; WK = cand[c][1]; tn_need = tn; tk_need = tk; }
  }
  if (WN == 0) return C3D_E_UNSUPPORTED;
  int inst = 4;
  for (int i = 0; i < 5; ++i)
    if (WG_INSTS[i].tn >= tn_need && WG_INSTS[i].tk >= tk_need) { inst = i; break; }
  const int TNi = WG_INSTS[inst].tn, TKi = WG_INSTS[inst].tk;
  // rows per tile: as tall as 256 staging threads per operand allow (WG_RPT rows x 8 channels each)
  const int maxG = (a.Np > a.Kp ? a.Np : a.Kp) >> 3;
  int MT = (256 / maxG) * WG_RPT / 32 * 32;
  if (MT > WG_MAXMT) MT = WG_MAXMT;
  static const int mt_env = c3d_env("C3D_WG_MT") ? atoi(c3d_env("C3D_WG_MT")) : 0;  // tuning knob
  if (mt_env >= 32 && mt_env < MT) MT = mt_env / 32 * 32;
  if (MT < 32) return C3D_E_UNSUPPORTED;
  size_t lds = 0;
  for (; MT >= 32; MT -= 32) {
    lds = (size_t)2 * (TNi * WN + TKi * WK) * 16 * (MT + MM::MPAD) * sizeof(typename MM::lds_t);   // padded tile rows
    if (lds <= 160 * 1024) break;
  }
  if (MT < 32) return C3D_E_UNSUPPORTED;
  const int64_t tiles = (a.M + MT - 1) / MT;
  int64_t blocks = (tiles + 3) / 4;  // >= 4 tiles per workgroup when there is enough work
  static const int cap_env = c3d_env("C3D_WG_BLOCKS") ? atoi(c3d_env("C3D_WG_BLOCKS")) : 0;  // tuning knob
  int64_t cap = device_cus() < WGRAD_MAX_PARTS ? device_cus() : WGRAD_MAX_PARTS;
  if (cap_env > 0 && cap_env <= WGRAD_MAX_PARTS) cap = cap_env;
  else if (c3d_side_launch) {
    // beside the data-gradient chain (stage driver's side stream): a cap on the workgroups (history below; 7/8 of the CUs now).  This single-round
    // kernel at full width holds every CU for its whole duration (launch_hints.h); measured on MI355X, B=32 bf16,
    // 60-step runs: 256 / 208 / 192 / 176 / 160 / 128 workgroups -> 32.52 / 32.08 / 31.84 / 32.11 / 32.48 / 32.87 ms
    static const int side_env = c3d_env("C3D_PWWG_SIDE_WGS") ? atoi(c3d_env("C3D_PWWG_SIDE_WGS")) : 0;
    // round 5 (after the data-gradient kernels' waits became exact, same-call sweeps through the instrumented build): 128 / 144 /
    // 160 / 176 / 192 / 256 workgroups -> 23.01 / 22.99 / 22.86 / 23.53 / 23.18 / 23.34 ms per step: 5/8 of the CUs.
    // ...and once c3d_block_out_bwd was folded into the conv_a data gradient (the elementwise pass that used to fill the CUs a
    // narrow weight gradient left): 96 / 128 / 160 / 192 / 208 / 224 / 240 / 256 -> 23.46 / 22.81 / 22.70 / 22.53 / 22.34 / 22.29 /
    // 22.33 / 22.34 ms (SCD and CC: 224 best by 0.5 % too): 7/8 of the CUs
    const int64_t side_cap = side_env > 0 ? side_env : (int64_t)device_cus() * 7 / 8;
    if (side_cap < cap) cap = side_cap;
  }
  if (taps > 1 && cap > WGRAD_MAX_PARTS / taps) cap = WGRAD_MAX_PARTS / taps;   // the workspace holds MAX_PARTS slabs
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  const dim3 grid((unsigned)blocks, taps);
  const int rc = (a.p_coef || a.p_fin.sums) ? launch_wgrad_pick<T, true>(a, qd, inst, grid, lds, tpw, WN, WK, MT, stream)
                                            : launch_wgrad_pick<T, false>(a, qd, inst, grid, lds, tpw, WN, WK, MT, stream);
  if (rc != 0) return rc;
  C3D_CHECK_LAUNCH();
  const int nk = a.N * a.K;
  pw_wgrad_reduce_kernel<<<dim3((nk + 31) / 32, taps), dim3(256), 0, stream>>>(a.ws, a.dw, a.N, a.K, (int)blocks,
                                                                                  a.dw_sn, a.dw_sk, a.dw_tap_stride);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// the reducer alone: c3d_pw_gemm's fused weight gradient (pw_gemm_impl.h) writes per-workgroup partials [parts][N][K]
__attribute__((visibility("hidden"))) int c3d_detail_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int parts, int sn,
                                                                     int sk, hipStream_t stream) {
  const int nk = N * K;
  pw_wgrad_reduce_kernel<<<dim3((nk + 31) / 32, 1), dim3(256), 0, stream>>>(ws, dw, N, K, parts, sn, sk, 0);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t c3d_pw_wgrad_ws_floats(int32_t N, int32_t K) { return (int64_t)WGRAD_MAX_PARTS * N * K; }

int c3d_detail_pw_wgrad_wide(const c3d_pw_wgrad_args* args, void* stream);   // pw_wide.hip
int c3d_detail_pw_wgrad_v2(const c3d_pw_wgrad_args* args, hipStream_t stream);   // pw_wgrad_v2.hip: bf16, dense rows
int c3d_detail_pw_wgrad_v2_flush(hipStream_t stream);                             // pending partials of a chained launch

extern "C" int c3d_pw_wgrad(const c3d_pw_wgrad_args* args, void* stream) {
  if (!args || !args->p || !args->q || !args->dw || !args->ws) return C3D_E_BADARG;
  const c3d_pw_wgrad_args& a = *args;
  if (a.M <= 0 || (a.Kp & 7) || (a.Np & 7) || a.K > a.Kp || a.N > a.Np) return C3D_E_BADARG;
  if (a.Kp > 224 || a.Np > 224) {   // wide layers (res5, caption-decoder linears): block-tiled kernel, f32 atomics
    if (a.Kp > 1024 || a.Np > 1024) return C3D_E_UNSUPPORTED;
    if ((a.p_coef || a.p_fin.sums) && !a.p2) return C3D_E_BADARG;
    if (a.q_mode == C3D_PRO_BN_SE_SWISH && (!a.q_ss || (a.q_gate && a.rows_per_sample <= 0))) return C3D_E_BADARG;
    return c3d_detail_pw_wgrad_wide(args, stream);
  }
  if ((a.p_coef || a.p_fin.sums) && !a.p2) return C3D_E_BADARG;
  if (a.q_mode == C3D_PRO_BN_SE_SWISH && (!a.q_ss || (a.q_gate && a.rows_per_sample <= 0))) return C3D_E_BADARG;
  if (a.M >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = C3D_E_BADARG;
  if (a.dtype == C3D_DT_F32) rc = launch_wgrad<float>(a, s);
  else if (a.dtype == C3D_DT_BF16) {
    // flat-staged, transposing-read kernel for dense rows (C3D_OPT_PW_WGRAD_V2); what it does not take runs here
    rc = c3d_option_pw_wgrad_v2 ? c3d_detail_pw_wgrad_v2(args, s) : C3D_E_UNSUPPORTED;
    if (rc == C3D_E_UNSUPPORTED) {
      // (a launch of this kernel may reuse the workspace pending partials sit in: they are reduced first)
      if (a.chain) { const int rcf = c3d_detail_pw_wgrad_v2_flush(s); if (rcf != 0) return rcf; }
      rc = launch_wgrad<bf16_t>(a, s);
    }
  }
  if (rc == C3D_E_UNSUPPORTED) rc = c3d_detail_pw_wgrad_wide(args, stream);   // shapes that do not fit its LDS plan
  return rc;
}

extern "C" int c3d_pw_wgrad_flush(void* stream) { return c3d_detail_pw_wgrad_v2_flush(reinterpret_cast<hipStream_t>(stream)); }

#ifdef C3D_PW_CLOCK
extern "C" int c3d_debug_wgrad_clock(unsigned long long* out, int reset) {   // out[WCLK_WAVES][8]
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_wg_clk), sizeof(unsigned long long) * WCLK_WAVES * 8);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(c3d_wg_clk));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * WCLK_WAVES * 8);
  }
  return (int)e;
}
#endif
