// Pointwise-convolution / linear-layer GEMM for WIDE layers (K or N above the 224 channels the wave-private-tile
// kernel of pw_gemm_impl.h holds in LDS): the res5 stage of X3D-L (96/192 -> 432 -> 192 channels; only the
// change-captioning path executes it: reference model/trainer.py:120-124) and the linear layers of the caption decoder
// (reference model/caption_decoder.py: nn.MultiheadAttention in/out projections 192 -> 576 / 192, `wdc` 192 -> vocab).
// Same C ABI (c3d_pw_gemm / c3d_pw_wgrad dispatch here by size), same fused prologues and epilogues, both storage types.
//
// These layers are SMALL (M = B*3*16*16 = 12 288 rows at B=16, weights 83 k): a classic block-tiled GEMM -- a
// workgroup owns a 64-row x 112-column output block, streams K in LDS chunks, and runs the fused epilogue from an f32
// copy of the block in LDS.  The weight gradient contracts over rows, so its operand tiles are written TRANSPOSED into
// LDS (the 16-byte MFMA operand of a lane is 8 consecutive rows of one channel).
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "bn_fin.h"

namespace {

constexpr int WB_M = 64, WB_NT = 7, WB_N = WB_NT * 16;   // output block: 64 rows x 112 columns
constexpr int WB_NV = WB_N / 8, WB_RG = 256 / WB_NV;     // epilogue map: 14 column vectors x 18 row groups
constexpr int WB_SLOTS = 5;                              // batch samples a 64-row block can touch (rows_per_sample >= 16)

__device__ __forceinline__ int64_t wide_row_offset(const c3d_pw_args& a, int64_t m) {
  if (a.row_mode == C3D_ROWS_STRIDE2) {
    const int64_t Wo = a.W >> 1, Ho = a.H >> 1;
    const int64_t wo = m % Wo, t = m / Wo, ho = t % Ho, bt = t / Ho;
    return ((bt * a.H + 2 * ho) * a.W + 2 * wo) * a.Kp;
  }
  return m * a.Kp;
}

template <typename T, int PRO, int EPI>
__global__ __launch_bounds__(256) void pw_wide_kernel(const c3d_pw_args a) {
  typedef Mma<T> MM;
  typedef typename MM::lds_t lds_t;
  constexpr int KC = sizeof(T) == 2 ? 64 : 32;          // K chunk
  constexpr int KL = KC + MM::KPAD;
  constexpr int KS = KC / MM::KSTEP;
  constexpr int OL = WB_N + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_t* Xs = reinterpret_cast<lds_t*>(smem);                       // [64][KL]
  lds_t* Ws = Xs + WB_M * KL;                                       // [112][KL]
  float* Os = reinterpret_cast<float*>(Ws + WB_N * KL);             // [64][OL]
  float* red = Os + WB_M * OL;                                      // SWISH_SE_BWD: [5][112][3] 64-bit fixed-point sums
  float* Pp = red + WB_SLOTS * WB_N * 3 * 2;                        // prologue parameters [3][Kp] (scale|shift or A|B|C)
  // SWISH_SE_BWD: the per-(sample, channel) sums of the workgroup are accumulated in 2^-40 FIXED POINT (64-bit integer LDS
  // atomics): integer addition is associative, so the sums -- and through the BatchNorm / SE coefficients built from them
  // the whole data gradient below res5 -- do not depend on the order in which the row groups arrive.  (f32 LDS atomics
  // here made the CC encoder gradient differ by ~1e-2 between two identical bf16 runs: the sum t1*bhat nearly cancels.)
  unsigned long long* red64 = reinterpret_cast<unsigned long long*>(red);
  constexpr double FIX = 1099511627776.0;   // 2^40: resolution 9e-13, range +-8e6
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * WB_M;
  const int n0 = (int)blockIdx.y * WB_N;
  const int Kp = a.Kp, Np = a.Np;
  const T* X = reinterpret_cast<const T*>(a.x);
  const T* X2 = reinterpret_cast<const T*>(a.x2);
  const int64_t rps = a.rows_per_sample > 0 ? a.rows_per_sample : a.M;
  f32x4_t acc[WB_NT];
#pragma unroll
  for (int nt = 0; nt < WB_NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (PRO == C3D_PRO_AFFINE2 && a.fin.sums) {
    // BatchNorm-backward coefficients rebuilt from the producer's completed sums (csrc/bn_fin.h; no c3d_bn_bwd_coef launch
    // in front of this kernel); workgroup (0, 0) also accumulates dgamma / dbeta
    for (int c = tid; c < Kp; c += 256) {
      float cA, cB, cC;
      c3dfin::bn_bwd_coef_consume(a.fin, a.K, Kp, c, blockIdx.x == 0 && blockIdx.y == 0, cA, cB, cC);
      Pp[c] = cA; Pp[Kp + c] = cB; Pp[2 * Kp + c] = cC;
    }
  } else if (PRO != C3D_PRO_NONE) {
    const int np = (PRO == C3D_PRO_AFFINE2 ? 3 : 2) * Kp;
    for (int i = tid; i < np; i += 256) Pp[i] = a.pro_p[i];
  }
  // weights: 4 consecutive elements along their contiguous dimension per load when the layout allows it
  const bool kc_contig = a.w_sk == 1;
  const bool w_vec4 = ((uintptr_t)a.w & 15) == 0 && (kc_contig ? ((a.w_sn & 3) == 0 && (a.K & 3) == 0)
                                                               : (a.w_sn == 1 && (a.w_sk & 3) == 0 && (a.N & 3) == 0));

  // Software pipeline over the K chunks: the global loads of chunk c+1 (raw X vectors, SE gates, weights) are issued into
  // registers before the MFMAs of chunk c and converted / written to LDS after them -- the loop used to load, convert,
  // barrier and multiply in turn, paying a full global round trip per 64-channel chunk (7 per launch on the 432-channel
  // res5 layers: 40-66 us for 15 MB of operands).
  constexpr int NXI = (WB_M * (KC / 8) + 255) / 256;       // X vectors per thread and chunk
  constexpr int NWI = (WB_N * KC / 4 + 255) / 256;         // weight float4s per thread and chunk (w_vec4 layouts)
  typedef typename Vec8<T>::raw_t xraw_t;
  xraw_t rx[NXI], rx2[PRO == C3D_PRO_AFFINE2 ? NXI : 1];
  float4 rgate[PRO == C3D_PRO_BN_SE_SWISH ? NXI : 1][2];
  float4 rw[NWI];
  unsigned xmask = 0;
  auto fetch = [&](const int kc) {
    xmask = 0;
#pragma unroll
    for (int s_ = 0; s_ < NXI; ++s_) {
      const int i = tid + s_ * 256;
      const int kv = i % (KC / 8), r = i / (KC / 8);
      const int64_t m = m0 + r;
      const int k0 = kc + kv * 8;
      if (i < WB_M * (KC / 8) && m < a.M && k0 < Kp) {
        const int64_t off = wide_row_offset(a, m) + k0;
        rx[s_] = Vec8<T>::load_raw(X + off);
        if (PRO == C3D_PRO_AFFINE2) rx2[PRO == C3D_PRO_AFFINE2 ? s_ : 0] = Vec8<T>::load_raw(X2 + off);
        if (PRO == C3D_PRO_BN_SE_SWISH && a.pro_gate) {
          const float* gp = a.pro_gate + (m / rps) * Kp + k0;
          rgate[PRO == C3D_PRO_BN_SE_SWISH ? s_ : 0][0] = *reinterpret_cast<const float4*>(gp);
          rgate[PRO == C3D_PRO_BN_SE_SWISH ? s_ : 0][1] = *reinterpret_cast<const float4*>(gp + 4);
        }
        xmask |= 1u << s_;
      }
    }
    if (w_vec4) {
#pragma unroll
      for (int s_ = 0; s_ < NWI; ++s_) {
        const int i = tid + s_ * 256;
        int n, k;
        if (kc_contig) { k = (i % (KC / 4)) * 4; n = i / (KC / 4); } else { n = (i % (WB_N / 4)) * 4; k = i / (WB_N / 4); }
        const int gn = n0 + n, gk = kc + k;
        rw[s_] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < WB_N * KC / 4 && gn < a.N && gk < a.K)
          rw[s_] = *reinterpret_cast<const float4*>(a.w + (size_t)gn * a.w_sn + (size_t)gk * a.w_sk);
      }
    }
  };
  auto commit = [&](const int kc) {
    // ---- X chunk: 8-element vectors (row, kv), prologue applied, converted to the MFMA operand type
#pragma unroll
    for (int s_ = 0; s_ < NXI; ++s_) {
      const int i = tid + s_ * 256;
      if (i >= WB_M * (KC / 8)) continue;
      const int kv = i % (KC / 8), r = i / (KC / 8);
      const int k0 = kc + kv * 8;
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      if ((xmask >> s_) & 1u) {
        Vec8<T>::cvt_raw(rx[s_], f);
        if (PRO == C3D_PRO_BN_SE_SWISH) {
          const float4 g0 = rgate[PRO == C3D_PRO_BN_SE_SWISH ? s_ : 0][0], g1 = rgate[PRO == C3D_PRO_BN_SE_SWISH ? s_ : 0][1];
          const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float q = (a.pro_gate ? g[j] : 1.f) * fmaf(f[j], Pp[k0 + j], Pp[Kp + k0 + j]);
            f[j] = q * sigmoid_t<T>(q);
          }
        } else if (PRO == C3D_PRO_AFFINE2) {
          float f2[8];
          Vec8<T>::cvt_raw(rx2[PRO == C3D_PRO_AFFINE2 ? s_ : 0], f2);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaf(Pp[k0 + j], f[j], fmaf(Pp[2 * Kp + k0 + j], f2[j], Pp[Kp + k0 + j]));
        }
      }
      MM::store8(Xs + r * KL + kv * 8, f);
    }
    // ---- W chunk: Ws[n][k] = w[(n0+n)*w_sn + (kc+k)*w_sk]; threads run along the contiguous dimension of w
    if (w_vec4) {
#pragma unroll
      for (int s_ = 0; s_ < NWI; ++s_) {
        const int i = tid + s_ * 256;
        if (i >= WB_N * KC / 4) continue;
        int n, k;
        if (kc_contig) { k = (i % (KC / 4)) * 4; n = i / (KC / 4); } else { n = (i % (WB_N / 4)) * 4; k = i / (WB_N / 4); }
        const float4 v = rw[s_];
        if (kc_contig) {
          Ws[n * KL + k] = MM::cvt(v.x); Ws[n * KL + k + 1] = MM::cvt(v.y); Ws[n * KL + k + 2] = MM::cvt(v.z); Ws[n * KL + k + 3] = MM::cvt(v.w);
        } else {
          Ws[n * KL + k] = MM::cvt(v.x); Ws[(n + 1) * KL + k] = MM::cvt(v.y); Ws[(n + 2) * KL + k] = MM::cvt(v.z); Ws[(n + 3) * KL + k] = MM::cvt(v.w);
        }
      }
    } else {
      for (int i = tid; i < WB_N * KC; i += 256) {
        int n, k;
        if (kc_contig) { k = i % KC; n = i / KC; } else { n = i % WB_N; k = i / WB_N; }
        const int gn = n0 + n, gk = kc + k;
        const float v = (gn < a.N && gk < a.K) ? a.w[(size_t)gn * a.w_sn + (size_t)gk * a.w_sk] : 0.f;
        Ws[n * KL + k] = MM::cvt(v);
      }
    }
  };
  fetch(0);
  for (int kc = 0; kc < Kp; kc += KC) {
    __syncthreads();   // (first pass: prologue parameters staged; later: the previous chunk's MFMAs have read Xs / Ws)
    commit(kc);
    __syncthreads();
    if (kc + KC < Kp) fetch(kc + KC);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const typename MM::frag_t xb = MM::load(Xs, wave * 16 + (lane & 15), ks, KL, lane);
#pragma unroll
      for (int nt = 0; nt < WB_NT; ++nt) {
        const typename MM::frag_t wa = MM::load(Ws, nt * 16 + (lane & 15), ks, KL, lane);
        acc[nt] = MM::mma(wa, xb, acc[nt]);
      }
    }
  }
  // ---- block result -> LDS (D[i = column][j = row]: lane holds 4 consecutive columns of one row)
#pragma unroll
  for (int nt = 0; nt < WB_NT; ++nt) {
    float* d = Os + (wave * 16 + (lane & 15)) * OL + nt * 16 + (lane >> 4) * 4;
    *reinterpret_cast<float4*>(d) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
  }
  if (EPI == C3D_EPI_SWISH_SE_BWD) {
    for (int i = tid; i < WB_SLOTS * WB_N * 3; i += 256) red64[i] = 0ull;
  }
  __syncthreads();
  // ---- fused epilogue: thread = (column vector v, row group rg)
  const int v = tid % WB_NV, rg = tid / WB_NV;
  const int c0 = n0 + v * 8;
  const bool col_ok = rg < WB_RG && c0 < Np;
  T* Y = reinterpret_cast<T*>(a.y);
  const T* E1 = reinterpret_cast<const T*>(a.e1);
  float s0[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  float bias[8], eS[8], eB[8], eM[8], eR[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = col_ok && c0 + j < Np;
    bias[j] = (a.bias && ok && c0 + j < a.N) ? a.bias[c0 + j] : 0.f;
    if (EPI == C3D_EPI_SWISH_SE_BWD) {
      eS[j] = ok ? a.epi_p[c0 + j] : 1.f; eB[j] = ok ? a.epi_p[Np + c0 + j] : 0.f;
      eM[j] = ok ? a.epi_q[c0 + j] : 0.f; eR[j] = ok ? a.epi_q[Np + c0 + j] : 0.f;
    }
  }
  const int64_t n_first = m0 / rps;
  int cur_slot = -1;
  auto flush_slot = [&]() {
    if (cur_slot >= 0 && cur_slot < WB_SLOTS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned long long* d = red64 + ((size_t)cur_slot * WB_N + v * 8 + j) * 3;
        atomicAdd(d, (unsigned long long)__double2ll_rn((double)s0[j] * FIX));
        atomicAdd(d + 1, (unsigned long long)__double2ll_rn((double)s1[j] * FIX));
        atomicAdd(d + 2, (unsigned long long)__double2ll_rn((double)s2[j] * FIX));
        s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f;
      }
    }
  };
  if (col_ok) {
    for (int r = rg; r < WB_M; r += WB_RG) {
      const int64_t m = m0 + r;
      if (m >= a.M) break;
      float f[8];
      {
        const float4 p = *reinterpret_cast<const float4*>(Os + r * OL + v * 8);
        const float4 q = *reinterpret_cast<const float4*>(Os + r * OL + v * 8 + 4);
        f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w; f[4] = q.x; f[5] = q.y; f[6] = q.z; f[7] = q.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += bias[j];
      const int64_t yoff = m * Np + c0;
      if (EPI == C3D_EPI_STATS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float rr = round_as<T>(f[j]); s0[j] += rr; s1[j] += rr * rr; }
      } else if (EPI == C3D_EPI_SWISH_SE_BWD) {
        const int64_t n = m / rps;
        const int slot = (int)(n - n_first);
        if (slot != cur_slot) { flush_slot(); cur_slot = slot; }
        float bv[8];
        Vec8<T>::load(E1 + yoff, bv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = a.epi_gate ? a.epi_gate[n * Np + c0 + j] : 1.f;
          const float pb = fmaf(bv[j], eS[j], eB[j]);
          const float q = g * pb;
          const float sg = sigmoid_t<T>(q);
          const float dq = f[j] * sg * (1.f + q * (1.f - sg));
          const float t = round_as<T>(dq * g);
          s0[j] += dq * pb; s1[j] += t; s2[j] += t * ((bv[j] - eM[j]) * eR[j]);
          f[j] = t;
        }
      } else if (EPI == C3D_EPI_ADD) {
        if (a.res_mode == 0) {
          float rv[8];
          Vec8<T>::load(E1 + yoff, rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += rv[j];
        } else {
          const int64_t w_ = m % a.W, t_ = m / a.W, h_ = t_ % a.H, bt = t_ / a.H;
          if (((w_ | h_) & 1) == 0) {
            float rv[8];
            Vec8<T>::load(E1 + ((bt * (a.H >> 1) + (h_ >> 1)) * (a.W >> 1) + (w_ >> 1)) * Np + c0, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += rv[j];
          }
        }
      }
      Vec8<T>::store(Y + yoff, f);
    }
  }
  if (EPI == C3D_EPI_STATS) {
    __syncthreads();   // Os is dead: reuse it for the per-thread partial sums [18][112][2]
    float* part = Os;
    if (col_ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { part[((size_t)rg * WB_N + v * 8 + j) * 2] = s0[j]; part[((size_t)rg * WB_N + v * 8 + j) * 2 + 1] = s1[j]; }
    }
    __syncthreads();
    double* dst = a.stats + (size_t)(blockIdx.x % C3D_STAT_STRIPES) * 2 * a.N;
    for (int i = tid; i < WB_N * 2; i += 256) {
      const int c = i >> 1, which = i & 1;
      if (n0 + c < a.N) {
        float s = 0.f;
        for (int g = 0; g < WB_RG; ++g) s += part[((size_t)g * WB_N + c) * 2 + which];
        atomicAdd(dst + (size_t)which * a.N + n0 + c, (double)s);
      }
    }
  } else if (EPI == C3D_EPI_SWISH_SE_BWD) {
    flush_slot();
    __syncthreads();
    const int64_t nmax = (a.M - 1) / rps;
    for (int i = tid; i < WB_SLOTS * WB_N * 3; i += 256) {
      const int which = i % 3, c = (i / 3) % WB_N, slot = i / (3 * WB_N);
      const int64_t n = n_first + slot;
      const long long fx = (long long)red64[i];
      if (n <= nmax && n0 + c < Np && fx != 0) atomicAdd(a.stats + ((size_t)n * Np + n0 + c) * 3 + which, (double)fx * (1.0 / FIX));
    }
  }
}

template <typename T, int PRO, int EPI>
int launch_wide(const c3d_pw_args& a, hipStream_t st) {
  typedef Mma<T> MM;
  constexpr int KC = sizeof(T) == 2 ? 64 : 32;
  constexpr int KL = KC + MM::KPAD;
  const size_t lds = (size_t)(WB_M + WB_N) * KL * sizeof(typename MM::lds_t) + (size_t)WB_M * (WB_N + 4) * 4 +
                     (size_t)WB_SLOTS * WB_N * 3 * 8 + (size_t)3 * a.Kp * 4;
  dim3 grid((unsigned)((a.M + WB_M - 1) / WB_M), (unsigned)((a.Np + WB_N - 1) / WB_N));
  pw_wide_kernel<T, PRO, EPI><<<grid, 256, lds, st>>>(a);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int dispatch_wide(const c3d_pw_args& a, hipStream_t s) {
  const int pro = a.pro_mode, epi = a.epi_mode;
#define WIDE_CASE(P, E) if (pro == P && epi == E) return launch_wide<T, P, E>(a, s);
  WIDE_CASE(C3D_PRO_NONE, C3D_EPI_STORE)
  WIDE_CASE(C3D_PRO_NONE, C3D_EPI_STATS)
  WIDE_CASE(C3D_PRO_NONE, C3D_EPI_ADD)
  WIDE_CASE(C3D_PRO_BN_SE_SWISH, C3D_EPI_STORE)
  WIDE_CASE(C3D_PRO_BN_SE_SWISH, C3D_EPI_STATS)
  WIDE_CASE(C3D_PRO_AFFINE2, C3D_EPI_STORE)
  WIDE_CASE(C3D_PRO_AFFINE2, C3D_EPI_SWISH_SE_BWD)
  WIDE_CASE(C3D_PRO_AFFINE2, C3D_EPI_ADD)
#undef WIDE_CASE
  return C3D_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int WG_N = 64, WG_K = 64, WG_R = 32;   // dW block [64 n][64 k], 32 rows per LDS chunk

template <typename T>
__global__ __launch_bounds__(256) void pw_wide_wgrad_kernel(const c3d_pw_wgrad_args a, const int64_t rows_per_split) {
  typedef Mma<T> MM;
  typedef typename MM::lds_t lds_t;
  constexpr int RL = WG_R + MM::KPAD;
  constexpr int RS = WG_R / MM::KSTEP;
  __shared__ __attribute__((aligned(16))) lds_t Pt[WG_N * RL];   // [n][row]
  __shared__ __attribute__((aligned(16))) lds_t Qt[WG_K * RL];   // [k][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * WG_N, k0 = blockIdx.y * WG_K;
  const int64_t mlo = (int64_t)blockIdx.z * rows_per_split;
  const int64_t mhi = mlo + rows_per_split < a.M ? mlo + rows_per_split : a.M;
  const T* P = reinterpret_cast<const T*>(a.p);
  const T* P2 = reinterpret_cast<const T*>(a.p2);
  const T* Q = reinterpret_cast<const T*>(a.q);
  const int64_t rps = a.rows_per_sample > 0 ? a.rows_per_sample : a.M;
  f32x4_t acc[4];   // wave = n tile, 4 k tiles
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // 32 rows x 8 column vectors for each operand = 256 vectors: one of each per thread, written transposed.  The raw
  // vectors (and SE gates) of chunk c+1 are loaded into registers before the MFMAs of chunk c (software pipeline, as
  // in pw_wide_kernel).
  const int cv = tid & 7, r = tid >> 3;
  const int cn = n0 + cv * 8, ck = k0 + cv * 8;
  typedef typename Vec8<T>::raw_t raw_t;
  raw_t rp, rp2, rq;
  float4 rgt[2];
  bool p_ok = false, q_ok = false;
  int64_t q_n = 0;
  auto fetch = [&](const int64_t mb) {
    const int64_t m = mb + r;
    p_ok = m < mhi && cn < a.Np;
    q_ok = m < mhi && ck < a.Kp;
    if (p_ok) {
      rp = Vec8<T>::load_raw(P + m * a.Np + cn);
      if (a.p_coef || a.p_fin.sums) rp2 = Vec8<T>::load_raw(P2 + m * a.Np + cn);
    }
    if (q_ok) {
      rq = Vec8<T>::load_raw(Q + m * a.Kp + ck);
      if (a.q_mode == C3D_PRO_BN_SE_SWISH && a.q_gate) {
        q_n = m / rps;
        rgt[0] = *reinterpret_cast<const float4*>(a.q_gate + q_n * a.Kp + ck);
        rgt[1] = *reinterpret_cast<const float4*>(a.q_gate + q_n * a.Kp + ck + 4);
      }
    }
  };
  float pcA[8], pcB[8], pcC[8], qs[8], qh[8];   // per-channel coefficients of this thread's column vectors
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool pc = a.p_coef && cn < a.Np;
    pcA[j] = pc ? a.p_coef[cn + j] : 1.f; pcB[j] = pc ? a.p_coef[a.Np + cn + j] : 0.f; pcC[j] = pc ? a.p_coef[2 * a.Np + cn + j] : 0.f;
    // coefficients rebuilt from the producer's sums (csrc/bn_fin.h); nothing is accumulated here
    if (a.p_fin.sums && cn < a.Np) c3dfin::bn_bwd_coef_consume(a.p_fin, a.N, a.Np, cn + j, false, pcA[j], pcB[j], pcC[j]);
    const bool qc = a.q_mode == C3D_PRO_BN_SE_SWISH && ck < a.Kp;
    qs[j] = qc ? a.q_ss[ck + j] : 1.f; qh[j] = qc ? a.q_ss[a.Kp + ck + j] : 0.f;
  }
  fetch(mlo);
  for (int64_t mb = mlo; mb < mhi; mb += WG_R) {
    __syncthreads();
    {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      if (p_ok) {
        Vec8<T>::cvt_raw(rp, f);
        if (a.p_coef || a.p_fin.sums) {
          float f2[8];
          Vec8<T>::cvt_raw(rp2, f2);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaf(pcA[j], f[j], fmaf(pcC[j], f2[j], pcB[j]));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Pt[(cv * 8 + j) * RL + r] = MM::cvt(f[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      if (q_ok) {
        Vec8<T>::cvt_raw(rq, f);
        if (a.q_mode == C3D_PRO_BN_SE_SWISH) {
          const float g[8] = {rgt[0].x, rgt[0].y, rgt[0].z, rgt[0].w, rgt[1].x, rgt[1].y, rgt[1].z, rgt[1].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float qq = (a.q_gate ? g[j] : 1.f) * fmaf(f[j], qs[j], qh[j]);
            f[j] = qq * sigmoid_t<T>(qq);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Qt[(cv * 8 + j) * RL + r] = MM::cvt(f[j]);
    }
    __syncthreads();
    if (mb + WG_R < mhi) fetch(mb + WG_R);
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) {
      const typename MM::frag_t pa = MM::load(Pt, wave * 16 + (lane & 15), rs, RL, lane);   // A[i = n][k = row]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const typename MM::frag_t qb = MM::load(Qt, kt * 16 + (lane & 15), rs, RL, lane);   // B[k = row][j = k column]
        acc[kt] = MM::mma(pa, qb, acc[kt]);
      }
    }
  }
  // D[i = n = wave*16 + (lane>>4)*4 + r][j = k = kt*16 + (lane&15)]
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + kt * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave * 16 + (lane >> 4) * 4 + r;
      if (n < a.N && k < a.K) atomicAdd(a.dw + (size_t)n * a.dw_sn + (size_t)k * a.dw_sk, acc[kt][r]);
    }
  }
}

template <typename T>
int launch_wide_wgrad(const c3d_pw_wgrad_args& a, hipStream_t st) {
  const int gn = (a.Np + WG_N - 1) / WG_N, gk = (a.Kp + WG_K - 1) / WG_K;
  int64_t split = (2 * (int64_t)device_cus() + gn * gk - 1) / (gn * gk);
  const int64_t max_split = (a.M + 4 * WG_R - 1) / (4 * WG_R);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  int64_t rows = (a.M + split - 1) / split;
  rows = (rows + WG_R - 1) / WG_R * WG_R;
  split = (a.M + rows - 1) / rows;
  pw_wide_wgrad_kernel<T><<<dim3(gn, gk, (unsigned)split), 256, 0, st>>>(a, rows);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int c3d_detail_pw_gemm_wide(const c3d_pw_args* args, void* stream) {
  const c3d_pw_args& a = *args;
  if (a.row_mode != C3D_ROWS_DENSE && a.row_mode != C3D_ROWS_STRIDE2) return C3D_E_UNSUPPORTED;
  if (a.epi_mode == C3D_EPI_SWISH_SE_BWD && a.rows_per_sample < 16) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype == C3D_DT_F32) return dispatch_wide<float>(a, s);
  if (a.dtype == C3D_DT_BF16) return dispatch_wide<bf16_t>(a, s);
  return C3D_E_BADARG;
}

int c3d_detail_pw_wgrad_wide(const c3d_pw_wgrad_args* args, void* stream) {
  const c3d_pw_wgrad_args& a = *args;
  if (a.row_mode != C3D_ROWS_DENSE || a.taps > 1) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype == C3D_DT_F32) return launch_wide_wgrad<float>(a, s);
  if (a.dtype == C3D_DT_BF16) return launch_wide_wgrad<bf16_t>(a, s);
  return C3D_E_BADARG;
}
