// Pointwise-convolution row GEMM on MFMA (gfx950), forward and data-gradient.
//
//   Y[m, n] = epilogue( sum_k prologue(X)[m, k] * Wt[n, k] ),   M ~ 1e5..6e6, K,N <= 224.
//
// Design (HBM-bound op: ~10-40 flop/B, far under the MFMA ridge):
//  * every WAVE owns whole 16-row tiles end to end (load -> MFMA -> store); waves of a
//    workgroup share only the weight matrix, staged once into LDS, so the tile loop has no
//    workgroup barrier;
//  * a 16-row tile of a channels-last tensor is ONE contiguous 16*Kp-element span: it is
//    loaded with 16-byte lane vectors (lane <-> (row, 8-channel vector), the vector index
//    fixed per lane so the per-channel prologue parameters live in registers);
//  * MFMA roles are swapped (A = weights, B = data rows) so each lane ends up holding 4
//    consecutive output channels of one data row -> one 16-byte LDS write per 16x16 tile;
//  * the result tile is read back row-contiguous and stored with 16/32-byte lane vectors,
//    again with a fixed 8-channel vector per lane, which is where the fused epilogues
//    (BN statistics, Swish/SE backward, residual add) run and accumulate per-lane partial
//    sums that are flushed with f64 atomics once per wave (or per batch sample).
//  * bf16 storage -> v_mfma_f32_16x16x32_bf16; f32 storage -> v_mfma_f32_16x16x4_f32
//    (exact f32 fma chain, used by the parity path).
#include "common.h"
#include "../../include/change3d_hip.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16_t lds_t;
  static constexpr int KSTEP = 32;
  static constexpr int KPAD = 8;
  typedef uint4 frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int row, int ks, int kl, int lane) {
    return *reinterpret_cast<const uint4*>(base + row * kl + ks * 32 + (lane >> 4) * 8);
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<bf16_t>::store(p, f); }
  static __device__ __forceinline__ lds_t cvt(float f) { return f32_to_bf16(f); }
};
template <> struct Mma<float> {
  typedef float lds_t;
  static constexpr int KSTEP = 4;
  static constexpr int KPAD = 4;
  typedef float frag_t;
  static __device__ __forceinline__ frag_t load(const lds_t* base, int row, int ks, int kl, int lane) {
    return base[row * kl + ks * 4 + (lane >> 4)];
  }
  static __device__ __forceinline__ f32x4_t mma(frag_t a, frag_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void store8(lds_t* p, const float (&f)[8]) { Vec8<float>::store(p, f); }
  static __device__ __forceinline__ lds_t cvt(float f) { return f; }
};

__device__ __forceinline__ int64_t row_offset(const c3d_pw_args& a, int64_t m) {
  if (a.row_mode == C3D_ROWS_DENSE) return m * a.Kp;
  if (a.row_mode == C3D_ROWS_FRAME) {
    const uint32_t g = (uint32_t)m / (uint32_t)a.rpg;
    const uint32_t r = (uint32_t)m - g * (uint32_t)a.rpg;
    return (int64_t)g * a.gstride + (int64_t)r * a.Kp;
  }
  // STRIDE2: m = (bt, ho, wo) over output [BT][H/2][W/2]
  const uint32_t Wo = (uint32_t)a.W >> 1, Ho = (uint32_t)a.H >> 1;
  const uint32_t um = (uint32_t)m;
  const uint32_t wo = um % Wo;
  const uint32_t t = um / Wo;
  const uint32_t ho = t % Ho;
  const uint32_t bt = t / Ho;
  return (((int64_t)bt * a.H + 2 * ho) * a.W + 2 * wo) * a.Kp;
}

// Sum `v` over the lanes {lane, lane+G, lane+2G, ...} (result valid in lanes < G).
__device__ __forceinline__ float strided_lane_sum(float v, int lane, int G, int RP) {
  float s = v;
  for (int k = 1; k < RP; ++k) s += __shfl(v, lane + k * G, 64);
  return s;
}

template <typename T, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pw_gemm_kernel(const c3d_pw_args a, const int tiles_per_wave) {
  typedef Mma<T> MM;
  typedef typename MM::lds_t lds_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int Kp = a.Kp, Np = a.Np;
  const int Kpad = (Kp + MM::KSTEP - 1) / MM::KSTEP * MM::KSTEP;
  const int KL = Kpad + MM::KPAD;
  const int NL = NT * 16 + 4;
  const int KS = Kpad / MM::KSTEP;

  lds_t* Ws = reinterpret_cast<lds_t*>(smem);
  const size_t w_bytes = ((size_t)NT * 16 * KL * sizeof(lds_t) + 15) / 16 * 16;
  size_t wave_bytes = (size_t)16 * KL * sizeof(lds_t);
  if ((size_t)16 * NL * 4 > wave_bytes) wave_bytes = (size_t)16 * NL * 4;
  wave_bytes = (wave_bytes + 15) / 16 * 16;
  unsigned char* wreg = smem + w_bytes + (size_t)wave * wave_bytes;
  lds_t* Xs = reinterpret_cast<lds_t*>(wreg);
  float* Os = reinterpret_cast<float*>(wreg);  // aliases Xs (used strictly after the MFMAs)

  // ---- stage weights (zero padded) ------------------------------------------------------
  for (int n = wave; n < NT * 16; n += WAVES) {
    for (int k = lane; k < KL; k += 64) {
      float v = 0.f;
      if (n < a.N && k < a.K) v = a.w[(int64_t)n * a.w_sn + (int64_t)k * a.w_sk];
      Ws[n * KL + k] = MM::cvt(v);
    }
  }
  __syncthreads();

  // ---- lane <-> (row-in-pass, channel vector) maps ----------------------------------------
  const int Gi = Kp >> 3, RPi = 64 / Gi;
  const bool act_i = lane < Gi * RPi;
  const int rr_i = lane / Gi, v_i = lane - rr_i * Gi;
  const int Go = Np >> 3, RPo = 64 / Go;
  const bool act_o = lane < Go * RPo;
  const int rr_o = lane / Go, v_o = lane - rr_o * Go;

  // per-lane prologue parameters
  float pA[8], pB[8], pC[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { pA[j] = 1.f; pB[j] = 0.f; pC[j] = 0.f; }
  if (act_i && a.pro_mode != C3D_PRO_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pA[j] = a.pro_p[v_i * 8 + j];
      pB[j] = a.pro_p[Kp + v_i * 8 + j];
      if (a.pro_mode == C3D_PRO_AFFINE2) pC[j] = a.pro_p[2 * Kp + v_i * 8 + j];
    }
  }
  // per-lane epilogue parameters / accumulators
  float eS[8], eB[8], eM[8], eR[8];
  float s0[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { eS[j] = 1.f; eB[j] = 0.f; eM[j] = 0.f; eR[j] = 0.f; s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  if (act_o && a.epi_mode == C3D_EPI_SWISH_SE_BWD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      eS[j] = a.epi_p[v_o * 8 + j]; eB[j] = a.epi_p[Np + v_o * 8 + j];
      eM[j] = a.epi_q[v_o * 8 + j]; eR[j] = a.epi_q[Np + v_o * 8 + j];
    }
  }

  const int64_t tiles = (a.M + 15) >> 4;
  const int64_t gw = (int64_t)blockIdx.x * WAVES + wave;
  int64_t t0 = gw * tiles_per_wave;
  int64_t t1 = t0 + tiles_per_wave;
  if (t1 > tiles) t1 = tiles;
  int64_t cur_n = -1;  // sample whose partial sums are being accumulated (SWISH_SE_BWD)

  const T* X = reinterpret_cast<const T*>(a.x);
  const T* X2 = reinterpret_cast<const T*>(a.x2);
  const T* E1 = reinterpret_cast<const T*>(a.e1);
  T* Y = reinterpret_cast<T*>(a.y);

  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t row0 = tile << 4;
    // ---------------- load + prologue -> LDS ------------------------------------------------
    if (Kpad > Kp) {  // re-zero the K padding columns (region is aliased with the out tile)
      const int pv = (Kpad - Kp) >> 3;  // bf16 only (f32 has KSTEP 4 | Kp)
      for (int i = lane; i < 16 * pv; i += 64) {
        const int r = i / pv, c = i - r * pv;
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        MM::store8(Xs + r * KL + Kp + c * 8, z);
      }
    }
    for (int p = 0; p * RPi < 16; ++p) {
      const int row = p * RPi + rr_i;
      if (act_i && row < 16) {
        const int64_t m = row0 + row;
        float f[8];
        if (m < a.M) {
          const int64_t off = row_offset(a, m) + v_i * 8;
          Vec8<T>::load(X + off, f);
          if (a.pro_mode == C3D_PRO_BN_SE_SWISH) {
            float g[8];
            if (a.pro_gate) {
              const int64_t n = m / a.rows_per_sample;
              const float* gp = a.pro_gate + n * Kp + v_i * 8;
              const float4 g0 = *reinterpret_cast<const float4*>(gp);
              const float4 g1 = *reinterpret_cast<const float4*>(gp + 4);
              g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w;
              g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) g[j] = 1.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float q = g[j] * fmaf(f[j], pA[j], pB[j]);
              f[j] = q * sigmoid_t<T>(q);
            }
          } else if (a.pro_mode == C3D_PRO_AFFINE2) {
            float f2[8];
            Vec8<T>::load(X2 + off, f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(pA[j], f[j], fmaf(pC[j], f2[j], pB[j]));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        MM::store8(Xs + row * KL + v_i * 8, f);
      }
    }
    // ---------------- MFMA -----------------------------------------------------------------
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KS; ++ks) {
      const typename MM::frag_t xb = MM::load(Xs, lane & 15, ks, KL, lane);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const typename MM::frag_t wa = MM::load(Ws, nt * 16 + (lane & 15), ks, KL, lane);
        acc[nt] = MM::mma(wa, xb, acc[nt]);
      }
    }
    // ---------------- stage result tile: Os[row = lane&15][channel] ---------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      *reinterpret_cast<float4*>(Os + (lane & 15) * NL + nt * 16 + (lane >> 4) * 4) =
          make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    }
    // ---------------- epilogue + store -----------------------------------------------------
    if (a.epi_mode == C3D_EPI_SWISH_SE_BWD) {
      const int64_t n_tile = row0 / a.rows_per_sample;
      if (n_tile != cur_n) {
        if (cur_n >= 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float r0 = strided_lane_sum(s0[j], lane, Go, RPo);
            const float r1 = strided_lane_sum(s1[j], lane, Go, RPo);
            const float r2 = strided_lane_sum(s2[j], lane, Go, RPo);
            if (lane < Go) {
              double* d = a.stats + ((int64_t)cur_n * Np + v_o * 8 + j) * 3;
              atomicAdd(d, (double)r0); atomicAdd(d + 1, (double)r1); atomicAdd(d + 2, (double)r2);
            }
            s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f;
          }
        }
        cur_n = n_tile;
      }
    }
    for (int p = 0; p * RPo < 16; ++p) {
      const int row = p * RPo + rr_o;
      const int64_t m = row0 + row;
      if (act_o && row < 16 && m < a.M) {
        float f[8];
        {
          const float4 o0 = *reinterpret_cast<const float4*>(Os + row * NL + v_o * 8);
          const float4 o1 = *reinterpret_cast<const float4*>(Os + row * NL + v_o * 8 + 4);
          f[0] = o0.x; f[1] = o0.y; f[2] = o0.z; f[3] = o0.w; f[4] = o1.x; f[5] = o1.y; f[6] = o1.z; f[7] = o1.w;
        }
        const int64_t yoff = m * Np + v_o * 8;
        if (a.epi_mode == C3D_EPI_STATS) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float r = round_as<T>(f[j]);
            s0[j] += r; s1[j] += r * r;
          }
        } else if (a.epi_mode == C3D_EPI_SWISH_SE_BWD) {
          float bv[8], g[8];
          Vec8<T>::load(E1 + yoff, bv);
          if (a.epi_gate) {
            const float* gp = a.epi_gate + cur_n * Np + v_o * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gp);
            const float4 g1 = *reinterpret_cast<const float4*>(gp + 4);
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 1.f;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float pb = fmaf(bv[j], eS[j], eB[j]);
            const float q = g[j] * pb;
            const float sg = sigmoid_t<T>(q);
            const float dq = f[j] * sg * (1.f + q * (1.f - sg));
            const float t = round_as<T>(dq * g[j]);
            s0[j] += dq * pb;  // d gate
            s1[j] += t;        // sum t1
            s2[j] += t * ((bv[j] - eM[j]) * eR[j]);  // sum t1*bhat (centred)
            f[j] = t;
          }
        } else if (a.epi_mode == C3D_EPI_ADD) {
          if (a.res_mode == 0) {
            float rv[8];
            Vec8<T>::load(E1 + yoff, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += rv[j];
          } else {
            const uint32_t um = (uint32_t)m;
            const uint32_t w = um % (uint32_t)a.W;
            const uint32_t t = um / (uint32_t)a.W;
            const uint32_t h = t % (uint32_t)a.H;
            const uint32_t bt = t / (uint32_t)a.H;
            if (((w | h) & 1u) == 0u) {
              const int64_t roff = (((int64_t)bt * (a.H >> 1) + (h >> 1)) * (a.W >> 1) + (w >> 1)) * Np + v_o * 8;
              float rv[8];
              Vec8<T>::load(E1 + roff, rv);
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] += rv[j];
            }
          }
        }
        Vec8<T>::store(Y + yoff, f);
      }
    }
  }

  // ---- final flush of per-lane partial sums -------------------------------------------------
  if (a.epi_mode == C3D_EPI_STATS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r0 = strided_lane_sum(s0[j], lane, Go, RPo);
      const float r1 = strided_lane_sum(s1[j], lane, Go, RPo);
      const int c = v_o * 8 + j;
      if (lane < Go && c < a.N && t0 < t1) {
        atomicAdd(a.stats + c, (double)r0);
        atomicAdd(a.stats + a.N + c, (double)r1);
      }
    }
  } else if (a.epi_mode == C3D_EPI_SWISH_SE_BWD && cur_n >= 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r0 = strided_lane_sum(s0[j], lane, Go, RPo);
      const float r1 = strided_lane_sum(s1[j], lane, Go, RPo);
      const float r2 = strided_lane_sum(s2[j], lane, Go, RPo);
      if (lane < Go) {
        double* d = a.stats + ((int64_t)cur_n * Np + v_o * 8 + j) * 3;
        atomicAdd(d, (double)r0); atomicAdd(d + 1, (double)r1); atomicAdd(d + 2, (double)r2);
      }
    }
  }
}

int g_cus = 0;
int device_cus() {
  if (g_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return g_cus;
}

template <typename T, int NT>
int launch_pw(const c3d_pw_args& a, hipStream_t stream) {
  constexpr int WAVES = 4;
  typedef Mma<T> MM;
  const int Kpad = (a.Kp + MM::KSTEP - 1) / MM::KSTEP * MM::KSTEP;
  const int KL = Kpad + MM::KPAD;
  const int NL = NT * 16 + 4;
  const size_t w_bytes = ((size_t)NT * 16 * KL * sizeof(typename MM::lds_t) + 15) / 16 * 16;
  size_t wave_bytes = (size_t)16 * KL * sizeof(typename MM::lds_t);
  if ((size_t)16 * NL * 4 > wave_bytes) wave_bytes = (size_t)16 * NL * 4;
  wave_bytes = (wave_bytes + 15) / 16 * 16;
  const size_t lds = w_bytes + WAVES * wave_bytes;
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<T, NT, WAVES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t tiles = (a.M + 15) >> 4;
  int occ = (int)((160 * 1024) / lds);
  if (occ > 8) occ = 8;
  if (occ < 1) occ = 1;
  int64_t max_blocks = (int64_t)device_cus() * occ;
  int64_t blocks = (tiles + WAVES * 2 - 1) / (WAVES * 2);  // >= 2 tiles per wave to amortise W staging
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks * WAVES - 1) / (blocks * WAVES));
  pw_gemm_kernel<T, NT, WAVES><<<dim3((unsigned)blocks), dim3(WAVES * 64), lds, stream>>>(a, tpw);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int dispatch_nt(const c3d_pw_args& a, hipStream_t stream) {
  const int nt = (a.Np + 15) / 16;
  if (nt <= 2) return launch_pw<T, 2>(a, stream);
  if (nt <= 3) return launch_pw<T, 3>(a, stream);
  if (nt <= 4) return launch_pw<T, 4>(a, stream);
  if (nt <= 6) return launch_pw<T, 6>(a, stream);
  if (nt <= 7) return launch_pw<T, 7>(a, stream);
  if (nt <= 14) return launch_pw<T, 14>(a, stream);
  return C3D_E_UNSUPPORTED;
}


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
  static constexpr int MPAD = 8;
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

constexpr int WG_MT = 64;      // rows per tile
constexpr int WG_THREADS = 512;  // 8 waves: threads 0-255 stage P, 256-511 stage Q

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

template <typename T>
__global__ __launch_bounds__(WG_THREADS) void pw_wgrad_kernel(const c3d_pw_wgrad_args a, const int tiles_per_wg,
                                                              const int WN, const int WK) {
  typedef MmaT<T> MM;
  typedef typename MM::lds_t lds_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Kp = a.Kp, Np = a.Np;
  const int NT = (Np + 15) >> 4, KT = (Kp + 15) >> 4;
  const int ML = WG_MT + MM::MPAD;
  lds_t* PT = reinterpret_cast<lds_t*>(smem);
  lds_t* QT = PT + (size_t)NT * 16 * ML;

  // zero the channel-padding rows once
  for (int i = tid; i < (NT * 16 - Np) * ML; i += WG_THREADS) PT[(size_t)Np * ML + i] = (lds_t)0;
  for (int i = tid; i < (KT * 16 - Kp) * ML; i += WG_THREADS) QT[(size_t)Kp * ML + i] = (lds_t)0;

  const int Gp = Np >> 3, Gq = Kp >> 3;
  const int half = tid >> 8;              // 0: stages the P tile, 1: stages the Q tile
  const int rg = tid & 7, vv = (tid & 255) >> 3;  // task = (row group of 8 rows, channel vector)
  const bool p_act = half == 0 && vv < Gp, q_act = half == 1 && vv < Gq;
  float cA[8], cB[8], cC[8], qS[8], qB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cA[j] = 1.f; cB[j] = 0.f; cC[j] = 0.f; qS[j] = 1.f; qB[j] = 0.f; }
  if (p_act && a.p_coef) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cA[j] = a.p_coef[vv * 8 + j]; cB[j] = a.p_coef[Np + vv * 8 + j]; cC[j] = a.p_coef[2 * Np + vv * 8 + j];
    }
  }
  if (q_act && a.q_mode == C3D_PRO_BN_SE_SWISH) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { qS[j] = a.q_ss[vv * 8 + j]; qB[j] = a.q_ss[Kp + vv * 8 + j]; }
  }

  const int wn_i = wave % WN, wk_i = wave / WN;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const T* P = reinterpret_cast<const T*>(a.p);
  const T* P2 = reinterpret_cast<const T*>(a.p2);
  const T* Q = reinterpret_cast<const T*>(a.q);
  const int64_t tiles = (a.M + WG_MT - 1) / WG_MT;
  int64_t t0 = (int64_t)blockIdx.x * tiles_per_wg, t1 = t0 + tiles_per_wg;
  if (t1 > tiles) t1 = tiles;

  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t row0 = tile * WG_MT + rg * 8;
    __syncthreads();  // previous tile's MFMA reads done (and the zero fill on the first trip)
    if (p_act) {
      float o[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t m = row0 + r;
        if (m < a.M) {
          float f[8];
          Vec8<T>::load(P + m * Np + vv * 8, f);
          if (a.p_coef) {
            float f2[8];
            Vec8<T>::load(P2 + m * Np + vv * 8, f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(cA[j], f[j], fmaf(cC[j], f2[j], cB[j]));
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) o[r][j] = f[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[r][j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float col[8] = {o[0][j], o[1][j], o[2][j], o[3][j], o[4][j], o[5][j], o[6][j], o[7][j]};
        MM::store8(PT + (size_t)(vv * 8 + j) * ML + rg * 8, col);
      }
    }
    if (q_act) {
      float o[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t m = row0 + r;
        const int64_t qoff = (m < a.M) ? q_row_offset(a, m) : -1;
        if (qoff >= 0) {
          float f[8];
          Vec8<T>::load(Q + qoff + vv * 8, f);
          if (a.q_mode == C3D_PRO_BN_SE_SWISH) {
            float g[8];
            if (a.q_gate) {
              const int64_t n = m / a.rows_per_sample;
              const float* gp = a.q_gate + n * Kp + vv * 8;
#pragma unroll
              for (int j = 0; j < 8; ++j) g[j] = gp[j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) g[j] = 1.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float qv = g[j] * fmaf(f[j], qS[j], qB[j]);
              f[j] = qv * sigmoid_t<T>(qv);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) o[r][j] = f[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[r][j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float col[8] = {o[0][j], o[1][j], o[2][j], o[3][j], o[4][j], o[5][j], o[6][j], o[7][j]};
        MM::store8(QT + (size_t)(vv * 8 + j) * ML + rg * 8, col);
      }
    }
    __syncthreads();
    for (int ks = 0; ks < WG_MT / MM::KSTEP; ++ks) {
      typename MM::frag_t pa[4], qb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nt = wn_i + i * WN;
        if (nt < NT) pa[i] = MM::load(PT, nt * 16 + (lane & 15), ks, ML, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kt = wk_i + j * WK;
        if (kt < KT) qb[j] = MM::load(QT, kt * 16 + (lane & 15), ks, ML, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wn_i + i * WN < NT) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (wk_i + j * WK < KT) acc[i][j] = MM::mma(pa[i], qb[j], acc[i][j]);
          }
        }
      }
    }
  }

  // partials -> workspace [grid][N][K]
  float* wsb = a.ws + (size_t)blockIdx.x * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nt = wn_i + i * WN;
    if (nt >= NT) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kt = wk_i + j * WK;
      if (kt >= KT) continue;
      const int k = kt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + (lane >> 4) * 4 + r;
        if (n < a.N && k < a.K) wsb[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
  }
}

__global__ void pw_wgrad_reduce_kernel(const float* __restrict__ ws, float* dw, int N, int K, int parts,
                                       int sn, int sk) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * K) return;
  float s = 0.f;
  for (int p = 0; p < parts; ++p) s += ws[(size_t)p * N * K + idx];
  const int n = idx / K, k = idx - n * K;
  dw[(size_t)n * sn + (size_t)k * sk] += s;
}

constexpr int WGRAD_MAX_PARTS = 512;

template <typename T>
int launch_wgrad(const c3d_pw_wgrad_args& a, hipStream_t stream) {
  typedef MmaT<T> MM;
  const int NT = (a.Np + 15) >> 4, KT = (a.Kp + 15) >> 4;
  const int ML = WG_MT + MM::MPAD;
  const size_t lds = (size_t)(NT + KT) * 16 * ML * sizeof(typename MM::lds_t);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  // wave grid: WN*WK = 8 with ceil(NT/WN) <= 4 and ceil(KT/WK) <= 4
  int WN = 0, WK = 0;
  const int cand[4][2] = {{8, 1}, {4, 2}, {2, 4}, {1, 8}};
  int best = 1 << 30;
  for (int c = 0; c < 4; ++c) {
    const int tn = (NT + cand[c][0] - 1) / cand[c][0], tk = (KT + cand[c][1] - 1) / cand[c][1];
    if (tn > 4 || tk > 4) continue;
    const int cost = tn * tk * 4 + tn + tk;  // MFMAs dominate, then fragment loads
    if (cost < best) { best = cost; WN = cand[c][0]; WK = cand[c][1]; }
  }
  if (WN == 0) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t tiles = (a.M + WG_MT - 1) / WG_MT;
  int64_t blocks = (tiles + 1) / 2;
  const int64_t cap = (int64_t)device_cus() * 2 < WGRAD_MAX_PARTS ? (int64_t)device_cus() * 2 : WGRAD_MAX_PARTS;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int tpw = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpw - 1) / tpw;
  pw_wgrad_kernel<T><<<dim3((unsigned)blocks), dim3(WG_THREADS), lds, stream>>>(a, tpw, WN, WK);
  C3D_CHECK_LAUNCH();
  const int nk = a.N * a.K;
  pw_wgrad_reduce_kernel<<<dim3((nk + 255) / 256), dim3(256), 0, stream>>>(a.ws, a.dw, a.N, a.K, (int)blocks,
                                                                            a.dw_sn, a.dw_sk);
  C3D_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int c3d_device_cus(void) { return device_cus(); }

extern "C" int c3d_pw_gemm(const c3d_pw_args* args, void* stream) {
  if (!args || !args->x || !args->y || !args->w) return C3D_E_BADARG;
  const c3d_pw_args& a = *args;
  if (a.M <= 0 || (a.Kp & 7) || (a.Np & 7) || a.K > a.Kp || a.N > a.Np || a.Kp > 224 || a.Np > 224)
    return C3D_E_BADARG;
  if (a.pro_mode == C3D_PRO_AFFINE2 && !a.x2) return C3D_E_BADARG;
  if (a.pro_mode != C3D_PRO_NONE && !a.pro_p) return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_STATS && !a.stats) return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_SWISH_SE_BWD &&
      (!a.stats || !a.e1 || !a.epi_p || !a.epi_q || a.rows_per_sample <= 0 || (a.rows_per_sample & 15)))
    return C3D_E_BADARG;
  if (a.epi_mode == C3D_EPI_ADD && !a.e1) return C3D_E_BADARG;
  if (a.pro_mode == C3D_PRO_BN_SE_SWISH && a.pro_gate && a.rows_per_sample <= 0) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype == C3D_DT_F32) return dispatch_nt<float>(a, s);
  if (a.dtype == C3D_DT_BF16) return dispatch_nt<bf16_t>(a, s);
  return C3D_E_BADARG;
}

extern "C" int64_t c3d_pw_wgrad_ws_floats(int32_t N, int32_t K) { return (int64_t)WGRAD_MAX_PARTS * N * K; }

extern "C" int c3d_pw_wgrad(const c3d_pw_wgrad_args* args, void* stream) {
  if (!args || !args->p || !args->q || !args->dw || !args->ws) return C3D_E_BADARG;
  const c3d_pw_wgrad_args& a = *args;
  if (a.M <= 0 || (a.Kp & 7) || (a.Np & 7) || a.K > a.Kp || a.N > a.Np || a.Kp > 224 || a.Np > 224)
    return C3D_E_BADARG;
  if (a.p_coef && !a.p2) return C3D_E_BADARG;
  if (a.q_mode == C3D_PRO_BN_SE_SWISH && (!a.q_ss || (a.q_gate && a.rows_per_sample <= 0))) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype == C3D_DT_F32) return launch_wgrad<float>(a, s);
  if (a.dtype == C3D_DT_BF16) return launch_wgrad<bf16_t>(a, s);
  return C3D_E_BADARG;
}
