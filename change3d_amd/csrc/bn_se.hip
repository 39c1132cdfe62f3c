// BatchNorm3d statistics finalisation, SqueezeExcitation FCs and the matching backward
// coefficient kernels.  All of these touch only O(B*C) numbers: one small workgroup each,
// f64 arithmetic for the statistics (torch-CPU accumulates BN statistics in double too).
//
// Train-mode BN is split three ways in this framework:
//   producer epilogue  : per-channel (or per-sample-per-channel) sum / sum-of-squares
//   c3d_bn_finalize    : -> scale/shift (+ running-stat update, saved mean/rstd)
//   consumer prologue  : y = x*scale + shift applied on operand load
// and backward mirrors it (sums in a producer epilogue -> coefficients here -> affine
// dx = A*g + B + C*x applied on operand load of the next kernel).
#include "common.h"
#include "../../include/change3d_hip.h"

namespace {

// -------------------------------------------------------------------------------------------
// sums: [2][C] (sum, sumsq) per channel.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int stripes, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   int64_t* nbt, float momentum, float eps, int C, int Cp, int training,
                                   float* __restrict__ ss, float* __restrict__ mr) {
  // 16 lanes per channel: each reads one stripe, fixed-order butterfly (the one-thread-per-channel loop was 16
  // dependent L2 round trips -- 9.6 us for a kernel on the critical path of every BN layer)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 4, kq = gid & 15;
  if (gid == 0 && training && nbt) *nbt += 1;
  const bool live = c < C;
  double s1 = 0, s2 = 0;
  if (training && live) {
    for (int k = kq; k < stripes; k += 16) { s1 += sums[(size_t)k * 2 * C + c]; s2 += sums[(size_t)k * 2 * C + C + c]; }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (kq != 0 || c >= Cp) return;
  if (c >= C) { ss[c] = 0.f; ss[Cp + c] = 0.f; if (mr) { mr[c] = 0.f; mr[Cp + c] = 0.f; } return; }
  double mean, var;
  if (training) {
    mean = s1 / count;
    var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    if (running_mean) {
      const double unb = count > 1 ? var * count / (count - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  const float sc = gamma[c] * rstd;
  ss[c] = sc;
  ss[Cp + c] = beta[c] - meanf * sc;
  if (mr) { mr[c] = meanf; mr[Cp + c] = rstd; }
}

constexpr int SE_THREADS = 1024;  // layer kernels: width buys latency
constexpr int SE_SLICES = 8;      // channel slices (workgroups) of the SE layer kernels

// -------------------------------------------------------------------------------------------
// Depthwise-conv output statistics arrive per (sample, channel): nc[B][Cp][2].
// Produces BN_b scale/shift (+running stats), and the SE gate[B][Cp].
__global__ __launch_bounds__(SE_THREADS) void bn_se_finalize_kernel(
    const double* __restrict__ nc, int B, double cnt_per_sample, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
    float eps, int C, int Cp, int training, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, int Cr, float* __restrict__ ss,
    float* __restrict__ mr, float* __restrict__ gate, float* __restrict__ hid) {
  extern __shared__ float sm[];  // z[B][C] then h[B][Cr]
  float* z = sm;
  float* h = sm + (size_t)B * C;
  const int Cw = C + ((8 - (C & 31)) & 31);   // w1 row stride in LDS = 8 (mod 32 banks): rows r, r+1, .. start 8 banks apart
  float* w1s = h + (size_t)B * Cr;      // [Cr][Cw] FC weights staged once, coalesced (the FC loops read them
  float* w2s = w1s + (size_t)Cr * Cw;   // [C][Cr]   element by element from global memory before)
  const int tid = threadIdx.x;
  if (w1) {
    for (int i = tid; i < Cr * C; i += blockDim.x) { w1s[(i / C) * Cw + i % C] = w1[i]; w2s[i] = w2[i]; }
  }
  const double count = cnt_per_sample * B;
  // gridDim.x workgroups each recompute the cheap whole-layer parts (statistics, FC1) and OWN a
  // channel slice [c_lo, c_hi) of everything that is written (running statistics, scale/shift, gate)
  const int cs = (Cp + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * cs, c_hi = c_lo + cs < Cp ? c_lo + cs : Cp;
  if (tid == 0 && blockIdx.x == 0 && training && nbt) *nbt += 1;
  // 4 lanes per channel split the batch loop (latency, not bandwidth, is what this kernel costs, so
  // every serial loop is spread over adjacent lanes)
  for (int idx = tid; idx < Cp * 4; idx += blockDim.x) {
    const int c = idx >> 2, q = idx & 3;
    const bool own = c >= c_lo && c < c_hi;
    if (c >= C) { if (q == 0 && own) { ss[c] = 0.f; ss[Cp + c] = 0.f; if (mr) { mr[c] = 0.f; mr[Cp + c] = 0.f; } } continue; }
    double mean, var;
    if (training == 2) {   // folded BatchNorm (eval): scale / shift are given, only the SE gate is computed
      const float sc2 = ss[c], sh2 = ss[Cp + c];
      if (w1) {
#pragma unroll 4
        for (int n = q; n < B; n += 4)
          z[(size_t)n * C + c] = fmaf(sc2, (float)(nc[((size_t)n * Cp + c) * 2] / cnt_per_sample), sh2);
      }
      continue;
    }
    if (training) {
      double s1 = 0, s2 = 0;
#pragma unroll 4
      for (int n = q; n < B; n += 4) { s1 += nc[((size_t)n * Cp + c) * 2]; s2 += nc[((size_t)n * Cp + c) * 2 + 1]; }
      s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
      s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
      mean = s1 / count;
      var = s2 / count - mean * mean;
      if (var < 0) var = 0;
      if (running_mean && q == 0 && own && training == 1) {
        const double unb = count > 1 ? var * count / (count - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
      }
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    const float sc = gamma[c] * rstd;
    const float sh = beta[c] - meanf * sc;
    if (q == 0 && own) {
      ss[c] = sc;
      ss[Cp + c] = sh;
      if (mr) { mr[c] = meanf; mr[Cp + c] = rstd; }
    }
    if (w1) {
#pragma unroll 4
      for (int n = q; n < B; n += 4)
        z[(size_t)n * C + c] = fmaf(sc, (float)(nc[((size_t)n * Cp + c) * 2] / cnt_per_sample), sh);
    }
  }
  if (!w1) return;  // no SE in this block: consumers take gate = 1 (NULL)
  __syncthreads();
  // FC1: eight partial sums per (sample, hidden unit) over c = q, q + 8, ...; a thread carries 4 samples, a wave's lanes differ in
  // q (fastest) and r: consecutive z addresses, w1 rows of distinct r.  (With 8 adjacent lanes = the 8 partial sums of one
  // (n, r) the w1 reads of a wave fell on a quarter of the banks -- see se_bn_bwd_coef_kernel.)  The partial sums are combined
  // through LDS in the association of the xor-shuffle tree they replace: bit-identical to it and to se_gate_consume (bn_fin.h).
  float* part = w2s + (size_t)Cr * C;   // [8][B * Cr]
  const int BR = B * Cr, NB4 = (B + 3) >> 2;
  for (int idx = tid; idx < 8 * NB4 * Cr; idx += blockDim.x) {
    const int q = idx & 7, j = idx >> 3;
    const int r = j % Cr, nb = j / Cr;
    const int n0 = 4 * nb;
    const float* z0 = z + (size_t)(n0 < B ? n0 : B - 1) * C;
    const float* z1 = z + (size_t)(n0 + 1 < B ? n0 + 1 : B - 1) * C;
    const float* z2 = z + (size_t)(n0 + 2 < B ? n0 + 2 : B - 1) * C;
    const float* z3 = z + (size_t)(n0 + 3 < B ? n0 + 3 : B - 1) * C;
    const float* wr = w1s + (size_t)r * Cw;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = q;
    for (; c + 24 < C; c += 32) {   // (four steps' reads in flight; every FMA chain keeps its order)
      float wv[4], x0[4], x1[4], x2[4], x3[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { wv[u] = wr[c + 8 * u]; x0[u] = z0[c + 8 * u]; x1[u] = z1[c + 8 * u]; x2[u] = z2[c + 8 * u]; x3[u] = z3[c + 8 * u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0 = fmaf(wv[u], x0[u], a0); a1 = fmaf(wv[u], x1[u], a1); a2 = fmaf(wv[u], x2[u], a2); a3 = fmaf(wv[u], x3[u], a3); }
    }
    for (; c < C; c += 8) {
      const float w_ = wr[c];
      a0 = fmaf(w_, z0[c], a0); a1 = fmaf(w_, z1[c], a1); a2 = fmaf(w_, z2[c], a2); a3 = fmaf(w_, z3[c], a3);
    }
    float* pq = part + (size_t)q * BR + r;
    if (n0 < B) pq[(size_t)n0 * Cr] = a0;
    if (n0 + 1 < B) pq[(size_t)(n0 + 1) * Cr] = a1;
    if (n0 + 2 < B) pq[(size_t)(n0 + 2) * Cr] = a2;
    if (n0 + 3 < B) pq[(size_t)(n0 + 3) * Cr] = a3;
  }
  __syncthreads();
  for (int i = tid; i < BR; i += blockDim.x) {
    const int r = i % Cr;
    float a = ((part[i] + part[BR + i]) + (part[2 * BR + i] + part[3 * BR + i])) +
              ((part[4 * BR + i] + part[5 * BR + i]) + (part[6 * BR + i] + part[7 * BR + i]));
    a += b1[r];
    a = a > 0.f ? a : 0.f;
    h[i] = a;
    if (hid && blockIdx.x == 0) hid[i] = a;
  }
  __syncthreads();
  const int cw = c_hi - c_lo;
  for (int i = tid; i < B * cw; i += blockDim.x) {
    const int n = i / cw, c = c_lo + (i - n * cw);
    float g = 0.f;
    if (c < C) {
      float a = b2[c];
      for (int r = 0; r < Cr; ++r) a = fmaf(w2s[(size_t)c * Cr + r], h[(size_t)n * Cr + r], a);
      g = 1.0f / (1.0f + expf(-a));
    }
    gate[(size_t)n * Cp + c] = g;
  }
}

// -------------------------------------------------------------------------------------------
// BN backward coefficients: dsums [2][C] = (sum g, sum g*xhat), xhat = (x-mean)*rstd accumulated
// CENTRED by the producer (as ATen does; avoids the sum(g*x) - mean*sum(g) cancellation).
// dx = A*g + B + C*x.
__global__ void bn_bwd_coef_kernel(const double* __restrict__ dsums, int stripes, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ mr, int C, int Cp, float* __restrict__ coef,
                                   float* dgamma, float* dbeta) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;   // 16 lanes per channel, one stripe each
  const int c = gid >> 4, kq = gid & 15;
  double s1 = 0, s2 = 0;  // sum g, sum g * xhat (over the striped accumulator sets)
  if (c < C) {
    for (int k = kq; k < stripes; k += 16) { s1 += dsums[(size_t)k * 2 * C + c]; s2 += dsums[(size_t)k * 2 * C + C + c]; }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (kq != 0 || c >= Cp) return;
  if (c >= C) { coef[c] = 0.f; coef[Cp + c] = 0.f; coef[2 * Cp + c] = 0.f; return; }
  const double mean = mr[c], rstd = mr[Cp + c];
  const double A = (double)gamma[c] * rstd;
  const double Cc = -A * rstd * s2 / count;
  const double Bc = -A * s1 / count - Cc * mean;
  coef[c] = (float)A;
  coef[Cp + c] = (float)Bc;
  coef[2 * Cp + c] = (float)Cc;
  if (dgamma) dgamma[c] += (float)s2;
  if (dbeta) dbeta[c] += (float)s1;
}

#ifdef C3D_SE_CLOCK   // tools/se_phase_clock.py: s_memtime stamps of workgroup 0 / thread 0 around the phases below
__device__ unsigned long long c3d_se_clk[16];
#define SECLK(i) { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); c3d_se_clk[i] += t_ - se_t_last; se_t_last = t_; } }
#else
#define SECLK(i)
#endif

// -------------------------------------------------------------------------------------------
// SE backward + BN_b backward coefficients.
//   nc3 [B][Cp][3] = per (n,c): sum dq*pb (d gate), sum t1, sum t1*bhat   (bhat = (b-mean)*rstd)
//   ncf [B][Cp][2] = forward per (n,c): sum b, sum b^2
//   db = A[c]*t1 + Bnc[n][c] + Cc[c]*b
__global__ __launch_bounds__(SE_THREADS) void se_bn_bwd_coef_kernel(
    const double* __restrict__ nc3, const double* __restrict__ ncf, int B, double cnt_per_sample,
    const float* __restrict__ gamma, const float* __restrict__ mr, const float* __restrict__ ss, int C, int Cp,
    const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ gate,
    const float* __restrict__ hid, int Cr, float* __restrict__ coefA, float* __restrict__ coefC,
    float* __restrict__ coefB, float* dgamma, float* dbeta, float* dw1, float* db1, float* dw2, float* db2) {
  extern __shared__ float sm[];
  float* du = sm;                      // [B][C]
  float* dh = du + (size_t)B * C;      // [B][Cr]
  float* dz = dh + (size_t)B * Cr;     // [B][C]
  float* zz = dz + (size_t)B * C;      // [B][C]  forward z (input of the SE FCs)
  // FC weights and the hidden activations are staged once, coalesced: the FC loops below used to read them from
  // global memory element by element (chains of dependent L2 round trips in a single-digit-workgroup kernel)
  float* wst = zz + (size_t)B * C;     // [Cr*C]  ONE staged FC weight matrix at a time: w2 for the hidden gradient, then
                                       //          w1 for dz (both at once did not fit LDS for res5: C=432, Cr=32, B=16)
  float* hids = wst + (size_t)C * Cr;  // [B][Cr]
  const int tid = threadIdx.x;
#ifdef C3D_SE_CLOCK
  unsigned long long se_t_last = __builtin_amdgcn_s_memtime();
#endif
  const double count = cnt_per_sample * B;
  const double inv_cps = 1.0 / cnt_per_sample;   // (SE terms only: an f64 division per sample and lane was a third of the last phase)
  const bool se = w1 != nullptr;
  // channel-sliced over gridDim.x workgroups: d gate and the FC2-backward hidden gradient need every
  // channel and are recomputed by each workgroup; everything written is owned by slice [c_lo, c_hi)
  const int cs = (Cp + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * cs, c_hi = c_lo + cs < Cp ? c_lo + cs : Cp;
  const int cw = c_hi - c_lo;
  // The BatchNorm coefficient phase at the end reads 3 f64 per (sample, channel) of this slice from global memory: requested here,
  // consumed ~15 us later (4 lanes per channel, <= PF samples per lane; other shapes load in the loop)
  constexpr int PF = 8;
  double pf1[PF], pf2[PF], pff[PF];
  const bool pre = se && cw * 4 <= (int)blockDim.x && B <= 4 * PF;
  if (pre && tid < cw * 4 && c_lo + (tid >> 2) < C) {
    const int c = c_lo + (tid >> 2), q = tid & 3;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int n = q + 4 * k;
      if (n < B) {
        pf1[k] = nc3[((size_t)n * Cp + c) * 3 + 1]; pf2[k] = nc3[((size_t)n * Cp + c) * 3 + 2]; pff[k] = ncf[((size_t)n * Cp + c) * 2];
      }
    }
  }
  if (se) {
    for (int i = tid; i < Cr * C; i += blockDim.x) wst[i] = w2[i];   // [C][Cr]
    for (int i = tid; i < B * Cr; i += blockDim.x) hids[i] = hid[i];
#pragma unroll 4
    for (int i = tid; i < B * C; i += blockDim.x) {
      const int n = i / C, c = i - n * C;
      const float g = gate[(size_t)n * Cp + c];
      du[i] = (float)nc3[((size_t)n * Cp + c) * 3] * g * (1.f - g);
      zz[i] = fmaf(ss[c], (float)(ncf[((size_t)n * Cp + c) * 2] / cnt_per_sample), ss[Cp + c]);
    }
    SECLK(0)
    __syncthreads();
    SECLK(1)
    // dh[n][r] = relu'(hid) sum_c w2[c][r] du[n][c]: eight partial sums per (n, r) over c = q, q + 8, ...  Lane index = r fastest:
    // a wave reads 16 / 32 CONSECUTIVE w2 values and broadcasts du (the first mapping -- 8 adjacent lanes = the 8 partial sums --
    // put 4 lanes on every LDS bank it touched: the phase was 40 % of the res4 launch, 60 % of res5's).  The partial sums go
    // through LDS (the dz region, not written yet) and are combined in the association of the xor-shuffle tree they replace.
    float* part = 8 * Cr <= C ? dz : hids + (size_t)B * Cr;   // [8][B * Cr]: inside dz where it fits, else its own region (launcher)
    const int BR = B * Cr;
    const int NB4 = (B + 3) >> 2;                     // a thread carries 4 samples: one w2 read serves 4 FMAs (the phase is LDS-bound)
    for (int idx = tid; idx < 8 * NB4 * Cr; idx += blockDim.x) {
      // idx = (nb * 8 + q) * Cr + r: a wave's lanes differ in r and q -- consecutive w2 addresses, du values of one sample row
      // (with the sample group as the middle index the 4 rows of a wave were 4 x 216 floats apart: one bank)
      const int r = idx % Cr, j = idx / Cr;
      const int q = j & 7, nb = j >> 3;
      const int n0 = 4 * nb;
      // (samples beyond B read sample B-1 again; their sums are not stored)
      const float* d0 = du + (size_t)(n0 < B ? n0 : B - 1) * C;
      const float* d1 = du + (size_t)(n0 + 1 < B ? n0 + 1 : B - 1) * C;
      const float* d2 = du + (size_t)(n0 + 2 < B ? n0 + 2 : B - 1) * C;
      const float* d3 = du + (size_t)(n0 + 3 < B ? n0 + 3 : B - 1) * C;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int c = q;
      for (; c + 24 < C; c += 32) {   // (four steps' reads in flight; every FMA chain keeps its order)
        float wv[4], x0[4], x1[4], x2[4], x3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          wv[u] = wst[(size_t)(c + 8 * u) * Cr + r];
          x0[u] = d0[c + 8 * u]; x1[u] = d1[c + 8 * u]; x2[u] = d2[c + 8 * u]; x3[u] = d3[c + 8 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 = fmaf(wv[u], x0[u], a0); a1 = fmaf(wv[u], x1[u], a1); a2 = fmaf(wv[u], x2[u], a2); a3 = fmaf(wv[u], x3[u], a3); }
      }
      for (; c < C; c += 8) {
        const float w_ = wst[(size_t)c * Cr + r];
        a0 = fmaf(w_, d0[c], a0); a1 = fmaf(w_, d1[c], a1); a2 = fmaf(w_, d2[c], a2); a3 = fmaf(w_, d3[c], a3);
      }
      float* pq = part + (size_t)q * BR + r;
      if (n0 < B) pq[(size_t)n0 * Cr] = a0;
      if (n0 + 1 < B) pq[(size_t)(n0 + 1) * Cr] = a1;
      if (n0 + 2 < B) pq[(size_t)(n0 + 2) * Cr] = a2;
      if (n0 + 3 < B) pq[(size_t)(n0 + 3) * Cr] = a3;
    }
    __syncthreads();
    for (int i = tid; i < BR; i += blockDim.x) {
      const float a = ((part[i] + part[BR + i]) + (part[2 * BR + i] + part[3 * BR + i])) +
                      ((part[4 * BR + i] + part[5 * BR + i]) + (part[6 * BR + i] + part[7 * BR + i]));
      dh[i] = hids[i] > 0.f ? a : 0.f;
    }
    SECLK(2)
    __syncthreads();
    for (int i = tid; i < Cr * C; i += blockDim.x) wst[i] = w1[i];   // [Cr][C]
    __syncthreads();
    SECLK(3)
    for (int i = tid; i < B * cw; i += blockDim.x) {
      const int n = i / cw, c = c_lo + (i - n * cw);
      if (c >= C) continue;
      float a = 0.f;
      for (int r = 0; r < Cr; ++r) a = fmaf(wst[(size_t)r * C + c], dh[(size_t)n * Cr + r], a);
      dz[(size_t)n * C + c] = a;
    }
    SECLK(4)
    // parameter gradients of the two FCs
    for (int i = tid; i < cw * Cr; i += blockDim.x) {
      const int c = c_lo + i / Cr, r = i % Cr;  // w2[c][r]
      if (c >= C) continue;
      float a = 0.f, b = 0.f;
      int n = 0;
      for (; n + 4 <= B; n += 4) {   // (reads of four samples in flight; the two FMA chains keep their order)
        float x0[4], x1[4], x2[4], x3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x0[u] = du[(size_t)(n + u) * C + c]; x1[u] = hids[(size_t)(n + u) * Cr + r];
          x2[u] = dh[(size_t)(n + u) * Cr + r]; x3[u] = zz[(size_t)(n + u) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a = fmaf(x0[u], x1[u], a); b = fmaf(x2[u], x3[u], b); }
      }
      for (; n < B; ++n) {
        a = fmaf(du[(size_t)n * C + c], hids[(size_t)n * Cr + r], a);
        b = fmaf(dh[(size_t)n * Cr + r], zz[(size_t)n * C + c], b);
      }
      // (one owner per address: a no-return atomic is the same sum without the load -> add -> store round trip)
      atomicAdd(&dw2[(size_t)c * Cr + r], a);
      atomicAdd(&dw1[(size_t)r * C + c], b);
    }
    for (int c = c_lo + tid; c < c_hi && c < C; c += blockDim.x) {
      float a = 0.f;
      for (int n = 0; n < B; ++n) a += du[(size_t)n * C + c];
      atomicAdd(&db2[c], a);
    }
    if (blockIdx.x == 0) {
      for (int r = tid; r < Cr; r += blockDim.x) {
        float a = 0.f;
        for (int n = 0; n < B; ++n) a += dh[(size_t)n * Cr + r];
        atomicAdd(&db1[r], a);
      }
    }
    SECLK(5)
    __syncthreads();
    SECLK(6)
  }
  for (int idx = tid; idx < cw * 4; idx += blockDim.x) {  // 4 lanes per channel split the batch loops
    const int c = c_lo + (idx >> 2), q = idx & 3;
    if (c >= C) {
      if (q == 0) { coefA[c] = 0.f; coefC[c] = 0.f; }
      for (int n = q; n < B; n += 4) coefB[(size_t)n * Cp + c] = 0.f;
      continue;
    }
    const double mean = mr[c], rstd = mr[Cp + c];
    double s1 = 0, s2 = 0;
    if (pre) {   // (idx == tid here: one pass)
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int n = q + 4 * k;
        if (n < B) {
          const double dzn = (double)dz[(size_t)n * C + c];
          s1 += pf1[k] + dzn;
          s2 += pf2[k] + dzn * inv_cps * rstd * (pff[k] - cnt_per_sample * mean);
        }
      }
    } else {
#pragma unroll 4
    for (int n = q; n < B; n += 4) {
      const double dzn = se ? (double)dz[(size_t)n * C + c] : 0.0;
      s1 += nc3[((size_t)n * Cp + c) * 3 + 1] + dzn;
      // the uniform SE term dz/cnt multiplies sum_thw bhat = rstd * (sum b - cnt*mean)
      s2 += nc3[((size_t)n * Cp + c) * 3 + 2] +
            dzn * inv_cps * rstd * (ncf[((size_t)n * Cp + c) * 2] - cnt_per_sample * mean);
    }
    }
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    const double A = (double)gamma[c] * rstd;
    const double Cc = -A * rstd * s2 / count;
    if (q == 0) { coefA[c] = (float)A; coefC[c] = (float)Cc; }
    for (int n = q; n < B; n += 4) {
      const double dzn = se ? (double)dz[(size_t)n * C + c] : 0.0;
      coefB[(size_t)n * Cp + c] = (float)(A * dzn * inv_cps - A * s1 / count - Cc * mean);
    }
    if (q == 0) {
      if (dgamma) dgamma[c] += (float)s2;
      if (dbeta) dbeta[c] += (float)s1;
    }
  }
  SECLK(7)
}

}  // namespace

extern "C" int c3d_bn_finalize(const double* sums, int32_t stripes, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float momentum, float eps, int32_t C, int32_t Cp, int32_t training, float* ss,
                               float* mr, void* stream) {
  if (!gamma || !beta || !ss || C <= 0 || Cp < C) return C3D_E_BADARG;
  if (training && (!sums || stripes < 1)) return C3D_E_BADARG;
  if (!training && (!running_mean || !running_var)) return C3D_E_BADARG;
  bn_finalize_kernel<<<dim3((Cp * 16 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      sums, stripes, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, C, Cp,
      training, ss, mr);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_se_finalize(const double* nc, int32_t B, double cnt_per_sample, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float momentum, float eps, int32_t C, int32_t Cp,
                                  int32_t training, const float* w1, const float* b1, const float* w2,
                                  const float* b2, int32_t Cr, float* ss, float* mr, float* gate, float* hid,
                                  void* stream) {
  if (!nc || !gamma || !beta || !ss || C <= 0 || Cp < C || B <= 0) return C3D_E_BADARG;
  if (w1 && (!b1 || !w2 || !b2 || !gate || Cr <= 0)) return C3D_E_BADARG;
  const int Cw = C + ((8 - (C & 31)) & 31);
  const size_t lds = w1 ? ((size_t)B * C + (size_t)9 * B * Cr + (size_t)Cr * Cw + (size_t)C * Cr) * sizeof(float) : 0;
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_se_finalize_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
  }
  bn_se_finalize_kernel<<<dim3(w1 ? SE_SLICES : 1), dim3(SE_THREADS), lds, reinterpret_cast<hipStream_t>(stream)>>>(
      nc, B, cnt_per_sample, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, C, Cp,
      training, w1, b1, w2, b2, Cr, ss, mr, gate, hid);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bn_bwd_coef(const double* dsums, int32_t stripes, double count, const float* gamma, const float* mr, int32_t C,
                               int32_t Cp, float* coef, float* dgamma, float* dbeta, void* stream) {
  if (!dsums || stripes < 1 || !gamma || !mr || !coef || C <= 0 || Cp < C) return C3D_E_BADARG;
  bn_bwd_coef_kernel<<<dim3((Cp * 16 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      dsums, stripes, count, gamma, mr, C, Cp, coef, dgamma, dbeta);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_se_bn_bwd_coef(const double* nc3, const double* ncf, int32_t B, double cnt_per_sample,
                                  const float* gamma, const float* mr, const float* ss, int32_t C, int32_t Cp,
                                  const float* w1, const float* w2, const float* gate, const float* hid, int32_t Cr,
                                  float* coefA, float* coefC, float* coefB, float* dgamma, float* dbeta,
                                  float* dw1, float* db1, float* dw2, float* db2, void* stream) {
  if (!nc3 || !ncf || !gamma || !mr || !ss || !coefA || !coefC || !coefB || C <= 0 || Cp < C || B <= 0)
    return C3D_E_BADARG;
  if (w1 && (!w2 || !gate || !hid || !dw1 || !db1 || !dw2 || !db2 || Cr <= 0)) return C3D_E_BADARG;
  const size_t lds = w1 ? ((size_t)3 * B * C + (size_t)2 * B * Cr + (size_t)C * Cr + (8 * Cr <= C ? 0 : (size_t)8 * B * Cr)) * sizeof(float) : 0;
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&se_bn_bwd_coef_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
  }
  se_bn_bwd_coef_kernel<<<dim3(w1 ? SE_SLICES : 1), dim3(SE_THREADS), lds, reinterpret_cast<hipStream_t>(stream)>>>(
      nc3, ncf, B, cnt_per_sample, gamma, mr, ss, C, Cp, w1, w2, gate, hid, Cr, coefA, coefC, coefB, dgamma, dbeta,
      dw1, db1, dw2, db2);
  C3D_CHECK_LAUNCH();
  return 0;
}

#ifdef C3D_SE_CLOCK
extern "C" int c3d_debug_se_clock(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_se_clk), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(c3d_se_clk), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
