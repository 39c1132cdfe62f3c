// In-kernel BatchNorm finalisation by the LAST workgroup of the producing kernel ("last-workgroup-done" ticket).
//
// Round 1 ran one tiny kernel after every statistics producer (c3d_bn_finalize / c3d_bn_bwd_coef: 166 launches of
// 5-7 us per B=32 step, every one on the critical path between two latency-bound kernels, plus two kernel
// boundaries each).  Here the producer's workgroups take a ticket after their statistics atomics; the workgroup
// that draws the last ticket reads the completed f64 sums and writes scale/shift (+ running statistics) or the
// BatchNorm-backward coefficients itself.
//
// Protocol (MI355X: 8 XCDs with private L2s; HIP promises no dispatch order):
//   every workgroup : f64 atomicAdd (agent scope, performed at the memory side, not in an XCD's L2)
//                     -> s_waitcnt vmcnt(0) in every wave that issued them  (the atomics have been performed)
//                     -> __syncthreads() -> ONE returning agent-scope atomic increment of the ticket word
//   last workgroup  : ticket == gridDim - 1 -> ONE agent-scope acquire fence -> __syncthreads()
//                     -> reads the sums with relaxed agent-scope atomic loads (L1 bypassed; the lines were only
//                        ever touched by memory-side atomics in this launch) -> writes the outputs with plain
//                        stores: their readers are LATER kernels.
// No release fence is needed: the payload is atomics, not plain stores.  The ticket word lives in the stage's
// accumulator region, zeroed by the same memset as the sums.
#pragma once
#include "common.h"
#include "../../include/change3d_hip.h"

namespace c3dfin {

__device__ __forceinline__ double ld_sum(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All threads of the workgroup call this after issuing their statistics atomics.  `flag` is one int of LDS scratch.
// Returns true (workgroup-uniform) in the workgroup that took the last of `nblocks` tickets.
__device__ __forceinline__ bool last_workgroup(uint32_t* ticket, uint32_t nblocks, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == nblocks - 1u) ? 1 : 0;
    if (t == nblocks - 1u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return *flag != 0;
}

// Forward: sums f64 [stripes][2][C] (sum, sum of squares) -> ss = (scale | shift), mr = (mean | rstd), running
// statistics (same arithmetic as bn_finalize_kernel).  Called by every thread of the last workgroup; `nthreads` is a
// multiple of 16 (16 adjacent lanes share a channel: one stripe each, fixed-order butterfly).
__device__ __forceinline__ void bn_forward(const c3d_bn_fin& f, const double* sums, int stripes, int C, int Cp,
                                           int tid, int nthreads) {
  if (tid == 0 && f.training && f.nbt) *f.nbt += 1;
  if (tid >= nthreads) return;   // (whole 16-lane groups only)
  const int kq = tid & 15;
  for (int c0 = 0; c0 < Cp; c0 += nthreads >> 4) {
    const int c = c0 + (tid >> 4);
    double s1 = 0, s2 = 0;
    if (c < C) {
      for (int k = kq; k < stripes; k += 16) { s1 += ld_sum(sums + (size_t)k * 2 * C + c); s2 += ld_sum(sums + (size_t)k * 2 * C + C + c); }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (kq != 0 || c >= Cp) continue;
    if (c >= C) { f.ss[c] = 0.f; f.ss[Cp + c] = 0.f; if (f.mr) { f.mr[c] = 0.f; f.mr[Cp + c] = 0.f; } continue; }
    const double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;
    if (var < 0) var = 0;
    if (f.running_mean) {
      const double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
      f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
      f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float meanf = (float)mean;
    const float sc = f.gamma[c] * rstd;
    f.ss[c] = sc;
    f.ss[Cp + c] = f.beta[c] - meanf * sc;
    if (f.mr) { f.mr[c] = meanf; f.mr[Cp + c] = rstd; }
  }
}

// Consumer side: the sums were completed by an EARLIER launch (plain loads).  Every thread of the workgroup calls this;
// scale/shift of the channels [c_lo, c_lo + n) land in lds_sc / lds_sh (zero for padding channels); the `owner`
// workgroup of those channels also writes ss / mr / running statistics.  Four adjacent lanes share a channel and add
// the 16 stripes in the SAME tree as bn_finalize_kernel's 16-lane butterfly (lane q: ((s[4q]+s[4q+1]) + (s[4q+2]+
// s[4q+3])), then xor 1, xor 2), so the result is bit-identical to the separate launch.  Ends with __syncthreads().
__device__ __forceinline__ void bn_consume(const c3d_bn_fin& f, int C, int Cp, int c_lo, int n, bool owner, float* lds_sc,
                                           float* lds_sh, int tid, int nthreads) {
  const int q = tid & 3;
  for (int l0 = 0; l0 < n; l0 += nthreads >> 2) {
    const bool grp = (tid >> 2) < (nthreads >> 2);      // (a trailing partial group of four idles)
    const int lc = l0 + (tid >> 2), c = c_lo + lc;
    const bool live = grp && lc < n && c < C;
    double s1 = 0, s2 = 0;
    if (live) {
      const double* p = f.sums + (size_t)(4 * q) * 2 * C + c;
      const double a0 = p[0], a1 = p[(size_t)2 * C], a2 = p[(size_t)4 * C], a3 = p[(size_t)6 * C];
      const double b0 = p[C], b1 = p[(size_t)3 * C], b2 = p[(size_t)5 * C], b3 = p[(size_t)7 * C];
      s1 = (a0 + a1) + (a2 + a3);
      s2 = (b0 + b1) + (b2 + b3);
    }
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    if (q != 0 || !grp || lc >= n) continue;
    float sc = 0.f, sh = 0.f, meanf = 0.f, rstd = 0.f;
    if (c < C) {
      const double mean = s1 / f.count;
      double var = s2 / f.count - mean * mean;
      if (var < 0) var = 0;
      if (owner && f.running_mean) {
        const double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
        f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
        f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
      }
      rstd = (float)(1.0 / sqrt(var + (double)f.eps));
      meanf = (float)mean;
      sc = f.gamma[c] * rstd;
      sh = f.beta[c] - meanf * sc;
    }
    lds_sc[lc] = sc; lds_sh[lc] = sh;
    if (owner && c < Cp) {
      f.ss[c] = sc; f.ss[Cp + c] = sh;
      if (f.mr) { f.mr[c] = meanf; f.mr[Cp + c] = rstd; }
    }
  }
  __syncthreads();
}

// Backward: dsums f64 [stripes][2][C] = (sum g, sum g*xhat) -> coef = (A | B | C) with dx = A*g + B + C*x, and the
// BatchNorm parameter gradients (+=) (same arithmetic as bn_bwd_coef_kernel).  f.ss is the coefficient vector [3][Cp],
// f.mr the saved (mean | rstd), f.running_mean / f.running_var carry dgamma / dbeta.
__device__ __forceinline__ void bn_backward(const c3d_bn_fin& f, const double* dsums, int stripes, int C, int Cp,
                                            int tid, int nthreads) {
  if (tid >= nthreads) return;   // (whole 16-lane groups only)
  const int kq = tid & 15;
  float* coef = f.ss;
  float* dgamma = f.running_mean;
  float* dbeta = f.running_var;
  for (int c0 = 0; c0 < Cp; c0 += nthreads >> 4) {
    const int c = c0 + (tid >> 4);
    double s1 = 0, s2 = 0;
    if (c < C) {
      for (int k = kq; k < stripes; k += 16) { s1 += ld_sum(dsums + (size_t)k * 2 * C + c); s2 += ld_sum(dsums + (size_t)k * 2 * C + C + c); }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (kq != 0 || c >= Cp) continue;
    if (c >= C) { coef[c] = 0.f; coef[Cp + c] = 0.f; coef[2 * Cp + c] = 0.f; continue; }
    const double mean = f.mr[c], rstd = f.mr[Cp + c];
    const double A = (double)f.gamma[c] * rstd;
    const double Cc = -A * rstd * s2 / f.count;
    const double Bc = -A * s1 / f.count - Cc * mean;
    coef[c] = (float)A;
    coef[Cp + c] = (float)Bc;
    coef[2 * Cp + c] = (float)Cc;
    if (dgamma) dgamma[c] += (float)s2;
    if (dbeta) dbeta[c] += (float)s1;
  }
}

// Consumer side of the backward coefficients: one channel, from the completed single-stripe sums of an EARLIER launch
// (plain loads).  Same arithmetic as bn_bwd_coef_kernel; `acc` = this caller also accumulates dgamma / dbeta
// (f.running_mean / f.running_var) and writes the coefficient vector f.ss when it is non-NULL.
__device__ __forceinline__ void bn_bwd_coef_consume(const c3d_bn_fin& f, int C, int Cp, int c, bool acc, float& cA,
                                                    float& cB, float& cC) {
  cA = 0.f; cB = 0.f; cC = 0.f;
  if (c < C) {
    const double s1 = f.sums[c], s2 = f.sums[C + c];
    const double mean = f.mr[c], rstd = f.mr[Cp + c];
    const double A = (double)f.gamma[c] * rstd;
    const double Cc = -A * rstd * s2 / f.count;
    const double Bc = -A * s1 / f.count - Cc * mean;
    cA = (float)A; cB = (float)Bc; cC = (float)Cc;
    if (acc) {
      if (f.running_mean) f.running_mean[c] += (float)s2;
      if (f.running_var) f.running_var[c] += (float)s1;
    }
  }
  if (acc && f.ss && c < Cp) { f.ss[c] = cA; f.ss[Cp + c] = cB; f.ss[2 * Cp + c] = cC; }
}

// Eight consecutive channels at once (no accumulation): every load is issued before the first result is computed.  Called in a
// loop, bn_bwd_coef_consume's guarded loads came out as eight dependent memory round trips (five loads, s_waitcnt vmcnt(0),
// arithmetic; next channel) in front of a kernel's first tile -- ~8 us of c3d_pw_wgrad's ~40 us on the 32 x 32 maps (round 5,
// from the ISA).  Channels past C read channel C - 1 (a valid address) and are zeroed; per-channel arithmetic unchanged.
__device__ __forceinline__ void bn_bwd_coef_consume8(const c3d_bn_fin& f, int C, int Cp, int c0, float (&cA)[8], float (&cB)[8],
                                                     float (&cC)[8]) {
  double s1[8], s2[8];
  float mean[8], rstd[8], gam[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j < C ? c0 + j : C - 1;
    s1[j] = f.sums[c]; s2[j] = f.sums[C + c];
    mean[j] = f.mr[c]; rstd[j] = f.mr[Cp + c];
    gam[j] = f.gamma[c];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double A = (double)gam[j] * (double)rstd[j];
    const double Cc = -A * (double)rstd[j] * s2[j] / f.count;
    const double Bc = -A * s1[j] / f.count - Cc * (double)mean[j];
    const bool live = c0 + j < C;
    cA[j] = live ? (float)A : 0.f; cB[j] = live ? (float)Bc : 0.f; cC[j] = live ? (float)Cc : 0.f;
  }
}

// ---- BatchNorm_b of blocks without SqueezeExcitation: per-sample sums consumed directly -----------------------------
// Forward.  nc f64 [B][Cp][2] (c3d_dw333_fwd) -> scale / shift of ALL channels into lds_sc / lds_sh; `owner` also writes
// ss / mr / the running statistics.  Four adjacent lanes split the batch loop and combine with xor 1, xor 2 -- the
// summation tree of bn_se_finalize_kernel, so the result is bit-identical to the separate launch.  Ends with
// __syncthreads().
__device__ __forceinline__ void bn_consume_nc(const c3d_bn_fin& f, int C, int Cp, bool owner, float* lds_sc, float* lds_sh,
                                              int tid, int nthreads) {
  const int B = f.batch, q = tid & 3;
  for (int c0 = 0; c0 < Cp; c0 += nthreads >> 2) {
    const int c = c0 + (tid >> 2);
    const bool live = c < C;
    double s1 = 0, s2 = 0;
    if (live) {
#pragma unroll 4
      for (int n = q; n < B; n += 4) { s1 += f.sums[((size_t)n * Cp + c) * 2]; s2 += f.sums[((size_t)n * Cp + c) * 2 + 1]; }
    }
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    if (q != 0 || c >= Cp) continue;
    float sc = 0.f, sh = 0.f, meanf = 0.f, rstd = 0.f;
    if (live) {
      const double mean = s1 / f.count;
      double var = s2 / f.count - mean * mean;
      if (var < 0) var = 0;
      if (owner && f.running_mean) {
        const double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
        f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
        f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unb);
      }
      rstd = (float)(1.0 / sqrt(var + (double)f.eps));
      meanf = (float)mean;
      sc = f.gamma[c] * rstd;
      sh = f.beta[c] - meanf * sc;
    }
    lds_sc[c] = sc; lds_sh[c] = sh;
    if (owner) {
      f.ss[c] = sc; f.ss[Cp + c] = sh;
      if (f.mr) { f.mr[c] = meanf; f.mr[Cp + c] = rstd; }
    }
  }
  __syncthreads();
}

// SqueezeExcitation gate of the samples [n_lo, n_lo + ns) by the consuming workgroup (after bn_consume_nc: scale / shift in
// lds_sc / lds_sh): the three steps of bn_se_finalize_kernel with its arithmetic and summation order -- z = fma(sc,
// (float)(sum / cnt), sh); FC1 over 8 adjacent lanes (c = q, q + 8, ...) combined with xor 1, 2, 4, + bias, ReLU; FC2 as one
// sequential fma chain over the hidden units, accurate expf -- so gate and hid are bit-identical to the separate launch.
// zg: LDS [ns][Cp] (z, then overwritten by the gate), hl: LDS [ns][Cr].  own_lo / own_hi: samples whose first row lies in
// this workgroup's rows (it writes their gate / hid to global memory for the backward pass).  Ends with __syncthreads().
__device__ __forceinline__ void se_gate_consume(const double* nc, double cnt_per_sample, int C, int Cp, const float* w1,
                                                const float* b1, const float* w2, const float* b2, int Cr, int n_lo, int ns,
                                                int own_lo, int own_hi, const float* lds_sc, const float* lds_sh, float* zg,
                                                float* hl, float* gate, float* hid, int tid, int nthreads) {
  for (int idx = tid; idx < ns * Cp; idx += nthreads) {
    const int n = idx / Cp, c = idx - n * Cp;
    zg[idx] = c < C ? fmaf(lds_sc[c], (float)(nc[((size_t)(n_lo + n) * Cp + c) * 2] / cnt_per_sample), lds_sh[c]) : 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < ns * Cr * 8; idx += nthreads) {   // 8 lanes per (sample, hidden unit)
    const int i = idx >> 3, q = idx & 7;
    const int n = i / Cr, r = i - n * Cr;
    // (w1 / w2 come from global memory (L1 / L2 hits): the loads of eight steps are requested together, the FMA chain keeps its
    // order -- one exposed cache round trip per FMA was ~2.5 us of every conv_c launch's prologue)
    float a = 0.f;
    int c = q;
    for (; c + 56 < C; c += 64) {
      float wv[8], zv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { wv[u] = w1[(size_t)r * C + c + 8 * u]; zv[u] = zg[n * Cp + c + 8 * u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) a = fmaf(wv[u], zv[u], a);
    }
    for (; c < C; c += 8) a = fmaf(w1[(size_t)r * C + c], zg[n * Cp + c], a);
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
    a += b1[r];
    a = a > 0.f ? a : 0.f;
    if (q == 0) {
      hl[i] = a;
      if (n_lo + n >= own_lo && n_lo + n <= own_hi) hid[(size_t)(n_lo + n) * Cr + r] = a;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < ns * Cp; idx += nthreads) {
    const int n = idx / Cp, c = idx - n * Cp;
    float g = 0.f;
    if (c < C) {
      float a = b2[c];
      int r = 0;
      for (; r + 8 <= Cr; r += 8) {
        float wv[8], hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { wv[u] = w2[(size_t)c * Cr + r + u]; hv[u] = hl[n * Cr + r + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) a = fmaf(wv[u], hv[u], a);
      }
      for (; r < Cr; ++r) a = fmaf(w2[(size_t)c * Cr + r], hl[n * Cr + r], a);
      g = 1.0f / (1.0f + expf(-a));
    }
    zg[idx] = g;
    if (n_lo + n >= own_lo && n_lo + n <= own_hi) gate[(size_t)(n_lo + n) * Cp + c] = g;
  }
  __syncthreads();
}

// Backward.  nc3 f64 [B][Cp][3] (Swish/SE-backward epilogue: d gate, sum t1, sum t1*bhat) -> db = A*t1 + Bc + C*b for the
// channel `c` (no SE: Bc is the same for every sample), same arithmetic and summation tree as se_bn_bwd_coef_kernel's
// no-SE branch.  Called by the four lanes q = 0..3 of a channel (adjacent lanes); the result is valid in all four.
__device__ __forceinline__ void bn_b_bwd_coef_nc(const c3d_bn_fin& f, int C, int Cp, int c, int q, bool owner, float& cA,
                                                 float& cB, float& cC) {
  const int B = f.batch;
  double s1 = 0, s2 = 0;
  if (c < C) {
#pragma unroll 4
    for (int n = q; n < B; n += 4) { s1 += f.sums[((size_t)n * Cp + c) * 3 + 1]; s2 += f.sums[((size_t)n * Cp + c) * 3 + 2]; }
  }
  s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
  s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
  cA = 0.f; cB = 0.f; cC = 0.f;
  if (c < C) {
    const double mean = f.mr[c], rstd = f.mr[Cp + c];
    const double A = (double)f.gamma[c] * rstd;
    const double Cc = -A * rstd * s2 / f.count;
    cA = (float)A; cC = (float)Cc;
    cB = (float)(-A * s1 / f.count - Cc * mean);
    if (owner && q == 0) {
      if (f.running_mean) f.running_mean[c] += (float)s2;
      if (f.running_var) f.running_var[c] += (float)s1;
    }
  }
}

}  // namespace c3dfin
