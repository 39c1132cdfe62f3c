// Caption-decoder kernels of the change-captioning path (reference model/caption_decoder.py:526-613 CaptionDecoder,
// :316-423 Mesh_TransformerDecoderLayer, :272-314 PositionalEncoding; loss of scripts/train_CC.py:124-132).
//
// Sizes (BASELINE.json configs[4]): L = 52 caption tokens, B = 16, d = 192, 8 heads of 24, memory S = 256 encoder tokens,
// vocabulary ~500, 3 layers.  Everything here is tiny next to the encoder (the whole decoder moves < 100 MB per step):
// the kernels are written for few launches and exact arithmetic (f32 compute, f64 reductions), not for a roofline.
// The linear layers run on c3d_pw_gemm / c3d_pw_wgrad (pw_wide.hip); what is left:
//   * token embedding + sinusoidal position table (+ dropout) and its gradient (scatter-add into the embedding rows)
//   * multi-head attention per (sample, head): scores, (causal) softmax, dropout on the weights, weighted values;
//     backward re-reads the saved probabilities
//   * post-norm residual LayerNorm y = LN(x + drop(a)) forward / backward
//   * cross-entropy over the decoded steps (time step < decode length, target != ignore_index), mean over them
//   * element-wise dropout with a counter-based mask that backward regenerates from (seed, element index)
// Activations are SEQUENCE-FIRST rows: row = l * B + b (nn.MultiheadAttention's default layout), Dp = round_up(D, 8).
#include "common.h"
#include "../../include/change3d_hip.h"

namespace {

// counter-based uniform in [0,1): splitmix64 of (seed, index); same value in forward and backward
__device__ __forceinline__ float u01(const uint64_t seed, const uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float keep_scale(const float p, const uint64_t seed, const uint64_t idx) {
  return p <= 0.f ? 1.f : (u01(seed, idx) < p ? 0.f : 1.f / (1.f - p));
}

// ---- embedding + positions:  out[l*B+b][d] = drop(emb[tok[b][l]][d] + pe[l][d])
template <typename T>
__global__ void embed_posenc_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pe,
                                    T* __restrict__ out, int B, int L, int D, int Dp, int V, float p, uint64_t seed) {
  const int64_t n = (int64_t)L * B * Dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % Dp);
    const int64_t row = i / Dp;
    const int b = (int)(row % B), l = (int)(row / B);
    float v = 0.f;
    if (d < D) {
      int64_t t = tok[(int64_t)b * L + l];
      t = t < 0 ? 0 : (t >= V ? V - 1 : t);
      v = (emb[t * D + d] + pe[(int64_t)l * D + d]) * keep_scale(p, seed, (uint64_t)(row * D + d));
    }
    st1<T>(out + i, v);
  }
}

template <typename T>
__global__ void embed_bwd_kernel(const int64_t* __restrict__ tok, const T* __restrict__ dout, float* __restrict__ demb, int B, int L,
                                 int D, int Dp, int V, float p, uint64_t seed) {
  const int64_t n = (int64_t)L * B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int64_t row = i / D;
    const int b = (int)(row % B), l = (int)(row / B);
    int64_t t = tok[(int64_t)b * L + l];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float g = ld1<T>(dout + row * Dp + d) * keep_scale(p, seed, (uint64_t)(row * D + d));
    atomicAdd(demb + t * D + d, g);
  }
}

// ---- element-wise dropout (same kernel forward and backward)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int D, int Dp, float p, uint64_t seed) {
  const int64_t n = rows * Dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % Dp);
    const int64_t row = i / Dp;
    st1<T>(y + i, d < D ? ld1<T>(x + i) * keep_scale(p, seed, (uint64_t)(row * D + d)) : 0.f);
  }
}

// ---- LayerNorm of (x + a): one wave per row; mean / rstd saved (f32 [rows][2])
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ a, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mr,
                                                            int64_t rows, int D, int Dp, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[8];
  int nv = 0;
  double s = 0.0;
  for (int d = lane; d < D; d += 64, ++nv) {
    const float t = ld1<T>(x + row * Dp + d) + (a ? ld1<T>(a + row * Dp + d) : 0.f);
    v[nv] = t;
    s += (double)t;
  }
  s = wave_sum_d(s);
  const double mean = s / D;
  double q = 0.0;
  for (int i = 0; i < nv; ++i) { const double c = (double)v[i] - mean; q += c * c; }
  q = wave_sum_d(q);
  const float rstd = (float)(1.0 / sqrt(q / D + (double)eps));
  const float mf = (float)mean;
  nv = 0;
  for (int d = lane; d < Dp; d += 64) {
    float o = 0.f;
    if (d < D) o = (v[nv++] - mf) * rstd * gamma[d] + beta[d];
    st1<T>(y + row * Dp + d, o);
  }
  if (lane == 0) { mr[row * 2] = mf; mr[row * 2 + 1] = rstd; }
}

// dx = (gamma*dy - mean(gamma*dy) - xhat * mean(gamma*dy*xhat)) * rstd ;  dgamma += dy*xhat ; dbeta += dy   (xhat from x + a)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mr,
                                                            T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int64_t rows, int D, int Dp, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float dg[8], db[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t row = gw * rows_per_wave; row < (gw + 1) * rows_per_wave && row < rows; ++row) {
    const float mean = mr[row * 2], rstd = mr[row * 2 + 1];
    float xh[8], gd[8];
    int nv = 0;
    float s1 = 0.f, s2 = 0.f;
    for (int d = lane; d < D; d += 64, ++nv) {
      const float t = ld1<T>(x + row * Dp + d) + (a ? ld1<T>(a + row * Dp + d) : 0.f);
      const float g = ld1<T>(dy + row * Dp + d);
      xh[nv] = (t - mean) * rstd;
      gd[nv] = g * gamma[d];
      s1 += gd[nv]; s2 += gd[nv] * xh[nv];
      dg[nv] += g * xh[nv]; db[nv] += g;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float m1 = s1 / D, m2 = s2 / D;
    nv = 0;
    for (int d = lane; d < Dp; d += 64) {
      float o = 0.f;
      if (d < D) { o = (gd[nv] - m1 - xh[nv] * m2) * rstd; ++nv; }
      st1<T>(dx + row * Dp + d, o);
    }
  }
  int nv = 0;
  for (int d = lane; d < D; d += 64, ++nv) { atomicAdd(dgamma + d, dg[nv]); atomicAdd(dbeta + d, db[nv]); }
}

// o[d] = sum_j w[j] * m[j * hs + d] for d = lane % hd, with the 64 lanes split into 64 / hd groups over j and four
// independent partial sums per lane: the serial form (lanes < hd, one fma chain of Lk dependent LDS reads) cost ~18 k clocks
// per query row at Lk = 256.  Result valid in lanes < hd.
__device__ __forceinline__ float attn_row_times_matrix(const float* w, const float* m, int Lk, int hs, int hd, int lane) {
  const int np = 64 / hd;                       // groups (2 for hd = 24)
  const int part = lane / hd, d = lane - part * hd;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
  if (part < np) {
    int j = part;
    for (; j + 3 * np < Lk; j += 4 * np) {
      o0 = fmaf(w[j], m[(size_t)j * hs + d], o0);
      o1 = fmaf(w[j + np], m[(size_t)(j + np) * hs + d], o1);
      o2 = fmaf(w[j + 2 * np], m[(size_t)(j + 2 * np) * hs + d], o2);
      o3 = fmaf(w[j + 3 * np], m[(size_t)(j + 3 * np) * hs + d], o3);
    }
    for (; j < Lk; j += np) o0 = fmaf(w[j], m[(size_t)j * hs + d], o0);
  }
  float o = (o0 + o1) + (o2 + o3);
  float tot = o;
  for (int g = 1; g < np; ++g) tot += __shfl(o, d + g * hd, 64);   // (all lanes execute the shuffle; lanes < hd keep the sum)
  return tot;
}

// ---- multi-head attention, one workgroup per (sample b, head h); rows are sequence-first (row = l*B + b)
//   S[i][j] = scale * q_i . k_j (+ -inf for j > i when causal);  P = softmax_j(S);  Pd = drop(P);  o_i = sum_j Pd[i][j] v_j
// q at qb + row*ldq + h*hd (ld in elements), likewise k, v; P (pre-dropout) saved f32 [B*H][Lq][Lk].
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qb, const T* __restrict__ kb, const T* __restrict__ vb,
                                                       int ldq, int ldk, int ldv, T* __restrict__ ob, int ldo, float* __restrict__ P,
                                                       int B, int H, int Lq, int Lk, int hd, float scale, int causal, float p,
                                                       uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // rows are padded to an ODD stride: the score loop reads k_j . q_i with the lanes over j, and a stride of hd = 24
  // floats put 64 lanes on 8 banks (8-way conflict on every read)
  const int hs = hd | 1;
  float* ks = sm;                         // [Lk][hs]
  float* vs = ks + (size_t)Lk * hs;       // [Lk][hs]
  float* qs = vs + (size_t)Lk * hs;       // [Lq][hs]
  float* ps = qs + (size_t)Lq * hs;       // [4 waves][Lk]
  const int b = blockIdx.x % B, h = blockIdx.x / B;
  // gridDim.y workgroups share the query rows of one (sample, head): [i_lo, i_hi) each (K / V are staged by all of them --
  // 128 workgroups of 13 serial rows per wave left half of the 256 CUs idle)
  const int rows_y = (Lq + (int)gridDim.y - 1) / (int)gridDim.y;
  const int i_lo = (int)blockIdx.y * rows_y, i_hi = i_lo + rows_y < Lq ? i_lo + rows_y : Lq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < Lk * hd; i += 256) {
    const int j = i / hd, d = i - j * hd;
    ks[j * hs + d] = ld1<T>(kb + ((int64_t)j * B + b) * ldk + h * hd + d);
    vs[j * hs + d] = ld1<T>(vb + ((int64_t)j * B + b) * ldv + h * hd + d);
  }
  for (int i = i_lo * hd + tid; i < i_hi * hd; i += 256) {
    const int l = i / hd, d = i - l * hd;
    qs[l * hs + d] = ld1<T>(qb + ((int64_t)l * B + b) * ldq + h * hd + d) * scale;
  }
  __syncthreads();
  float* pw = ps + (size_t)wave * Lk;
  float* Pout = P + (size_t)blockIdx.x * Lq * Lk;   // [(h*B + b)][Lq][Lk]
  for (int i = i_lo + wave; i < i_hi; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
      float s = -INFINITY;
      if (!causal || j <= i) {
        s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(qs[i * hs + d], ks[j * hs + d], s);
      }
      pw[j] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float e = pw[j] == -INFINITY ? 0.f : expf(pw[j] - mx); pw[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < Lk; j += 64) {
      const float pr = pw[j] * inv;
      Pout[(size_t)i * Lk + j] = pr;
      pw[j] = pr * keep_scale(p, seed, ((uint64_t)blockIdx.x * Lq + i) * Lk + j);
    }
    __builtin_amdgcn_wave_barrier();
    {
      const float o = attn_row_times_matrix(pw, vs, Lk, hs, hd, lane);
      if (lane < hd) st1<T>(ob + ((int64_t)i * B + b) * ldo + h * hd + lane, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// backward: dPd = dO V^T ; dP = dPd * mask ; dS = P * (dP - sum_j dP*P) ; dQ = scale * dS K ; dK = scale * dS^T Q ; dV = Pd^T dO
// Two passes over one [Lq][Lk] LDS matrix (first dS, then the dropped probabilities): every (key, channel) element of
// dK / dV is owned by one thread that sums over the query rows -- no atomics (the first version accumulated dK / dV with
// LDS atomics per query row: 0.59 ms per launch for 128 workgroups).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qb, const T* __restrict__ kb, const T* __restrict__ vb,
                                                       int ldq, int ldk, int ldv, const T* __restrict__ dob, int ldo,
                                                       const float* __restrict__ P, T* __restrict__ dqb, T* __restrict__ dkb,
                                                       T* __restrict__ dvb, int lddq, int lddk, int lddv, int B, int H, int Lq,
                                                       int Lk, int hd, float scale, float p, uint64_t seed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hs = hd | 1;                   // odd row stride: see attn_fwd_kernel
  float* ks = sm;                          // [Lk][hs]
  float* vs = ks + (size_t)Lk * hs;        // [Lk][hs]
  float* qs = vs + (size_t)Lk * hs;        // [Lq][hs]
  float* dos = qs + (size_t)Lq * hs;       // [Lq][hs]
  float* DS = dos + (size_t)Lq * hs;       // [Lq][Lk]: dS, then dropped P
  const int b = blockIdx.x % B, h = blockIdx.x / B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < Lk * hd; i += 256) {
    const int j = i / hd, d = i - j * hd;
    ks[j * hs + d] = ld1<T>(kb + ((int64_t)j * B + b) * ldk + h * hd + d);
    vs[j * hs + d] = ld1<T>(vb + ((int64_t)j * B + b) * ldv + h * hd + d);
  }
  for (int i = tid; i < Lq * hd; i += 256) {
    const int l = i / hd, d = i - l * hd;
    qs[l * hs + d] = ld1<T>(qb + ((int64_t)l * B + b) * ldq + h * hd + d);
    dos[l * hs + d] = ld1<T>(dob + ((int64_t)l * B + b) * ldo + h * hd + d);
  }
  // the saved probabilities of this (sample, head) go to LDS with the operands, in one coalesced pass: read inside the
  // row loops below they were a serial global round trip per 64 keys (13 rows x 4-8 trips per wave: most of the launch)
  const float* Pin = P + (size_t)blockIdx.x * Lq * Lk;
  for (int e = tid; e < Lq * Lk; e += 256) DS[e] = Pin[e];
  __syncthreads();
  for (int i = wave; i < Lq; i += 4) {
    float* ds = DS + (size_t)i * Lk;
    float dot = 0.f;
    float prr[8];   // this lane's probabilities of the row (Lk <= 512; beyond that they are read again from memory)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + 64 * u;
      prr[u] = 0.f;
      if (j < Lk) {
        const float pr = ds[j];
        prr[u] = pr;
        float dpd = 0.f;
        for (int d = 0; d < hd; ++d) dpd = fmaf(dos[i * hs + d], vs[j * hs + d], dpd);
        const float dp = dpd * keep_scale(p, seed, ((uint64_t)blockIdx.x * Lq + i) * Lk + j);
        ds[j] = dp;
        dot += dp * pr;
      }
    }
    for (int j = lane + 512; j < Lk; j += 64) {
      const float pr = ds[j];
      float dpd = 0.f;
      for (int d = 0; d < hd; ++d) dpd = fmaf(dos[i * hs + d], vs[j * hs + d], dpd);
      const float dp = dpd * keep_scale(p, seed, ((uint64_t)blockIdx.x * Lq + i) * Lk + j);
      ds[j] = dp;
      dot += dp * pr;
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + 64 * u;
      if (j < Lk) ds[j] = prr[u] * (ds[j] - dot) * scale;
    }
    for (int j = lane + 512; j < Lk; j += 64) ds[j] = Pin[(size_t)i * Lk + j] * (ds[j] - dot) * scale;
    __builtin_amdgcn_wave_barrier();
    {   // dQ_i = sum_j dS[i][j] K_j
      const float o = attn_row_times_matrix(ds, ks, Lk, hs, hd, lane);
      if (lane < hd) st1<T>(dqb + ((int64_t)i * B + b) * lddq + h * hd + lane, o);
    }
  }
  __syncthreads();
  for (int e = tid; e < Lk * hd; e += 256) {   // dK_j = sum_i dS[i][j] Q_i
    const int j = e / hd, d = e - j * hd;
    float o = 0.f;
    for (int i = 0; i < Lq; ++i) o = fmaf(DS[(size_t)i * Lk + j], qs[i * hs + d], o);
    st1<T>(dkb + ((int64_t)j * B + b) * lddk + h * hd + d, o);
  }
  __syncthreads();
  for (int e = tid; e < Lq * Lk; e += 256)
    DS[e] = Pin[e] * keep_scale(p, seed, (uint64_t)blockIdx.x * Lq * Lk + e);
  __syncthreads();
  for (int e = tid; e < Lk * hd; e += 256) {   // dV_j = sum_i Pd[i][j] dO_i
    const int j = e / hd, d = e - j * hd;
    float o = 0.f;
    for (int i = 0; i < Lq; ++i) o = fmaf(DS[(size_t)i * Lk + j], dos[i * hs + d], o);
    st1<T>(dvb + ((int64_t)j * B + b) * lddv + h * hd + d, o);
  }
}

// ---- cross-entropy over the decoded steps.  logits row (l, b) = l*B + b, Vp columns; target = caps[b][l+1];
// valid iff l < declen[b] and target != ignore (a counted target outside [0, V) poisons the loss with NaN).
// acc f64 [3] = (sum nll, count, top-1 hits), zeroed by the entry point.
template <typename T>
__global__ __launch_bounds__(256) void cap_ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ caps,
                                                         const int64_t* __restrict__ declen, double* __restrict__ acc,
                                                         float* __restrict__ lse, int B, int L, int V, int Vp, int64_t ignore) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)L * B) return;
  const int b = (int)(row % B), l = (int)(row / B);
  const int64_t tgt = l + 1 < L ? caps[(int64_t)b * L + l + 1] : ignore;
  const bool valid = l < declen[b] && tgt != ignore && tgt >= 0 && tgt < V;
  if (!valid) {
    if (lane == 0) {
      lse[row] = 0.f;
      // a counted position whose target is outside [0, V) (wrong --vocab_size, corrupt word map): nn.CrossEntropyLoss
      // in the reference trips a device assert; here the loss turns NaN -- loud in every log, no host read-back
      if (l < declen[b] && tgt != ignore) atomicAdd(acc, (double)NAN);
    }
    return;
  }
  float mx = -INFINITY;
  int am = 0;
  for (int j = lane; j < V; j += 64) { const float x = ld1<T>(logits + row * Vp + j); if (x > mx) { mx = x; am = j; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {   // arg max (first index on ties, as torch.topk(1))
    const float mo = __shfl_xor(mx, o, 64);
    const int ao = __shfl_xor(am, o, 64);
    if (mo > mx || (mo == mx && ao < am)) { mx = mo; am = ao; }
  }
  double s = 0.0;
  for (int j = lane; j < V; j += 64) s += (double)expf(ld1<T>(logits + row * Vp + j) - mx);
  s = wave_sum_d(s);
  const float l_s_e = mx + (float)log(s);
  if (lane == 0) {
    lse[row] = l_s_e;
    atomicAdd(acc, (double)(l_s_e - ld1<T>(logits + row * Vp + tgt)));
    atomicAdd(acc + 1, 1.0);
    if (am == tgt) atomicAdd(acc + 2, 1.0);
  }
}

__global__ void cap_ce_finalize_kernel(const double* __restrict__ acc, float* __restrict__ loss) {
  loss[0] = acc[1] > 0 ? (float)(acc[0] / acc[1]) : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void cap_ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ caps,
                                                         const int64_t* __restrict__ declen, const double* __restrict__ acc,
                                                         const float* __restrict__ lse, const float* __restrict__ dloss,
                                                         T* __restrict__ dlogits, int B, int L, int V, int Vp, int64_t ignore) {
  const int64_t n = (int64_t)L * B * Vp;
  const float sc = (dloss ? dloss[0] : 1.f) / (acc[1] > 0 ? (float)acc[1] : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % Vp);
    const int64_t row = i / Vp;
    const int b = (int)(row % B), l = (int)(row / B);
    const int64_t tgt = l + 1 < L ? caps[(int64_t)b * L + l + 1] : ignore;
    const bool valid = l < declen[b] && tgt != ignore && tgt >= 0 && tgt < V;
    float g = 0.f;
    if (valid && j < V) g = (expf(ld1<T>(logits + i) - lse[row]) - (j == tgt ? 1.f : 0.f)) * sc;
    st1<T>(dlogits + i, g);
  }
}

// ---- gradient clipping by value (reference model/utils.py:481-491 clip_gradient) over a flat buffer
__global__ void clamp_kernel(float* __restrict__ g, int64_t n, float lim) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    g[i] = fminf(fmaxf(g[i], -lim), lim);
}

inline int grid_for(int64_t n, int block = 256) {
  int64_t g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

#define CAP_DISPATCH(dtype, F32, BF16) \
  if ((dtype) == C3D_DT_F32) { F32; } else if ((dtype) == C3D_DT_BF16) { BF16; } else return C3D_E_BADARG;

}  // namespace

extern "C" int c3d_cap_embed_fwd(const int64_t* tokens, const float* emb, const float* pe, void* out, int32_t B, int32_t L,
                                 int32_t D, int32_t V, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!tokens || !emb || !pe || !out || B <= 0 || L <= 0 || D <= 0 || V <= 0 || p < 0.f || p >= 1.f) return C3D_E_BADARG;
  const int Dp = (D + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)L * B * Dp);
  CAP_DISPATCH(dtype, (embed_posenc_kernel<float><<<g, 256, 0, s>>>(tokens, emb, pe, (float*)out, B, L, D, Dp, V, p, seed)),
               (embed_posenc_kernel<bf16_t><<<g, 256, 0, s>>>(tokens, emb, pe, (bf16_t*)out, B, L, D, Dp, V, p, seed)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_embed_bwd(const int64_t* tokens, const void* dout, float* demb, int32_t B, int32_t L, int32_t D, int32_t V,
                                 float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!tokens || !dout || !demb || B <= 0 || L <= 0 || D <= 0 || V <= 0) return C3D_E_BADARG;
  const int Dp = (D + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)L * B * D);
  CAP_DISPATCH(dtype, (embed_bwd_kernel<float><<<g, 256, 0, s>>>(tokens, (const float*)dout, demb, B, L, D, Dp, V, p, seed)),
               (embed_bwd_kernel<bf16_t><<<g, 256, 0, s>>>(tokens, (const bf16_t*)dout, demb, B, L, D, Dp, V, p, seed)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_dropout(const void* x, void* y, int64_t rows, int32_t D, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!x || !y || rows <= 0 || D <= 0 || p < 0.f || p >= 1.f) return C3D_E_BADARG;
  const int Dp = (D + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(rows * Dp);
  CAP_DISPATCH(dtype, (dropout_kernel<float><<<g, 256, 0, s>>>((const float*)x, (float*)y, rows, D, Dp, p, seed)),
               (dropout_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, rows, D, Dp, p, seed)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_layernorm_fwd(const void* x, const void* a, const float* gamma, const float* beta, void* y, float* mr,
                                     int64_t rows, int32_t D, float eps, int32_t dtype, void* stream) {
  if (!x || !gamma || !beta || !y || !mr || rows <= 0 || D <= 0 || D > 512) return C3D_E_BADARG;
  const int Dp = (D + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = (int)((rows + 3) / 4);
  CAP_DISPATCH(dtype, (layernorm_fwd_kernel<float><<<g, 256, 0, s>>>((const float*)x, (const float*)a, gamma, beta, (float*)y, mr, rows, D, Dp, eps)),
               (layernorm_fwd_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)a, gamma, beta, (bf16_t*)y, mr, rows, D, Dp, eps)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_layernorm_bwd(const void* x, const void* a, const void* dy, const float* gamma, const float* mr, void* dx,
                                     float* dgamma, float* dbeta, int64_t rows, int32_t D, int32_t dtype, void* stream) {
  if (!x || !dy || !gamma || !mr || !dx || !dgamma || !dbeta || rows <= 0 || D <= 0 || D > 512) return C3D_E_BADARG;
  const int Dp = (D + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpw = (int)((rows + 1023) / 1024);            // <= 1024 waves: bounded parameter-gradient atomics
  const int64_t waves = (rows + rpw - 1) / rpw;
  const int g = (int)((waves + 3) / 4);
  CAP_DISPATCH(dtype, (layernorm_bwd_kernel<float><<<g, 256, 0, s>>>((const float*)x, (const float*)a, (const float*)dy, gamma, mr, (float*)dx, dgamma, dbeta, rows, D, Dp, rpw)),
               (layernorm_bwd_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)a, (const bf16_t*)dy, gamma, mr, (bf16_t*)dx, dgamma, dbeta, rows, D, Dp, rpw)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_attn_fwd(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk, int32_t ldv, void* o, int32_t ldo,
                                float* P, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale, int32_t causal,
                                float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!q || !k || !v || !o || !P || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || hd <= 0 || hd > 64 || p < 0.f || p >= 1.f) return C3D_E_BADARG;
  const int hs = hd | 1;
  const size_t lds = ((size_t)2 * Lk * hs + (size_t)Lq * hs + (size_t)4 * Lk) * sizeof(float);
  const int qsplit = Lq >= 16 ? 4 : 1;   // query rows of a (sample, head) over 4 workgroups
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  CAP_DISPATCH(dtype, (attn_fwd_kernel<float><<<dim3(B * H, qsplit), 256, lds, s>>>((const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, (float*)o, ldo, P, B, H, Lq, Lk, hd, scale, causal, p, seed)),
               (attn_fwd_kernel<bf16_t><<<dim3(B * H, qsplit), 256, lds, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, (bf16_t*)o, ldo, P, B, H, Lq, Lk, hd, scale, causal, p, seed)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_attn_bwd(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk, int32_t ldv, const void* dout,
                                int32_t ldo, const float* P, void* dq, void* dk, void* dv, int32_t lddq, int32_t lddk, int32_t lddv,
                                int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale, float p, uint64_t seed,
                                int32_t dtype, void* stream) {
  if (!q || !k || !v || !dout || !P || !dq || !dk || !dv || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || hd <= 0 || hd > 64) return C3D_E_BADARG;
  const int hs = hd | 1;
  const size_t lds = ((size_t)2 * Lk * hs + (size_t)2 * Lq * hs + (size_t)Lq * Lk) * sizeof(float);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  CAP_DISPATCH(dtype, (attn_bwd_kernel<float><<<B * H, 256, lds, s>>>((const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, (const float*)dout, ldo, P, (float*)dq, (float*)dk, (float*)dv, lddq, lddk, lddv, B, H, Lq, Lk, hd, scale, p, seed)),
               (attn_bwd_kernel<bf16_t><<<B * H, 256, lds, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, (const bf16_t*)dout, ldo, P, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv, B, H, Lq, Lk, hd, scale, p, seed)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_ce_fwd(const void* logits, const int64_t* caps, const int64_t* declen, double* acc2, float* lse, float* loss,
                              int32_t B, int32_t L, int32_t V, int64_t ignore_index, int32_t dtype, void* stream) {
  if (!logits || !caps || !declen || !acc2 || !lse || !loss || B <= 0 || L <= 0 || V <= 0) return C3D_E_BADARG;
  const int Vp = (V + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(acc2, 0, 3 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  const int g = (int)(((int64_t)L * B + 3) / 4);
  CAP_DISPATCH(dtype, (cap_ce_fwd_kernel<float><<<g, 256, 0, s>>>((const float*)logits, caps, declen, acc2, lse, B, L, V, Vp, ignore_index)),
               (cap_ce_fwd_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)logits, caps, declen, acc2, lse, B, L, V, Vp, ignore_index)));
  cap_ce_finalize_kernel<<<1, 1, 0, s>>>(acc2, loss);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cap_ce_bwd(const void* logits, const int64_t* caps, const int64_t* declen, const double* acc2, const float* lse,
                              const float* dloss, void* dlogits, int32_t B, int32_t L, int32_t V, int64_t ignore_index, int32_t dtype,
                              void* stream) {
  if (!logits || !caps || !declen || !acc2 || !lse || !dlogits || B <= 0 || L <= 0 || V <= 0) return C3D_E_BADARG;
  const int Vp = (V + 7) / 8 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((int64_t)L * B * Vp);
  CAP_DISPATCH(dtype, (cap_ce_bwd_kernel<float><<<g, 256, 0, s>>>((const float*)logits, caps, declen, acc2, lse, dloss, (float*)dlogits, B, L, V, Vp, ignore_index)),
               (cap_ce_bwd_kernel<bf16_t><<<g, 256, 0, s>>>((const bf16_t*)logits, caps, declen, acc2, lse, dloss, (bf16_t*)dlogits, B, L, V, Vp, ignore_index)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_clamp_(float* g, int64_t n, float limit, void* stream) {
  if (!g || n <= 0 || !(limit > 0.f)) return C3D_E_BADARG;
  clamp_kernel<<<grid_for(n), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(g, n, limit);
  C3D_CHECK_LAUNCH();
  return 0;
}
