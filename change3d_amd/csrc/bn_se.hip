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
  float* w1s = h + (size_t)B * Cr;      // [Cr][C]  FC weights staged once, coalesced (the FC loops read them
  float* w2s = w1s + (size_t)Cr * C;    // [C][Cr]   element by element from global memory before)
  const int tid = threadIdx.x;
  if (w1) {
    for (int i = tid; i < Cr * C; i += blockDim.x) { w1s[i] = w1[i]; w2s[i] = w2[i]; }
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
  for (int idx = tid; idx < B * Cr * 8; idx += blockDim.x) {  // 8 lanes per (sample, hidden unit)
    const int i = idx >> 3, q = idx & 7;
    const int n = i / Cr, r = i - n * Cr;
    float a = 0.f;
    for (int c = q; c < C; c += 8) a = fmaf(w1s[(size_t)r * C + c], z[(size_t)n * C + c], a);
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
    a += b1[r];
    a = a > 0.f ? a : 0.f;
    if (q == 0) { h[i] = a; if (hid && blockIdx.x == 0) hid[i] = a; }
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
  const double count = cnt_per_sample * B;
  const bool se = w1 != nullptr;
  // channel-sliced over gridDim.x workgroups: d gate and the FC2-backward hidden gradient need every
  // channel and are recomputed by each workgroup; everything written is owned by slice [c_lo, c_hi)
  const int cs = (Cp + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c_lo = (int)blockIdx.x * cs, c_hi = c_lo + cs < Cp ? c_lo + cs : Cp;
  const int cw = c_hi - c_lo;
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
    __syncthreads();
    for (int idx = tid; idx < B * Cr * 8; idx += blockDim.x) {  // 8 lanes per (sample, hidden unit)
      const int i = idx >> 3, q = idx & 7;
      const int n = i / Cr, r = i - n * Cr;
      float a = 0.f;
      for (int c = q; c < C; c += 8) a = fmaf(wst[(size_t)c * Cr + r], du[(size_t)n * C + c], a);
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
      if (q == 0) dh[i] = hids[i] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < Cr * C; i += blockDim.x) wst[i] = w1[i];   // [Cr][C]
    __syncthreads();
    for (int i = tid; i < B * cw; i += blockDim.x) {
      const int n = i / cw, c = c_lo + (i - n * cw);
      if (c >= C) continue;
      float a = 0.f;
      for (int r = 0; r < Cr; ++r) a = fmaf(wst[(size_t)r * C + c], dh[(size_t)n * Cr + r], a);
      dz[(size_t)n * C + c] = a;
    }
    // parameter gradients of the two FCs
    for (int i = tid; i < cw * Cr; i += blockDim.x) {
      const int c = c_lo + i / Cr, r = i % Cr;  // w2[c][r]
      if (c >= C) continue;
      float a = 0.f, b = 0.f;
      for (int n = 0; n < B; ++n) {
        a = fmaf(du[(size_t)n * C + c], hids[(size_t)n * Cr + r], a);
        b = fmaf(dh[(size_t)n * Cr + r], zz[(size_t)n * C + c], b);
      }
      dw2[(size_t)c * Cr + r] += a;
      dw1[(size_t)r * C + c] += b;
    }
    for (int c = c_lo + tid; c < c_hi && c < C; c += blockDim.x) {
      float a = 0.f;
      for (int n = 0; n < B; ++n) a += du[(size_t)n * C + c];
      db2[c] += a;
    }
    if (blockIdx.x == 0) {
      for (int r = tid; r < Cr; r += blockDim.x) {
        float a = 0.f;
        for (int n = 0; n < B; ++n) a += dh[(size_t)n * Cr + r];
        db1[r] += a;
      }
    }
    __syncthreads();
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
#pragma unroll 4
    for (int n = q; n < B; n += 4) {
      const double dzn = se ? (double)dz[(size_t)n * C + c] : 0.0;
      s1 += nc3[((size_t)n * Cp + c) * 3 + 1] + dzn;
      // the uniform SE term dz/cnt multiplies sum_thw bhat = rstd * (sum b - cnt*mean)
      s2 += nc3[((size_t)n * Cp + c) * 3 + 2] +
            dzn / cnt_per_sample * rstd * (ncf[((size_t)n * Cp + c) * 2] - cnt_per_sample * mean);
    }
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    const double A = (double)gamma[c] * rstd;
    const double Cc = -A * rstd * s2 / count;
    if (q == 0) { coefA[c] = (float)A; coefC[c] = (float)Cc; }
    for (int n = q; n < B; n += 4) {
      const double dzn = se ? (double)dz[(size_t)n * C + c] : 0.0;
      coefB[(size_t)n * Cp + c] = (float)(A * dzn / cnt_per_sample - A * s1 / count - Cc * mean);
    }
    if (q == 0) {
      if (dgamma) dgamma[c] += (float)s2;
      if (dbeta) dbeta[c] += (float)s1;
    }
  }
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
  const size_t lds = w1 ? ((size_t)B * C + (size_t)B * Cr + (size_t)2 * C * Cr) * sizeof(float) : 0;
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
  const size_t lds = w1 ? ((size_t)3 * B * C + (size_t)2 * B * Cr + (size_t)C * Cr) * sizeof(float) : 0;
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
