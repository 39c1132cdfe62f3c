// Training-step shell kernels:
//   * BCEDiceLoss forward / backward (reference model/utils.py:154-169; sums run over the
//     whole batch; BCE log terms clamped at -100 and its gradient denominator at 1e-12 exactly
//     as ATen's binary_cross_entropy does on the reference's CPU path)
//   * Adam with coupled L2 weight decay over one flat f32 parameter buffer
//     (reference scripts/train_BCD.py:284-290: betas (0.9, 0.99), eps 1e-8, weight_decay 1e-4),
//     using torch.optim.Adam's operation order (lerp for exp_avg; sqrt(v)/sqrt(bc2) + eps).
//   * binarise + 2x2 confusion matrix on device (reference scripts/train_BCD.py:204-208,
//     utils/metric_tool.py:111-128) so the per-iteration metric needs no full D2H copy.
#include "common.h"
#include "../../include/change3d_hip.h"

namespace {

__global__ __launch_bounds__(256) void bce_dice_reduce_kernel(const float* __restrict__ p,
                                                              const float* __restrict__ t, int64_t n,
                                                              double* __restrict__ sums) {
  __shared__ double red[4][4];
  double s_b = 0, s_i = 0, s_p = 0, s_t = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pv = p[i], tv = t[i];
    const float lp = fmaxf(logf(pv), -100.f), l1p = fmaxf(logf(1.f - pv), -100.f);
    s_b += (double)((tv - 1.f) * l1p - tv * lp);
    s_i += (double)(pv * tv);
    s_p += (double)pv;
    s_t += (double)tv;
  }
  s_b = wave_sum_d(s_b); s_i = wave_sum_d(s_i); s_p = wave_sum_d(s_p); s_t = wave_sum_d(s_t);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s_b; red[wave][1] = s_i; red[wave][2] = s_p; red[wave][3] = s_t; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double a = 0;
    for (int w = 0; w < 4; ++w) a += red[w][threadIdx.x];
    atomicAdd(sums + threadIdx.x, a);
  }
}

__global__ void bce_dice_finalize_kernel(const double* __restrict__ sums, int64_t n, float* __restrict__ loss) {
  const float bce = (float)(sums[0] / (double)n);
  const float inter = (float)sums[1], sp = (float)sums[2], st = (float)sums[3];
  const float eps = 1e-5f;
  const float dice = (2.f * inter + eps) / (sp + st + eps);
  loss[0] = bce + 1.f - dice;
}

__global__ void bce_dice_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                    const double* __restrict__ sums, const float* __restrict__ dloss, int64_t n,
                                    float* __restrict__ dp) {
  const float gl = dloss ? dloss[0] : 1.f;
  const float inter = (float)sums[1], sp = (float)sums[2], st = (float)sums[3];
  const float eps = 1e-5f;
  const float D = sp + st + eps, Nn = 2.f * inter + eps;
  const float invD = 1.f / D, k = Nn * invD * invD;
  const float invn = 1.f / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pv = p[i], tv = t[i];
    const float gb = (pv - tv) / fmaxf((1.f - pv) * pv, 1e-12f) * invn;
    const float gd = 2.f * tv * invD - k;  // d dice / d p
    dp[i] = gl * (gb - gd);
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, const float* __restrict__ hp_dev, float lr, float bc1,
                            float bc2_sqrt, float beta1, float beta2, float eps, float wd) {
  if (hp_dev) { lr = hp_dev[0]; bc1 = hp_dev[1]; bc2_sqrt = hp_dev[2]; }
  const float step_size = lr / bc1;
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pv = p[i];
    float gv = g[i];
    if (wd != 0.f) gv = gv + wd * pv;
    float mv = m[i];
    mv = mv + w1 * (gv - mv);
    const float vv = v[i] * beta2 + w2 * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv + (-step_size) * (mv / denom);
  }
}

// cm[2*gt + pred] += 1 with pred = prob > 0.5 (strict), gt in {0,1}
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ prob, const float* __restrict__ gt,
                                                        int64_t n, unsigned long long* __restrict__ cm) {
  __shared__ unsigned int red[4];
  if (threadIdx.x < 4) red[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[4] = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int pr = prob[i] > 0.5f ? 1 : 0;
    const int g = (int)gt[i];
    if (g >= 0 && g < 2) c[2 * g + pr]++;
  }
  for (int k = 0; k < 4; ++k) {
    unsigned int v = c[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(cm + threadIdx.x, (unsigned long long)red[threadIdx.x]);
}

inline int grid_for(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


// ---- SCD losses (reference model/utils.py:171-203, scripts/train_SCD.py:226-229) ----
// Logits are NCHW f32 views: element (b, c, p) at x[b*bstride + c*cstride + p], p < HW.
constexpr int SCD_MAXC = 8;

// CrossEntropyLoss2d: nll_loss(log_softmax(x, 1), target, ignore_index, 'mean').  sums = (sum nll, count).
__global__ __launch_bounds__(256) void ce2d_reduce_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                                          int64_t npix, int64_t HW, int64_t bstride, int64_t cstride,
                                                          int NC, int64_t ignore_index, double* __restrict__ sums) {
  __shared__ double red[4][2];
  double s_l = 0, s_n = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t t = tgt[i];
    if (t == ignore_index) continue;
    const int64_t b = i / HW, p = i - b * HW;
    const float* xp = x + b * bstride + p;
    float v[SCD_MAXC], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      v[c] = c < NC ? xp[c * cstride] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float se = 0.f, vt = 0.f;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      if (c < NC) se += expf(v[c] - m);
      if (c == (int)t) vt = v[c];
    }
    s_l += (double)(m + logf(se) - vt);
    s_n += 1.0;
  }
  s_l = wave_sum_d(s_l); s_n = wave_sum_d(s_n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s_l; red[wave][1] = s_n; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double a = 0;
    for (int w = 0; w < 4; ++w) a += red[w][threadIdx.x];
    atomicAdd(sums + threadIdx.x, a);
  }
}

__global__ void ce2d_finalize_kernel(const double* __restrict__ sums, float* __restrict__ loss) {
  loss[0] = (float)(sums[0] / sums[1]);   // 0/0 = NaN, as nll_loss gives when every pixel is ignored
}

__global__ __launch_bounds__(256) void ce2d_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                                       const double* __restrict__ sums, const float* __restrict__ dloss,
                                                       int64_t npix, int64_t HW, int64_t bstride, int64_t cstride, int NC,
                                                       int64_t ignore_index, float* __restrict__ dx) {
  const float scale = (dloss ? dloss[0] : 1.f) / (float)sums[1];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t t = tgt[i];
    const int64_t b = i / HW, p = i - b * HW;
    float* dp = dx + b * NC * HW + p;   // dx is a dense [B][NC][HW] tensor
    if (t == ignore_index) {
      for (int c = 0; c < NC; ++c) dp[c * HW] = 0.f;
      continue;
    }
    const float* xp = x + b * bstride + p;
    float v[SCD_MAXC], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      v[c] = c < NC ? xp[c * cstride] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      v[c] = c < NC ? expf(v[c] - m) : 0.f;
      se += v[c];
    }
    const float inv = 1.f / se;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c)
      if (c < NC) dp[c * HW] = scale * (v[c] * inv - (c == (int)t ? 1.f : 0.f));
  }
}

// ChangeSimilarity: cosine_embedding_loss(softmax(x1), softmax(x2), +1 unchanged / -1 changed, margin 0, 'mean')
// with ATen's formula cos = <p1,p2> / sqrt((|p1|^2 + 1e-12)(|p2|^2 + 1e-12)).
__device__ __forceinline__ void softmax_c(const float* xp, int64_t cstride, int NC, float (&p)[SCD_MAXC]) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < SCD_MAXC; ++c) {
    p[c] = c < NC ? xp[c * cstride] : -INFINITY;
    m = fmaxf(m, p[c]);
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < SCD_MAXC; ++c) {
    p[c] = c < NC ? expf(p[c] - m) : 0.f;
    se += p[c];
  }
  const float inv = 1.f / se;
#pragma unroll
  for (int c = 0; c < SCD_MAXC; ++c) p[c] *= inv;
}

__global__ __launch_bounds__(256) void cossim_reduce_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                            const int64_t* __restrict__ change, int64_t npix, int64_t HW,
                                                            int64_t bs1, int64_t cs1, int64_t bs2, int64_t cs2, int NC,
                                                            double* __restrict__ sums) {
  __shared__ double red[4];
  double s = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t b = i / HW, p = i - b * HW;
    float p1[SCD_MAXC], p2[SCD_MAXC];
    softmax_c(x1 + b * bs1 + p, cs1, NC, p1);
    softmax_c(x2 + b * bs2 + p, cs2, NC, p2);
    float dot = 0.f, n1 = 1e-12f, n2 = 1e-12f;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) { dot = fmaf(p1[c], p2[c], dot); n1 = fmaf(p1[c], p1[c], n1); n2 = fmaf(p2[c], p2[c], n2); }
    const float cs = dot / sqrtf(n1 * n2);
    s += (double)(change[i] != 0 ? fmaxf(cs, 0.f) : 1.f - cs);
  }
  s = wave_sum_d(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sums, red[0] + red[1] + red[2] + red[3]);
}

__global__ void cossim_finalize_kernel(const double* __restrict__ sums, int64_t npix, float* __restrict__ loss) {
  loss[0] = (float)(sums[0] / (double)npix);
}

__global__ __launch_bounds__(256) void cossim_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                         const int64_t* __restrict__ change, const float* __restrict__ dloss,
                                                         int64_t npix, int64_t HW, int64_t bs1, int64_t cs1, int64_t bs2,
                                                         int64_t cs2, int NC, float* __restrict__ dx1, float* __restrict__ dx2) {
  const float scale = (dloss ? dloss[0] : 1.f) / (float)npix;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const int64_t b = i / HW, p = i - b * HW;
    float p1[SCD_MAXC], p2[SCD_MAXC];
    softmax_c(x1 + b * bs1 + p, cs1, NC, p1);
    softmax_c(x2 + b * bs2 + p, cs2, NC, p2);
    float dot = 0.f, n1 = 1e-12f, n2 = 1e-12f;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) { dot = fmaf(p1[c], p2[c], dot); n1 = fmaf(p1[c], p1[c], n1); n2 = fmaf(p2[c], p2[c], n2); }
    const float den = sqrtf(n1 * n2), cs = dot / den;
    // d loss / d cos: -1 where unchanged; where changed the hinge max(cos, 0) passes 1 for cos > 0
    const float gcos = scale * (change[i] != 0 ? (cs > 0.f ? 1.f : 0.f) : -1.f);
    float g1[SCD_MAXC], g2[SCD_MAXC], a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      g1[c] = gcos * (p2[c] / den - cs * p1[c] / n1);   // d loss / d p1
      g2[c] = gcos * (p1[c] / den - cs * p2[c] / n2);
      a1 = fmaf(g1[c], p1[c], a1);
      a2 = fmaf(g2[c], p2[c], a2);
    }
    float* d1 = dx1 + b * NC * HW + p;
    float* d2 = dx2 + b * NC * HW + p;
#pragma unroll
    for (int c = 0; c < SCD_MAXC; ++c) {
      if (c < NC) {
        d1[c * HW] = p1[c] * (g1[c] - a1);            // softmax backward
        d2[c * HW] = p2[c] * (g2[c] - a2);
      }
    }
  }
}

}  // namespace

extern "C" int c3d_bce_dice_fwd(const float* prob, const float* target, int64_t n, double* sums4, float* loss,
                                void* stream) {
  if (!prob || !target || !sums4 || !loss || n <= 0) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums4, 0, 4 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  bce_dice_reduce_kernel<<<grid_for(n, 256, 1024), 256, 0, s>>>(prob, target, n, sums4);
  C3D_CHECK_LAUNCH();
  bce_dice_finalize_kernel<<<1, 1, 0, s>>>(sums4, n, loss);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bce_dice_bwd(const float* prob, const float* target, const double* sums4, const float* dloss,
                                int64_t n, float* dprob, void* stream) {
  if (!prob || !target || !sums4 || !dprob || n <= 0) return C3D_E_BADARG;
  bce_dice_bwd_kernel<<<grid_for(n, 256, 4096), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
      prob, target, sums4, dloss, n, dprob);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const float* hparams_dev, float lr, float bias_correction1, float bias_correction2_sqrt,
                             float beta1, float beta2, float eps, float weight_decay, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return C3D_E_BADARG;
  adam_kernel<<<grid_for(n, 256, 4096), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
      param, grad, exp_avg, exp_avg_sq, n, hparams_dev, lr, bias_correction1, bias_correction2_sqrt, beta1, beta2,
      eps, weight_decay);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_confusion2(const float* prob, const float* target, int64_t n, unsigned long long* cm4,
                              void* stream) {
  if (!prob || !target || !cm4 || n <= 0) return C3D_E_BADARG;
  confusion_kernel<<<grid_for(n, 256, 1024), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(prob, target, n, cm4);
  C3D_CHECK_LAUNCH();
  return 0;
}

// n x n joint histogram hist[a][b] of two integer label maps (SCD validation: prediction x label, reference
// model/utils.py:313-345 fast_hist / get_hist: np.bincount(n * a[k] + b[k]) over k = (0 <= a < n)).  Workgroup-local
// counts in LDS, one 64-bit atomic per bin and workgroup.  Pairs with b outside [0, n) -- numpy's reshape would raise
// there -- are counted in hist[n*n] so that the host wrapper can raise too.
constexpr int HIST_MAXN = 16;
__global__ __launch_bounds__(256) void hist2d_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n_elems,
                                                     int n, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[HIST_MAXN * HIST_MAXN + 1];
  for (int i = threadIdx.x; i <= n * n; i += blockDim.x) h[i] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t av = a[i], bv = b[i];
    if (av >= 0 && av < n) atomicAdd(&h[(bv >= 0 && bv < n) ? (int)(av * n + bv) : n * n], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= n * n; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

extern "C" int c3d_hist2d(const int64_t* a, const int64_t* b, int64_t n_elems, int32_t n, unsigned long long* hist,
                          void* stream) {
  if (!a || !b || !hist || n_elems <= 0 || n < 1 || n > HIST_MAXN) return C3D_E_BADARG;
  // (a workgroup's LDS counters are 32-bit: at most 2^32 - 1 elements per workgroup; 1024 workgroups cover 2^42)
  hist2d_kernel<<<grid_for(n_elems, 256, 1024), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(a, b, n_elems, n, hist);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_abi_version(void) { return 1; }
extern "C" const char* c3d_build_info(void) { return "change3d_hip gfx950 (MI355X) hipcc " __VERSION__; }

extern "C" int c3d_ce2d_fwd(const float* logits, const int64_t* target, int64_t B, int32_t NC, int64_t HW,
                            int64_t bstride, int64_t cstride, int64_t ignore_index, double* sums2, float* loss,
                            void* stream) {
  if (!logits || !target || !sums2 || !loss || B <= 0 || HW <= 0 || NC < 1 || NC > SCD_MAXC) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums2, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  const int64_t n = B * HW;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  ce2d_reduce_kernel<<<grid, 256, 0, s>>>(logits, target, n, HW, bstride, cstride, NC, ignore_index, sums2);
  ce2d_finalize_kernel<<<1, 1, 0, s>>>(sums2, loss);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_ce2d_bwd(const float* logits, const int64_t* target, const double* sums2, const float* dloss,
                            int64_t B, int32_t NC, int64_t HW, int64_t bstride, int64_t cstride,
                            int64_t ignore_index, float* dlogits, void* stream) {
  if (!logits || !target || !sums2 || !dlogits || B <= 0 || HW <= 0 || NC < 1 || NC > SCD_MAXC) return C3D_E_BADARG;
  const int64_t n = B * HW;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  ce2d_bwd_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(logits, target, sums2, dloss, n, HW, bstride,
                                                                          cstride, NC, ignore_index, dlogits);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cossim_fwd(const float* x1, const float* x2, const int64_t* label_change, int64_t B, int32_t NC,
                              int64_t HW, int64_t bstride1, int64_t cstride1, int64_t bstride2, int64_t cstride2,
                              double* sums1, float* loss, void* stream) {
  if (!x1 || !x2 || !label_change || !sums1 || !loss || B <= 0 || HW <= 0 || NC < 1 || NC > SCD_MAXC) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(sums1, 0, sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  const int64_t n = B * HW;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  cossim_reduce_kernel<<<grid, 256, 0, s>>>(x1, x2, label_change, n, HW, bstride1, cstride1, bstride2, cstride2, NC, sums1);
  cossim_finalize_kernel<<<1, 1, 0, s>>>(sums1, n, loss);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cossim_bwd(const float* x1, const float* x2, const int64_t* label_change, const float* dloss,
                              int64_t B, int32_t NC, int64_t HW, int64_t bstride1, int64_t cstride1, int64_t bstride2,
                              int64_t cstride2, float* dx1, float* dx2, void* stream) {
  if (!x1 || !x2 || !label_change || !dx1 || !dx2 || B <= 0 || HW <= 0 || NC < 1 || NC > SCD_MAXC) return C3D_E_BADARG;
  const int64_t n = B * HW;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  cossim_bwd_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x1, x2, label_change, dloss, n, HW, bstride1,
                                                                            cstride1, bstride2, cstride2, NC, dx1, dx2);
  C3D_CHECK_LAUNCH();
  return 0;
}
