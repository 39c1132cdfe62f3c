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

extern "C" int c3d_abi_version(void) { return 1; }
extern "C" const char* c3d_build_info(void) { return "change3d_hip gfx950 (MI355X) hipcc " __VERSION__; }
