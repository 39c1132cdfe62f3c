// Elementwise / reduction kernels on channels-last tensors, 8-channel vectors per thread:
//   * res-block output: y = relu(bn_c(c) + shortcut)      (reference model/x3d.py:326-327 and the
//     stem's BN+ReLU, model/x3d.py:94-106) and its backward g = dy*(y>0) with the BN-backward sums
//   * the pieces of Encoder.enhance (reference model/trainer.py:71-108)
// Threads keep a FIXED channel vector (blockDim is a multiple of Cp/8 and so is the grid stride),
// so per-channel parameters and partial sums stay in registers.
#include "common.h"
#include "bn_fin.h"
#include <cstring>
#include <cstdlib>
#include <cstdlib>
#include "../../include/change3d_hip.h"

namespace {

__host__ __device__ inline int ew_block(int G) { return G * (256 / G); }

// shortcut modes
enum { SC_NONE = 0, SC_IDENTITY = 1, SC_BN = 2, SC_RAW = 3 };

template <typename T>
__global__ void block_out_fwd_kernel(const T* __restrict__ c, const float* __restrict__ ss_c,
                                     const T* __restrict__ sc, const float* __restrict__ ss_1, int sc_mode,
                                     T* __restrict__ y, int64_t nvec, int G, int C, const c3d_bn_fin fin_c,
                                     const c3d_bn_fin fin_1) {
  __shared__ float lss[4][256];   // scale_c | shift_c | scale_1 | shift_1 (consumer-side BatchNorm finalisation)
  const int v = threadIdx.x % G;
  const int Cp = G * 8;
  float a[8], b[8], a1[8], b1[8];
  if (fin_c.sums) {
    // every workgroup rebuilds the vectors from the producers' completed sums (csrc/bn_fin.h); workgroup 0 owns the
    // global outputs (scale/shift, mean/rstd for backward, running statistics)
    const bool owner = blockIdx.x == 0;
    if (owner && threadIdx.x == 0 && fin_c.nbt) *fin_c.nbt += 1;
    c3dfin::bn_consume(fin_c, C, Cp, 0, Cp, owner, lss[0], lss[1], threadIdx.x, blockDim.x);
    if (sc_mode == SC_BN) {
      if (owner && threadIdx.x == 0 && fin_1.nbt) *fin_1.nbt += 1;
      c3dfin::bn_consume(fin_1, C, Cp, 0, Cp, owner, lss[2], lss[3], threadIdx.x, blockDim.x);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = lss[0][v * 8 + j]; b[j] = lss[1][v * 8 + j];
      a1[j] = (sc_mode == SC_BN) ? lss[2][v * 8 + j] : 1.f;
      b1[j] = (sc_mode == SC_BN) ? lss[3][v * 8 + j] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = ss_c[v * 8 + j]; b[j] = ss_c[Cp + v * 8 + j];
      a1[j] = (sc_mode == SC_BN) ? ss_1[v * 8 + j] : 1.f;
      b1[j] = (sc_mode == SC_BN) ? ss_1[Cp + v * 8 + j] : 0.f;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float f[8], s[8];
    Vec8<T>::load(c + i * 8, f);
    if (sc_mode != SC_NONE) Vec8<T>::load(sc + i * 8, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = fmaf(f[j], a[j], b[j]);
      if (sc_mode != SC_NONE) r += fmaf(s[j], a1[j], b1[j]);
      f[j] = fmaxf(r, 0.f);
    }
    Vec8<T>::store(y + i * 8, f);
  }
}

// g = dy * (y > 0); dsums_c += (sum g, sum g*chat); dsums_1 += (sum g, sum g*shat) when the shortcut
// has BN, with chat = (c - mean)*rstd accumulated centred (mr = mean[Cp], rstd[Cp]).
// PRE: `dy` is already dy * (y > 0) (masked by its producer, c3d_pw_args.wg_mask_out): y is not read, g is not written -- the
// pass only produces the BatchNorm-backward sums.  (A compile-time switch: the same test at run time cost the unmasked form
// 12 % -- 1.57 -> 1.76 ms per step.)
template <typename T, bool PRE>
__global__ __launch_bounds__(256) void block_out_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ c,
                                     const T* __restrict__ s, T* __restrict__ g, const float* __restrict__ mr_c,
                                     const float* __restrict__ mr_1, double* __restrict__ dsums_c,
                                     double* __restrict__ dsums_1, int64_t nvec, int G, int C, const c3d_bn_fin fin_c,
                                     const c3d_bn_fin fin_1) {
  extern __shared__ float red[];  // [blockDim][24]
  const int v = threadIdx.x % G;
  const int Cp = G * 8;
  // per-thread partial sums in f32 (a thread's grid-stride chain is 12-200 terms: its rounding error is ~1e-5 of ONE
  // term and random across the ~1e5 threads), everything across threads in f64 below -- the near-cancellation of
  // (sum g*chat) happens between threads, not inside one.  (Per-element v_cvt_f64_f32 + v_add_f64 triples -- f64 VALU runs
  // at a fraction of the f32 rate -- plus 48 accumulator registers made this elementwise pass run at 3.7 TB/s.)
  float s1[8], s2[8], s3[8];
  float mc[8], rc[8], m1[8], r1[8];
  {
    // eight 16-byte loads in ONE round trip (the rows are 32-byte aligned: Cp is a multiple of 8, the vectors come from the
    // stage workspace).  As 32 scalar loads with `s ? mr_1[..] : 0` selects the compiler put s_waitcnt vmcnt(0) behind every
    // pair: eight dependent memory round trips, ~8 us of a 27 us launch on the 32 x 32 maps (round 5, from the ISA)
    const float* q1 = s ? mr_1 : mr_c;   // a valid address either way: the loads are unconditional
    const float4 a0 = *reinterpret_cast<const float4*>(mr_c + v * 8), a1 = *reinterpret_cast<const float4*>(mr_c + v * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(mr_c + Cp + v * 8), b1 = *reinterpret_cast<const float4*>(mr_c + Cp + v * 8 + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(q1 + v * 8), c1 = *reinterpret_cast<const float4*>(q1 + v * 8 + 4);
    const float4 d0 = *reinterpret_cast<const float4*>(q1 + Cp + v * 8), d1 = *reinterpret_cast<const float4*>(q1 + Cp + v * 8 + 4);
    const float va[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float vc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, vd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s1[j] = 0.f; s2[j] = 0.f; s3[j] = 0.f;
      mc[j] = va[j]; rc[j] = vb[j];
      m1[j] = s ? vc[j] : 0.f; r1[j] = s ? vd[j] : 0.f;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // The grid is capped (the closing same-address atomics): ~1.5 workgroups per CU, so the bytes in flight come from the
  // loop itself -- BOB_U iterations' loads (raw vectors) are issued before the first is consumed; a thread's terms are
  // added in the order of the plain loop (bit-identical sums).
  constexpr int BOB_U = 4;
  typedef typename Vec8<T>::raw_t raw_t;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += BOB_U * stride) {
    raw_t rd[BOB_U], ry[BOB_U], rc_[BOB_U], rs[BOB_U];
#pragma unroll
    for (int u = 0; u < BOB_U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
        rd[u] = Vec8<T>::load_raw(dy + i * 8);
        if (!PRE) ry[u] = Vec8<T>::load_raw(y + i * 8);
        rc_[u] = Vec8<T>::load_raw(c + i * 8);
        if (s) rs[u] = Vec8<T>::load_raw(s + i * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < BOB_U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
        float d[8], yv[8], cv[8], sv[8];
        Vec8<T>::cvt_raw(rd[u], d);
        if (!PRE) Vec8<T>::cvt_raw(ry[u], yv);
        Vec8<T>::cvt_raw(rc_[u], cv);
        if (s) Vec8<T>::cvt_raw(rs[u], sv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gg = (PRE || yv[j] > 0.f) ? d[j] : 0.f;
          d[j] = gg;
          s1[j] += gg; s2[j] += gg * ((cv[j] - mc[j]) * rc[j]);
          if (s) s3[j] += gg * ((sv[j] - m1[j]) * r1[j]);
        }
        if (!PRE) Vec8<T>::store(g + i * 8, d);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[threadIdx.x * 24 + j] = s1[j]; red[threadIdx.x * 24 + 8 + j] = s2[j]; red[threadIdx.x * 24 + 16 + j] = s3[j];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < G * 24; idx += blockDim.x) {
    const int vv = idx / 24, k = idx % 24;
    double acc = 0;
    for (int t = vv; t < (int)blockDim.x; t += G) acc += red[t * 24 + k];
    const int ch = vv * 8 + (k & 7);
    const int which = k >> 3;
    if (ch < C) {
      if (which == 0) { atomicAdd(dsums_c + ch, acc); if (dsums_1) atomicAdd(dsums_1 + ch, acc); }
      else if (which == 1) atomicAdd(dsums_c + C + ch, acc);
      else if (dsums_1) atomicAdd(dsums_1 + C + ch, acc);
    }
  }
  if (fin_c.ticket) {   // last workgroup: BatchNorm_c (and shortcut BatchNorm) backward coefficients
    if (c3dfin::last_workgroup(fin_c.ticket, gridDim.x, reinterpret_cast<int*>(red))) {
      c3dfin::bn_backward(fin_c, dsums_c, 1, C, Cp, threadIdx.x, blockDim.x & ~15);
      if (dsums_1 && fin_1.ss) c3dfin::bn_backward(fin_1, dsums_1, 1, C, Cp, threadIdx.x, blockDim.x & ~15);
    }
  }
}

// ---- enhance pieces -----------------------------------------------------------------------
// d[m][c] = | y[b, t_pre, p, c] - y[b, t_post, p, c] |   (dense [B*HW][Cp])
template <typename T>
__global__ void frame_absdiff_kernel(const T* __restrict__ y, T* __restrict__ d, int64_t nvec, int64_t hwv,
                                     int Tn, int t_pre, int t_post) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int64_t b = i / hwv, r = i - b * hwv;
    float p[8], q[8];
    Vec8<T>::load(y + ((b * Tn + t_pre) * hwv + r) * 8, p);
    Vec8<T>::load(y + ((b * Tn + t_post) * hwv + r) * 8, q);
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = fabsf(p[j] - q[j]);
    Vec8<T>::store(d + i * 8, p);
  }
}

// out = copy(y); out[:, t_mid] += relu(e)      (e dense [B*HW][Cp])
template <typename T>
__global__ void enhance_apply_kernel(const T* __restrict__ y, const T* __restrict__ e, T* __restrict__ out,
                                     int64_t nvec_all, int64_t hwv, int Tn, int t_mid) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_all; i += stride) {
    const int64_t bt = i / hwv, r = i - bt * hwv;
    const int64_t b = bt / Tn;
    const int t = (int)(bt - b * Tn);
    float f[8];
    Vec8<T>::load(y + i * 8, f);
    if (t == t_mid) {
      float ev[8];
      Vec8<T>::load(e + (b * hwv + r) * 8, ev);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += fmaxf(ev[j], 0.f);
    }
    Vec8<T>::store(out + i * 8, f);
  }
}

// de[m][c] = dout[b, t_mid, p, c] * (e[m][c] > 0)
template <typename T>
__global__ void enhance_bwd_mask_kernel(const T* __restrict__ dout, const T* __restrict__ e, T* __restrict__ de,
                                        int64_t nvec, int64_t hwv, int Tn, int t_mid) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int64_t b = i / hwv, r = i - b * hwv;
    float g[8], ev[8];
    Vec8<T>::load(dout + ((b * Tn + t_mid) * hwv + r) * 8, g);
    Vec8<T>::load(e + i * 8, ev);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = ev[j] > 0.f ? g[j] : 0.f;
    Vec8<T>::store(de + i * 8, g);
  }
}

// dy = copy(dout); dy[:, t_pre] += dd*sign(pre-post); dy[:, t_post] -= dd*sign(pre-post)
template <typename T>
__global__ void enhance_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                         const T* __restrict__ dd, T* __restrict__ dy, int64_t nvec_all,
                                         int64_t hwv, int Tn, int t_pre, int t_post) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_all; i += stride) {
    const int64_t bt = i / hwv, r = i - bt * hwv;
    const int64_t b = bt / Tn;
    const int t = (int)(bt - b * Tn);
    float f[8];
    Vec8<T>::load(dout + i * 8, f);
    if (t == t_pre || t == t_post) {
      float p[8], q[8], dv[8];
      Vec8<T>::load(y + ((b * Tn + t_pre) * hwv + r) * 8, p);
      Vec8<T>::load(y + ((b * Tn + t_post) * hwv + r) * 8, q);
      Vec8<T>::load(dd + (b * hwv + r) * 8, dv);
      const float sgn_self = (t == t_pre) ? 1.f : -1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float df = p[j] - q[j];
        const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
        f[j] += sgn_self * sg * dv[j];
      }
    }
    Vec8<T>::store(dy + i * 8, f);
  }
}

// Generic helpers ------------------------------------------------------------------------------
// dst[b, t, p, :] (frame of an NDHWC tensor) += / = src dense [B*HW][Cp]
template <typename T>
__global__ void frame_scatter_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t nvec, int64_t hwv,
                                     int Tn, int t_dst, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int64_t b = i / hwv, r = i - b * hwv;
    float f[8];
    Vec8<T>::load(src + i * 8, f);
    T* p = dst + ((b * Tn + t_dst) * hwv + r) * 8;
    if (accumulate) {
      float o[8];
      Vec8<T>::load(p, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += o[j];
    }
    Vec8<T>::store(p, f);
  }
}

inline int ew_grid(int64_t nvec, int block) {
  int64_t g = (nvec + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define EW_DISPATCH(dtype, CALL_F32, CALL_BF16) \
  if ((dtype) == C3D_DT_F32) { CALL_F32; }      \
  else if ((dtype) == C3D_DT_BF16) { CALL_BF16; } \
  else return C3D_E_BADARG;

namespace {
int block_out_fwd_launch(const void* c, const float* ss_c, const void* shortcut, const float* ss_1, int32_t sc_mode,
                         void* y, int64_t M, int32_t C, int32_t Cp, int32_t dtype, void* stream, const c3d_bn_fin* fin_c,
                         const c3d_bn_fin* fin_1) {
  const int G = Cp / 8, blk = ew_block(G);
  const int64_t nvec = M * G;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  c3d_bn_fin f0;
  std::memset(&f0, 0, sizeof(f0));
  const c3d_bn_fin fc = fin_c ? *fin_c : f0, f1 = fin_1 ? *fin_1 : f0;
  // grid stride must stay a multiple of G: blk is, so any grid works.  With the finalisation folded in every
  // workgroup starts with a ~1.5 us dependent prologue: ONE round of workgroups (4 per CU), each walking the rows.
  int grid = ew_grid(nvec, blk);
  static const int env_grid = c3d_env("C3D_BOF_GRID") ? atoi(c3d_env("C3D_BOF_GRID")) : 0;   // tuning knob
  if (fin_c && grid > (env_grid > 0 ? env_grid : 1024)) grid = env_grid > 0 ? env_grid : 1024;
  EW_DISPATCH(dtype,
              (block_out_fwd_kernel<float><<<grid, blk, 0, s>>>(
                  (const float*)c, ss_c, (const float*)shortcut, ss_1, sc_mode, (float*)y, nvec, G, C, fc, f1)),
              (block_out_fwd_kernel<bf16_t><<<grid, blk, 0, s>>>(
                  (const bf16_t*)c, ss_c, (const bf16_t*)shortcut, ss_1, sc_mode, (bf16_t*)y, nvec, G, C, fc, f1)));
  C3D_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int c3d_block_out_fwd(const void* c, const float* ss_c, const void* shortcut, const float* ss_1,
                                 int32_t sc_mode, void* y, int64_t M, int32_t Cp, int32_t dtype, void* stream) {
  if (!c || !ss_c || !y || M <= 0 || (Cp & 7) || Cp > 256) return C3D_E_BADARG;
  if (sc_mode != SC_NONE && !shortcut) return C3D_E_BADARG;
  if (sc_mode == SC_BN && !ss_1) return C3D_E_BADARG;
  return block_out_fwd_launch(c, ss_c, shortcut, ss_1, sc_mode, y, M, Cp, Cp, dtype, stream, nullptr, nullptr);
}

extern "C" int c3d_block_out_fwd_fin(const void* c, const c3d_bn_fin* fin_c, const void* shortcut,
                                     const c3d_bn_fin* fin_1, int32_t sc_mode, void* y, int64_t M, int32_t C,
                                     int32_t Cp, int32_t dtype, void* stream) {
  if (!c || !fin_c || !fin_c->sums || !fin_c->ss || !fin_c->training || !y || M <= 0 || (Cp & 7) || Cp > 256 || C > Cp)
    return C3D_E_BADARG;
  if (sc_mode != SC_NONE && !shortcut) return C3D_E_BADARG;
  if (sc_mode == SC_BN && (!fin_1 || !fin_1->sums || !fin_1->ss)) return C3D_E_BADARG;
  return block_out_fwd_launch(c, fin_c->ss, shortcut, fin_1 ? fin_1->ss : nullptr, sc_mode, y, M, C, Cp, dtype, stream,
                              fin_c, sc_mode == SC_BN ? fin_1 : nullptr);
}

extern "C" int c3d_block_out_bwd(const void* dy, const void* y, const void* c, const void* s_bn, void* g,
                                 const float* mr_c, const float* mr_1, double* dsums_c, double* dsums_1, int64_t M,
                                 int32_t C, int32_t Cp, int32_t dtype, void* stream) {
  return c3d_block_out_bwd_fin(dy, y, c, s_bn, g, mr_c, mr_1, dsums_c, dsums_1, M, C, Cp, dtype, nullptr, nullptr, stream);
}

extern "C" int c3d_block_out_bwd_fin(const void* dy, const void* y, const void* c, const void* s_bn, void* g,
                                     const float* mr_c, const float* mr_1, double* dsums_c, double* dsums_1, int64_t M,
                                     int32_t C, int32_t Cp, int32_t dtype, const c3d_bn_fin* fc, const c3d_bn_fin* f1,
                                     void* stream) {
  if (!dy || !c || !mr_c || !dsums_c || M <= 0 || (Cp & 7) || Cp > 256) return C3D_E_BADARG;
  if (((uintptr_t)mr_c & 15) || (s_bn && ((uintptr_t)mr_1 & 15))) return C3D_E_BADARG;   // mean | rstd rows are read as 16-byte vectors
  if ((y == nullptr) != (g == nullptr)) return C3D_E_BADARG;   // both NULL: dy is already masked, only the sums are produced
  if ((s_bn == nullptr) != (dsums_1 == nullptr) || (s_bn && !mr_1)) return C3D_E_BADARG;
  if (fc && fc->ticket && (!fc->gamma || !fc->ss || !fc->mr || !(fc->count > 0))) return C3D_E_BADARG;
  if (fc && fc->ticket && s_bn && (!f1 || !f1->gamma || !f1->ss || !f1->mr || !(f1->count > 0))) return C3D_E_BADARG;
  const c3d_bn_fin fin_c = fc ? *fc : c3d_bn_fin{};
  const c3d_bn_fin fin_1 = (f1 && s_bn) ? *f1 : c3d_bn_fin{};
  const int G = Cp / 8, blk = ew_block(G);
  const int64_t nvec = M * G;
  int grid = ew_grid(nvec, blk);
  static const int env_cap = c3d_env("C3D_BOB_GRID") ? atoi(c3d_env("C3D_BOB_GRID")) : 0;
  // every workgroup ends with an LDS reduction and G*24 same-address f64 atomics: measured on MI355X
  // 128/256/384/512/1024/2048 workgroups -> 2.55/1.70/1.59/1.68/2.22/3.07 ms per step
  const int cap = env_cap > 0 ? env_cap : 384;
  if (grid > cap) grid = cap;
  const size_t lds = (size_t)blk * 24 * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define BOB_LAUNCH(TT, PRE_)                                                                                                   \
  block_out_bwd_kernel<TT, PRE_><<<grid, blk, lds, st>>>((const TT*)dy, (const TT*)y, (const TT*)c, (const TT*)s_bn, (TT*)g, mr_c, \
                                                         mr_1, dsums_c, dsums_1, nvec, G, C, fin_c, fin_1)
  if (y) { EW_DISPATCH(dtype, (BOB_LAUNCH(float, false)), (BOB_LAUNCH(bf16_t, false))); }
  else { EW_DISPATCH(dtype, (BOB_LAUNCH(float, true)), (BOB_LAUNCH(bf16_t, true))); }
#undef BOB_LAUNCH
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_frame_absdiff(const void* y, void* d, int32_t B, int32_t T, int64_t HW, int32_t Cp,
                                 int32_t t_pre, int32_t t_post, int32_t dtype, void* stream) {
  if (!y || !d || B <= 0 || T <= 0 || HW <= 0 || (Cp & 7) || t_pre < 0 || t_pre >= T || t_post < 0 || t_post >= T)
    return C3D_E_BADARG;
  const int64_t hwv = HW * (Cp / 8), nvec = (int64_t)B * hwv;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EW_DISPATCH(dtype,
              (frame_absdiff_kernel<float><<<ew_grid(nvec, 256), 256, 0, s>>>((const float*)y, (float*)d, nvec, hwv,
                                                                               T, t_pre, t_post)),
              (frame_absdiff_kernel<bf16_t><<<ew_grid(nvec, 256), 256, 0, s>>>((const bf16_t*)y, (bf16_t*)d, nvec,
                                                                                hwv, T, t_pre, t_post)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_enhance_apply(const void* y, const void* e, void* out, int32_t B, int32_t T, int64_t HW,
                                 int32_t Cp, int32_t t_mid, int32_t dtype, void* stream) {
  if (!y || !e || !out || B <= 0 || T <= 0 || HW <= 0 || (Cp & 7) || t_mid < 0 || t_mid >= T) return C3D_E_BADARG;
  const int64_t hwv = HW * (Cp / 8), nvec = (int64_t)B * T * hwv;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EW_DISPATCH(dtype,
              (enhance_apply_kernel<float><<<ew_grid(nvec, 256), 256, 0, s>>>((const float*)y, (const float*)e,
                                                                               (float*)out, nvec, hwv, T, t_mid)),
              (enhance_apply_kernel<bf16_t><<<ew_grid(nvec, 256), 256, 0, s>>>((const bf16_t*)y, (const bf16_t*)e,
                                                                                (bf16_t*)out, nvec, hwv, T, t_mid)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_enhance_bwd_mask(const void* dout, const void* e, void* de, int32_t B, int32_t T, int64_t HW,
                                    int32_t Cp, int32_t t_mid, int32_t dtype, void* stream) {
  if (!dout || !e || !de || B <= 0 || T <= 0 || HW <= 0 || (Cp & 7) || t_mid < 0 || t_mid >= T) return C3D_E_BADARG;
  const int64_t hwv = HW * (Cp / 8), nvec = (int64_t)B * hwv;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EW_DISPATCH(dtype,
              (enhance_bwd_mask_kernel<float><<<ew_grid(nvec, 256), 256, 0, s>>>((const float*)dout, (const float*)e,
                                                                                  (float*)de, nvec, hwv, T, t_mid)),
              (enhance_bwd_mask_kernel<bf16_t><<<ew_grid(nvec, 256), 256, 0, s>>>(
                  (const bf16_t*)dout, (const bf16_t*)e, (bf16_t*)de, nvec, hwv, T, t_mid)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_enhance_bwd_apply(const void* dout, const void* y, const void* dd, void* dy, int32_t B,
                                     int32_t T, int64_t HW, int32_t Cp, int32_t t_pre, int32_t t_post,
                                     int32_t dtype, void* stream) {
  if (!dout || !y || !dd || !dy || B <= 0 || T <= 0 || HW <= 0 || (Cp & 7)) return C3D_E_BADARG;
  const int64_t hwv = HW * (Cp / 8), nvec = (int64_t)B * T * hwv;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EW_DISPATCH(dtype,
              (enhance_bwd_apply_kernel<float><<<ew_grid(nvec, 256), 256, 0, s>>>(
                  (const float*)dout, (const float*)y, (const float*)dd, (float*)dy, nvec, hwv, T, t_pre, t_post)),
              (enhance_bwd_apply_kernel<bf16_t><<<ew_grid(nvec, 256), 256, 0, s>>>(
                  (const bf16_t*)dout, (const bf16_t*)y, (const bf16_t*)dd, (bf16_t*)dy, nvec, hwv, T, t_pre,
                  t_post)));
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_frame_scatter(const void* src, void* dst, int32_t B, int32_t T, int64_t HW, int32_t Cp,
                                 int32_t t_dst, int32_t accumulate, int32_t dtype, void* stream) {
  if (!src || !dst || B <= 0 || T <= 0 || HW <= 0 || (Cp & 7) || t_dst < 0 || t_dst >= T) return C3D_E_BADARG;
  const int64_t hwv = HW * (Cp / 8), nvec = (int64_t)B * hwv;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EW_DISPATCH(dtype,
              (frame_scatter_kernel<float><<<ew_grid(nvec, 256), 256, 0, s>>>((const float*)src, (float*)dst, nvec,
                                                                               hwv, T, t_dst, accumulate)),
              (frame_scatter_kernel<bf16_t><<<ew_grid(nvec, 256), 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst,
                                                                                nvec, hwv, T, t_dst, accumulate)));
  C3D_CHECK_LAUNCH();
  return 0;
}
