// Depthwise 3x3x3 forward on the matrix cores (bf16 storage, T <= 3, stride 1): reference model/x3d.py:184-193
// (conv_b of the bottleneck, BN_a + ReLU applied on load), interface and the VALU kernels: dw_conv.hip.
//
// A depthwise convolution has no contraction over channels, but per channel the three taps along x are a banded
// (Toeplitz) matrix: for a 16-wide block of output columns
//     D[x_out][y] += Tz(w[kt][ky][.])[x_out][x_in] * In[x_in][y + ky - 1]          (x_in: 18 of a 32-wide K)
// is one v_mfma_f32_16x16x32_bf16 per (kt, ky) pair -- 21 MFMAs per 16x16 block and 3 frames -- at 9 % useful MACs,
// which is still several times the f32 VALU rate the stencil kernels are bound by (27 FMA + LDS reads per output).
//
//   * The Toeplitz fragments depend only on the channel, not on the position: a wave owns 4 channels for the WHOLE
//     launch and keeps their 4 x 9 fragments in registers (144 VGPRs; the kernel runs one wave per SIMD, LDS-limited).
//   * A workgroup = 16 channels x one 32x32 spatial tile (+halo) x 3 frames, walks tiles / samples.  The tile lives in
//     LDS as per-channel bf16 planes [t][34][40]: the B operand of a lane is ONE ds_read_b128 (8 consecutive x of row y).
//   * channels-last <-> planes: staging loads 8 pixels x 8 channels per thread and packs per channel (the transpose is
//     which values share a cvt_pk), one ds_write_b128 per channel; the epilogue writes each channel's results over its
//     own (dead) input plane and a gather pass transposes 8x8 blocks back with v_perm for 16-byte global stores.
//   * activations AND weights enter the MFMA as bf16 (as in the pointwise GEMMs); accumulation is f32.
//
// STATUS (round 2): EXPERIMENT, off by default (C3D_DW_TZ=1 routes c3d_dw333_fwd here).  Numerically it passes the
// depthwise operator tests; it is SLOWER than the VALU kernel: 229 / 129 / 90 us against 163 / 79 / 43 us for the
// 128x128x54 / 64x64x108 / 32x32x216 stages at B=32.  Phase timing (C3D_DW_TZ_DBG compiles phases out of the walk),
// per 32x32x16-channel tile: staging 11 us, MFMA + statistics 8.4 us, gather 7.5 us = 27 us for 49 k outputs, against
// ~20 us for the same outputs in the VALU kernel (two workgroups per CU overlapping each other).  The matrix time is
// not the problem (336 MFMAs = 2.2 us); with ONE wave per SIMD (131 KB of planes per workgroup) nothing overlaps the
// two transposes, and batching the loads / LDS reads per phase did not move them.  C3D_DW_TZ_NCH=8 (8-channel planes,
// 65 KB, TWO workgroups per CU) measures 344 / 127 / 76 us: occupancy is not the cure either (and 16-byte pieces of
// each pixel row cost the 128x128 stage dearly).  Next: in-kernel clocks on the three phases.
#include "common.h"
#include "dw_toeplitz.h"
#include <cstdlib>
#include <cstring>
#include "../../include/change3d_hip.h"

namespace {

constexpr int TS = 32;                 // spatial tile side (outputs)
constexpr int PH = TS + 2, PW = 40;    // plane rows (halo), row stride in elements (34 used; 80 B keeps 16-B alignment)
constexpr int TT = 3;
constexpr int PLANE_E = TT * PH * PW;  // 4080 elements
constexpr int PLANE_B = PLANE_E * 2 + 16;   // bytes, +16: the two channel octets of a pixel fall on different banks
constexpr int NTHR = 256;
constexpr int OW = TS;                 // output plane row stride (elements)

struct Geom { int B, T, H, W, C, Cp; };

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ f32x4_t mfma_bf16(const u32x4 a, const u32x4 b, const f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NCH>
__global__ __launch_bounds__(NTHR) void dw_fwd_tz_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ss,
                                                         const float* __restrict__ w, bf16_t* __restrict__ y,
                                                         double* __restrict__ nc, const Geom g, const int walkers, const int dbg,
                                                         unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* planes = smem;                                        // [NCH][PLANE_B]
  float* lss = reinterpret_cast<float*>(smem + NCH * PLANE_B + 64);    // scale[NCH] | shift[NCH]
  constexpr int CPW = NCH / 4;       // channels per wave
  constexpr int NV = NCH / 8;        // channel octets per pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, gq = lane >> 4;
  // phase clocks (C3D_DW_TZ_CLK=1): s_memtime deltas summed per wave, added to clk[phase] at the end
  unsigned long long ck[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ck_last = clk ? __builtin_amdgcn_s_memtime() : 0;
#define TZCK(i) if (clk) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ck[i] += t_ - ck_last; ck_last = t_; }
  const int c0 = blockIdx.y * NCH;
  const int walker = blockIdx.x;
  // (an XCD-aware 1-D order that puts the channel groups of one walker on one L2 was measured: no gain for the 128x128
  //  stage, and the walker count rounded to whole XCD rows left CUs idle for the others)
  if (tid < 2 * NCH) {
    const int c = c0 + (tid % NCH);
    lss[tid] = c < g.C ? ss[(tid / NCH) * g.Cp + c] : 0.f;
  }
  // (the 16 bytes behind each plane are read -- never used, their Toeplitz coefficients are zero -- and zeroed below)

  // weights of the 16 channels through LDS (one coalesced pass; the fragment build below reads them as broadcasts)
  float* wl = reinterpret_cast<float*>(planes);     // [16][27], the planes are not live yet
  for (int i = tid; i < NCH * 27; i += NTHR) {
    const int c = c0 + i / 27;
    wl[i] = c < g.C ? w[(size_t)c * 27 + (i % 27)] : 0.f;
  }
  __syncthreads();
  // ---- Toeplitz weight fragments of this wave's 4 channels: A[m = x_out][k = x_in] = w[kx = k - m], k = 8*gq + j
  u32x4 A[CPW][3][3];
  const int s0 = n - 8 * gq;        // element j of this lane's window holds w[kx = j - s0]
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci) {
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* wp = wl + (wave * CPW + ci) * 27 + kt * 9 + ky * 3;
        const uint32_t b0 = f32_to_bf16(wp[0]), b1 = f32_to_bf16(wp[1]), b2 = f32_to_bf16(wp[2]);
        uint32_t h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kx = j - s0;
          h[j] = kx == 0 ? b0 : kx == 1 ? b1 : kx == 2 ? b2 : 0u;
        }
        A[ci][kt][ky] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
      }
  }
  __syncthreads();
  if (tid < NCH) *reinterpret_cast<uint4*>(planes + (size_t)tid * PLANE_B + PLANE_E * 2) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int tiles_x = (g.W + TS - 1) / TS, tiles_y = (g.H + TS - 1) / TS;
  const int nitems = g.B * tiles_y * tiles_x;
  for (int item = walker; item < nitems; item += walkers) {
    const int b = item / (tiles_y * tiles_x), tl = item - b * tiles_y * tiles_x;
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    TZCK(9)
    // ---- stage: group-item = (t, plane row py, group of 8 plane columns, channel octet); the loads of TWO passes
    //      (16 x 16 B per thread) are issued before the first conversion (one wave per SIMD: nothing else hides them)
    constexpr int NGI = TT * PH * 5 * NV;
    constexpr int NPASS = (NGI + NTHR - 1) / NTHR;   // 4
    if (!(dbg & 1)) {
#pragma unroll
      for (int p0 = 0; p0 < (NPASS + 1) / 2 * 2; p0 += 2) {
        uint4 raw[2][8];
        int st_v[2], st_xg[2], st_py[2], st_t[2];
        bool st_row[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gi = (p0 + u) * NTHR + tid;
          const int v = gi % NV;
          int q = gi / NV;
          const int xg = q % 5; q /= 5;
          const int py = q % PH, t = q / PH;
          const int gy = y0 - 1 + py;
          const bool row_ok = gi < NGI && t < g.T && gy >= 0 && gy < g.H && c0 + 8 * v < g.Cp;
          st_v[u] = v; st_xg[u] = xg; st_py[u] = py; st_t[u] = t; st_row[u] = row_ok;
          const bf16_t* rowp = x + ((((size_t)b * g.T + (row_ok ? t : 0)) * g.H + (row_ok ? gy : 0)) * g.W) * g.Cp + c0 + 8 * v;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int px = 8 * xg + j, gx = x0 - 1 + px;
            const bool ok = row_ok && px < PH && gx >= 0 && gx < g.W;
            // clamped address + select: no divergent branch around the load
            const uint4 val = *reinterpret_cast<const uint4*>(rowp + (size_t)(ok ? gx : 0) * g.Cp);
            raw[u][j] = ok ? val : make_uint4(0, 0, 0, 0);
          }
        }
        TZCK(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TZCK(1)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gi = (p0 + u) * NTHR + tid;
          if (gi >= NGI) continue;
          const int v = st_v[u], xg = st_xg[u], py = st_py[u], t = st_t[u];
          float sc[8], sh[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc[e] = lss[8 * v + e]; sh[e] = lss[NCH + 8 * v + e]; }
          uint32_t P[8][4];   // [channel e][pixel pair]
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            const uint4 r0 = raw[u][2 * jp], r1 = raw[u][2 * jp + 1];
            const int px0 = 8 * xg + 2 * jp, gx0 = x0 - 1 + px0;
            const float k0 = (st_row[u] && px0 < PH && gx0 >= 0 && gx0 < g.W) ? 1.f : 0.f;
            const float k1 = (st_row[u] && px0 + 1 < PH && gx0 + 1 >= 0 && gx0 + 1 < g.W) ? 1.f : 0.f;
            const uint32_t a0[4] = {r0.x, r0.y, r0.z, r0.w}, a1[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t d0 = a0[e >> 1], d1 = a1[e >> 1];
              const float f0 = __uint_as_float((e & 1) ? (d0 & 0xffff0000u) : (d0 << 16));
              const float f1 = __uint_as_float((e & 1) ? (d1 & 0xffff0000u) : (d1 << 16));
              // padding is applied to relu(bn(a)): positions outside the image are exact zeros
              P[e][jp] = pack_bf16x2(k0 * fmaxf(fmaf(f0, sc[e], sh[e]), 0.f), k1 * fmaxf(fmaf(f1, sc[e], sh[e]), 0.f));
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            unsigned char* dst = planes + (size_t)(8 * v + e) * PLANE_B + ((size_t)(t * PH + py) * PW + 8 * xg) * 2;
            *reinterpret_cast<uint4*>(dst) = make_uint4(P[e][0], P[e][1], P[e][2], P[e][3]);
          }
        }
        TZCK(2)
      }
    }
    __syncthreads();
    TZCK(3)
    // ---- MFMA: this wave's 4 channels, 2x2 blocks of 16x16 outputs, 3 frames
    float cs1[CPW], cs2[CPW];
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) { cs1[ci] = 0.f; cs2[ci] = 0.f; }
    if (!(dbg & 2))
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
      const int c = c0 + wave * CPW + ci;
      const unsigned char* pl = planes + (size_t)(wave * CPW + ci) * PLANE_B;
      uint32_t outp[4][TT][2];
      float s1 = 0.f, s2 = 0.f;
      // all nine B fragments of a block are requested before its first MFMA, the next block's while it computes
      // (one wave per SIMD: a read -> wait -> 3 MFMA pattern exposed the LDS latency 36 times per block)
      u32x4 Bc[9], Bn[9];
      const uint32_t lbase = (uint32_t)((n * PW + 8 * gq) * 2);
#define TZ_LOAD(DST, BLK)                                                                                        \
  _Pragma("unroll") for (int ti_ = 0; ti_ < TT; ++ti_) _Pragma("unroll") for (int ky_ = 0; ky_ < 3; ++ky_)         \
    DST[ti_ * 3 + ky_] = *reinterpret_cast<const u32x4*>(                                                        \
        pl + lbase + (uint32_t)(((ti_ * PH + ((BLK) >> 1) * 16 + ky_) * PW + ((BLK) & 1) * 16) * 2));
      TZ_LOAD(Bc, 0)
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int yb = blk >> 1, xb = blk & 1;
        if (blk < 3) { TZ_LOAD(Bn, blk + 1) }
        f32x4_t D[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) D[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TT; ++ti)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
              const int to = ti - kt + 1;
              if (to >= 0 && to < TT) D[to] = mfma_bf16(A[ci][kt][ky], Bc[ti * 3 + ky], D[to]);
            }
          }
#pragma unroll
        for (int q_ = 0; q_ < 9; ++q_) Bc[q_] = Bn[q_];
        const int gy = y0 + yb * 16 + n;
        const int gxb = x0 + xb * 16 + 4 * gq;
        const bool rowv = c < g.C && gy < g.H;
        // statistics of the stored (rounded) values, branch-free: positions outside the image count as zero
        const float m0 = rowv && gxb < g.W ? 1.f : 0.f, m1 = rowv && gxb + 1 < g.W ? 1.f : 0.f;
        const float m2 = rowv && gxb + 2 < g.W ? 1.f : 0.f, m3 = rowv && gxb + 3 < g.W ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const uint32_t p01 = pack_bf16x2(D[t][0], D[t][1]), p23 = pack_bf16x2(D[t][2], D[t][3]);
          outp[blk][t][0] = p01;
          outp[blk][t][1] = p23;
          if (t < g.T) {   // wave-uniform
            const float r0 = __uint_as_float(p01 << 16) * m0, r1 = __uint_as_float(p01 & 0xffff0000u) * m1;
            const float r2 = __uint_as_float(p23 << 16) * m2, r3 = __uint_as_float(p23 & 0xffff0000u) * m3;
            s1 += (r0 + r1) + (r2 + r3);
            s2 = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, s2))));
          }
        }
      }
#undef TZ_LOAD
      // the channel's input plane is dead now (same wave, LDS operations complete in order): results over it
      unsigned char* op = planes + (size_t)(wave * CPW + ci) * PLANE_B;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int yb = blk >> 1, xb = blk & 1;
#pragma unroll
        for (int t = 0; t < TT; ++t)
          *reinterpret_cast<uint2*>(op + ((size_t)(t * TS + yb * 16 + n) * OW + xb * 16 + 4 * gq) * 2) =
              make_uint2(outp[blk][t][0], outp[blk][t][1]);
      }
      cs1[ci] = s1; cs2[ci] = s2;
    }
    TZCK(4)
    if (nc) {   // the eight wave reductions run interleaved (a chain of six dependent cross-lane steps each)
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) { cs1[ci] += __shfl_xor(cs1[ci], o, 64); cs2[ci] += __shfl_xor(cs2[ci], o, 64); }
      }
#pragma unroll
      for (int ci = 0; ci < CPW; ++ci) {
        const int c = c0 + wave * CPW + ci;
        if (lane == 0 && c < g.C) {
          atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)cs1[ci]);
          atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)cs2[ci]);
        }
      }
    }
    TZCK(5)
    __syncthreads();
    TZCK(6)
    // ---- gather: (t, row, group of 8 columns, channel octet): 8 channels x 8 pixels -> 8 pixel vectors; the LDS
    //      reads of all passes are issued first
    constexpr int NGO = TT * TS * 4 * NV;
    constexpr int NGP = (NGO + NTHR - 1) / NTHR;
    if (!(dbg & 4)) {
      uint4 R[NGP][8];
#pragma unroll
      for (int u = 0; u < NGP; ++u) {
        const int gi = u * NTHR + tid;
        const int v = gi % NV, xg = (gi / NV) & 3, q = gi / (4 * NV);
        const int py = q % TS, t = (q / TS) % TT;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          R[u][e] = *reinterpret_cast<const uint4*>(planes + (size_t)(8 * v + e) * PLANE_B + ((size_t)(t * TS + py) * OW + 8 * xg) * 2);
      }
      TZCK(7)
#pragma unroll
      for (int u = 0; u < NGP; ++u) {
        const int gi = u * NTHR + tid;
        const int v = gi % NV, xg = (gi / NV) & 3, q = gi / (4 * NV);
        const int py = q % TS, t = q / TS;
        const int gy = y0 + py;
        if (gi < NGO && t < g.T && gy < g.H && c0 + 8 * v < g.Cp) {
          bf16_t* rowp = y + ((((size_t)b * g.T + t) * g.H + gy) * g.W) * g.Cp + c0 + 8 * v;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int gx = x0 + 8 * xg + j;
            uint32_t o[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const uint4 ra = R[u][2 * qd], rb = R[u][2 * qd + 1];
              const uint32_t da = (j >> 1) == 0 ? ra.x : (j >> 1) == 1 ? ra.y : (j >> 1) == 2 ? ra.z : ra.w;
              const uint32_t db = (j >> 1) == 0 ? rb.x : (j >> 1) == 1 ? rb.y : (j >> 1) == 2 ? rb.z : rb.w;
              o[qd] = (j & 1) ? __builtin_amdgcn_perm(db, da, 0x07060302u) : __builtin_amdgcn_perm(db, da, 0x05040100u);
            }
            if (gx < g.W) *reinterpret_cast<uint4*>(rowp + (size_t)gx * g.Cp) = make_uint4(o[0], o[1], o[2], o[3]);
          }
        }
      }
    }
    TZCK(8)
    __syncthreads();
  }
#undef TZCK
  if (clk && lane == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) atomicAdd(clk + i, ck[i]);
    atomicAdd(clk + 10, 1ull);
  }
}

}  // namespace

bool c3d_dw_toeplitz_enabled() {
  const char* e = c3d_env("C3D_DW_TZ");   // read per call (tests toggle it inside one process)
  return e && atoi(e) == 1;
}

// C3D_DW_TZ_CLK=1: device buffer of 11 counters (10 phases + waves), read back by c3d_debug_tz_clock
static unsigned long long* tz_clk_buffer() {
  static unsigned long long* buf = nullptr;
  static const bool on = c3d_env("C3D_DW_TZ_CLK") && atoi(c3d_env("C3D_DW_TZ_CLK")) == 1;
  if (on && !buf && hipMalloc(&buf, 16 * sizeof(unsigned long long)) == hipSuccess) hipMemset(buf, 0, 16 * sizeof(unsigned long long));
  return on ? buf : nullptr;
}

template <int NCH>
static int tz_launch(const void* x, const float* ss, const float* w, void* y, double* nc, const Geom& g, hipStream_t s) {
  const size_t lds = (size_t)NCH * PLANE_B + 64 + 2 * NCH * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_tz_kernel<NCH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int groups = (g.Cp + NCH - 1) / NCH;
  const int nitems = g.B * ((g.H + TS - 1) / TS) * ((g.W + TS - 1) / TS);
  int walkers = (NCH == 16 ? 256 : 512) / groups;     // one (16 channels) or two (8 channels) workgroups per CU
  if (walkers < 1) walkers = 1;
  if (walkers > nitems) walkers = nitems;
  static const int env_w = c3d_env("C3D_DW_TZ_WALKERS") ? atoi(c3d_env("C3D_DW_TZ_WALKERS")) : 0;
  if (env_w > 0) walkers = env_w < nitems ? env_w : nitems;
  dw_fwd_tz_kernel<NCH><<<dim3(walkers, groups), NTHR, lds, s>>>(reinterpret_cast<const bf16_t*>(x), ss, w,
                                                               reinterpret_cast<bf16_t*>(y), nc, g, walkers,
                                                               c3d_env("C3D_DW_TZ_DBG") ? atoi(c3d_env("C3D_DW_TZ_DBG")) : 0,
                                                               tz_clk_buffer());
  return 0;
}

int c3d_dw333_fwd_toeplitz(const void* x, const float* ss, const float* w, void* y, double* nc, int B, int T, int H, int W,
                           int C, int Cp, hipStream_t s) {
  if (T > TT || T < 1) return C3D_E_UNSUPPORTED;
  const Geom g{B, T, H, W, C, Cp};
  const char* e = c3d_env("C3D_DW_TZ_NCH");   // 16 (one workgroup per CU) or 8 (two)
  if (e && atoi(e) == 8) return tz_launch<8>(x, ss, w, y, nc, g, s);
  return tz_launch<16>(x, ss, w, y, nc, g, s);
}

// (diagnosis) copies the phase counters to `out[11]` and clears them; returns 0 when C3D_DW_TZ_CLK is not set
extern "C" __attribute__((visibility("default"))) int c3d_debug_tz_clock(unsigned long long* out) {
  unsigned long long* b = tz_clk_buffer();
  if (!b) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(out, b, 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hipMemset(b, 0, 16 * sizeof(unsigned long long));
  return 1;
}
