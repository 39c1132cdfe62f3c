// Depthwise 3x3x3 Conv3d (groups = C, padding 1, stride (1,s,s), s in {1,2}) on the VALU with
// LDS-tiled inputs; replaces conv_b of the X3D bottleneck (reference model/x3d.py:184-193), FORWARD, with the
// neighbouring BatchNorm arithmetic fused (the backward pass -- data and weight gradient in one kernel -- is
// dw_bwd_fused.hip; the separate data-gradient / weight-gradient kernels of rounds 1-2 were deleted in round 4):
//
//   fwd      : in  = relu(a*scale_a + shift_a) applied once per element while staging the tile
//              (zero padding is applied AFTER the activation, as in the reference graph);
//              out = raw conv output b; epilogue = per-(sample,channel) sum / sum-of-squares
//              (feeds BN_b statistics and the SE squeeze).
//
// All T frames of a spatial tile are resident in LDS (T = 3 for BCD, 5 for SCD).  A thread
// owns one output pixel and one 8-channel vector for all T frames.
#include "pw_common.h"  // common.h + device_cus()
#include "bn_fin.h"
#include "dw_common.h"
#include "launch_hints.h"
#ifdef C3D_TUNING
#include "dw_toeplitz.h"   // Toeplitz-MFMA forward experiment (slower; instrumented build only)
#endif
#include "../../include/change3d_hip.h"
#include <cstdlib>
#include <cstring>

thread_local int c3d_side_launch = 0;

namespace {

template <typename T> struct LdsStore;  // tile element type in LDS
template <> struct LdsStore<float> { typedef float type; };
template <> struct LdsStore<bf16_t> { typedef bf16_t type; };

// raw (unconverted) 8-element vectors: what a register prefetch holds between issue and use
template <typename T> struct RawD;
template <> struct RawD<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RawD<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};


__device__ __forceinline__ void lds_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ----------------------------------------------------------------------------------------------
// Forward.  grid = (tiles_x*tiles_y, channel chunks, B); block = TH*TW*DW_CV threads.
template <typename T, int S, int TH, int TW, int TT>
__global__ __launch_bounds__(TH * TW * DW_CV) void dw_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ w, T* __restrict__ y,
    double* __restrict__ nc, const DwGeom g, const int tiles_per_wg) {
  typedef typename LdsStore<T>::type L;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  constexpr int NTHR = TH * TW * DW_CV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                         // [27][32]
  float* red = wl + 27 * 32;                                          // [NTHR/64][DW_CV][16]
  L* tile = reinterpret_cast<L*>(red + (NTHR / 64) * DW_CV * 16);     // [T][IH][IW][32]

  const int tid = threadIdx.x;
  const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx_ = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx_ * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx_, tg = co.group % gx_;
  const int c0 = chunk * DW_CV * 8;
  const int cv = tid % DW_CV;
  const int cbase = c0 + cv * 8;
  const bool c_ok = cbase < g.Cp;

  // weights -> LDS as [tap][32 channels] (zero for channels >= C)
  for (int i = tid; i < 27 * 32; i += NTHR) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = c_ok ? ss[cbase + j] : 0.f; sh[j] = c_ok ? ss[g.Cp + cbase + j] : 0.f; }

  // a workgroup walks `tiles_per_wg` tiles (one statistics flush; TT sizes the accumulators)
  const int pix = tid / DW_CV;
  const int px = pix % TW, py = pix / TW;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int tl0 = tg * tiles_per_wg;
  const int tl1 = tl0 + tiles_per_wg < ntiles ? tl0 + tiles_per_wg : ntiles;
  for (int tl = tl0; tl < tl1; ++tl) {
  const int tx = tl % tiles_x, ty = tl / tiles_x;
  __syncthreads();   // previous tile consumed (and the weights are staged)
  // input tile (all T frames) with the BN+ReLU prologue; zero outside the image
  const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
  const int items = g.T * IH * IW * DW_CV;
  for (int i = tid; i < items; i += NTHR) {  // NTHR % DW_CV == 0 -> cv fixed per thread
    const int p = i / DW_CV;
    const int ix = p % IW;
    const int q = p / IW;
    const int iy = q % IH, t = q / IH;
    const int gy = iy0 + iy, gx = ix0 + ix;
    float f[8];
    if (c_ok && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
      Vec8<T>::load(x + ((((size_t)b * g.T + t) * g.H + gy) * g.W + gx) * g.Cp + cbase, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
    }
    Vec8<L>::store(tile + (size_t)p * 32 + cv * 8, f);
  }
  __syncthreads();

  const int oy = ty * TH + py, ox = tx * TW + px;
  const bool p_ok = c_ok && oy < g.Ho && ox < g.Wo;

  float acc[TT][8];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;

#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float wk[3][8];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const float4 w0 = *reinterpret_cast<const float4*>(wl + (kt * 9 + ky * 3 + kx) * 32 + cv * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + (kt * 9 + ky * 3 + kx) * 32 + cv * 8 + 4);
        wk[kt][0] = w0.x; wk[kt][1] = w0.y; wk[kt][2] = w0.z; wk[kt][3] = w0.w;
        wk[kt][4] = w1.x; wk[kt][5] = w1.y; wk[kt][6] = w1.z; wk[kt][7] = w1.w;
      }
#pragma unroll
      for (int ti = 0; ti < TT; ++ti) {
        if (ti < g.T) {
          float v[8];
          Vec8<L>::load(tile + ((size_t)(ti * IH + py * S + ky) * IW + px * S + kx) * 32 + cv * 8, v);
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT && to < g.T) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[to][j] = fmaf(v[j], wk[kt][j], acc[to][j]);
            }
          }
        }
      }
    }
  }

  if (p_ok) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < g.T) {
        Vec8<T>::store(y + ((((size_t)b * g.T + t) * g.Ho + oy) * g.Wo + ox) * g.Cp + cbase, acc[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float r = round_as<T>(acc[t][j]);
          s1[j] += r; s2[j] += r * r;
        }
      }
    }
  }
  }  // tile walk
  if (nc == nullptr) return;
  // reduce over the pixels of the workgroup: lanes with equal cv inside a wave, then across waves
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = DW_CV; o < 64; o <<= 1) {
      s1[j] += __shfl_xor(s1[j], o, 64);
      s2[j] += __shfl_xor(s2[j], o, 64);
    }
  }
  if (lane < DW_CV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(wave * DW_CV + lane) * 16 + j] = s1[j];
      red[(wave * DW_CV + lane) * 16 + 8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (tid < DW_CV * 16) {
    const int v = tid / 16, k = tid & 15;
    float s = 0.f;
    for (int wv = 0; wv < NTHR / 64; ++wv) s += red[(wv * DW_CV + v) * 16 + k];
    const int c = c0 + v * 8 + (k & 7);
    if (c < g.C) atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + (k >> 3), (double)s);
  }
}

// ----------------------------------------------------------------------------------------------
// Forward v2 (stride 1).  Mapping chosen from the round-1 profile (v1 was LDS-read bound at
// 3.7 FMA per LDS read):
//   * wave  = one 8-channel vector; the LDS tile is stored as per-vector planes [cv][t][y][x][8] so a
//     wave's lanes read consecutive 16-byte pixels (conflict-free) and weights are wave-uniform
//     (LDS broadcast reads);
//   * lane  = a 2-pixel strip along x; the 4 input pixels of a kernel row stay in registers for all
//     kx / kt taps and all T frames (about 11 FMA per LDS read);
//   * workgroup = 8x16 output pixels x 32 channels, walks `tiles_per_wg` tiles of one sample keeping
//     the per-(sample,channel) statistics in registers, with the next tile's raw rows prefetched.
// PYR = rows per lane: 2 (8 x 16 tiles) for three frames; 1 (4 x 16 tiles) for five frames, whose 8 x 16 tile is 115 KB of
// LDS = one workgroup per CU (1.3 TB/s); the 4 x 16 tile is 69 KB = two.
constexpr int V2_TW = 16, V2_IW = V2_TW + 2;

__device__ __forceinline__ void lds_ld8v2(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// LDS plane geometry of the v2 kernels: one plane per (channel vector, half vector) holding a float4
// per pixel, [t][iy][ix]; the plane stride is padded by one float4 so the 8 planes of a pixel fall on
// distinct bank groups (conflict-free staging writes), and lanes walk x so stencil reads are dense.
template <int TT> struct V2Geo {
  static constexpr int PYR = TT <= 3 ? 2 : 1;
  static constexpr int TH = 4 * PYR, IH = TH + 2;
  static constexpr int PLANE = TT * IH * V2_IW + 1;             // float4 units
  static constexpr int NI = TT * IH * V2_IW * DW_CV;            // staged 8-channel vectors per tile
  static constexpr int SL = (NI + 255) / 256;
};

// PK: every prefetch slot's tile-independent part -- (frame, row, column) of its item in the staged tile and the element
// offset from the tile origin -- is decoded once into one packed register per slot; the per-tile request is then two adds,
// two unsigned compares and one 64-bit address instead of two divisions by constants, a frame multiply and three 64-bit
// multiply-adds per slot (9 slots: ~200 of a tile's ~1 060 VALU instructions in a kernel whose VALU is its busiest unit).
// The launcher takes it when the offset fits 22 bits and the tensor 2^31 elements.
// HV (round 6, three frames): lane = (column, FOUR channels, four output rows) instead of (column, eight channels, two rows) --
// the two halves of a wave read the two half-vector planes of its channel vector.  The tap walk then runs column tap -> frame ->
// INPUT row: every staged value is read once per column tap and feeds all the output rows / frames it reaches (6 input rows
// for 4 output rows), and the 9 weights of a column tap stay in registers: 81 ds_read_b128 per lane and tile for the same
// 1 008 FMAs that took 162 (the kernel was co-limited by LDS reads and VALU issue at two waves per SIMD; the round-4 attempt
// at more waves -- one row per lane -- doubled the reads per FMA and lost).
template <typename T, int TT, bool PK, bool HV = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HV ? 2 : 1, 2))) void dw_fwd_v2_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                        const float* __restrict__ w, T* __restrict__ y,
                                                        double* __restrict__ nc, const DwGeom g,
                                                        const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef RawD<T> RW;
  typedef V2Geo<TT> G;
  constexpr int NI = G::NI, SL = G::SL, PLANE = G::PLANE, PYR = G::PYR, V2_TH = G::TH, V2_IH = G::IH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                  // [27][32]
  float* fss = wl + 27 * 32;                                   // [2][32] scale | shift of this chunk (fin.sums mode)
  float4* tile = reinterpret_cast<float4*>(fss + 64);          // [4 cv][2 halves][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wcv = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave's channel vector
  // lane = column x, rows PYR*yp .. PYR*yp + PYR-1.  The columns of ODD row groups are rotated: a ds_read_b128 is served in
  // four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- i.e. half a row group plus
  // half of the next one, PYR * V2_IW float4 further on: with the plain x = lane & 15 those two halves overlap in 4 of
  // 16 bank quads (2 * 18 = 4 mod 16) and every stencil read takes 8 LDS cycles instead of 4
  // (/opt/skills/guides/MI355X_MICROARCH.md, LDS table; tools/lds_bank_model.py).
  const int yp = lane >> 4;
  // (both storage types since round 5: the rotation re-associates the per-sample statistics' wave sums, which used to flip a
  // ReLU unit on the hand-picked kink-free seeds of the strict f32 tests -- those tests now name and grant such a unit,
  // oracle/kinks.py)
  const int lx = (lane + (yp & 1) * ((16 - (PYR * V2_IW) % 16) & 15)) & 15;
  const int tiles_x = (g.W + V2_TW - 1) / V2_TW, tiles_y = (g.H + V2_TH - 1) / V2_TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx, tg = co.group % gx;
  const int c0 = chunk * DW_CV * 8;
  // staging role: item i = tid + 256*slot -> (cv = i & 3, pixel = i >> 2)
  const int scv = tid & 3;
  const int sbase = c0 + scv * 8;
  const bool s_ok = sbase < g.Cp;
  const int cbase = c0 + wcv * 8;       // compute role
  const bool c_ok = cbase < g.Cp;

  for (int i = tid; i < 27 * 32; i += 256) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  typename RW::type raw[SL];
  unsigned vmask = 0;
  unsigned dsc[PK ? SL : 1];   // rel << 10 | valid << 9 | ix << 4 | iy
  if constexpr (PK) {
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i_ = tid + sl * 256;
      const int p_ = i_ >> 2;
      const int ix_ = p_ % V2_IW, q_ = p_ / V2_IW;
      const int iy_ = q_ % V2_IH, t_ = q_ / V2_IH;
      const bool ok_ = i_ < NI && s_ok && t_ < g.T;
      const unsigned rel_ = (unsigned)(((t_ * g.H + iy_) * g.W + ix_) * g.Cp + sbase);
      dsc[sl] = ok_ ? (rel_ << 10) | 512u | ((unsigned)ix_ << 4) | (unsigned)iy_ : 0u;
    }
  }
#define V2_ISSUE(TL)                                                                            \
  if constexpr (PK) {                                                                           \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    const int by_ = ty_ * V2_TH - 1, bx_ = tx_ * V2_TW - 1;                                     \
    const T* xt_ = x + (ptrdiff_t)(((b * g.T * g.H + by_) * g.W + bx_) * g.Cp);   /* wave-uniform */ \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const unsigned d_ = dsc[sl];                                                              \
      const unsigned gy_ = (unsigned)(by_ + (int)(d_ & 15u)), gx_ = (unsigned)(bx_ + (int)((d_ >> 4) & 31u)); \
      if ((d_ & 512u) && gy_ < (unsigned)g.H && gx_ < (unsigned)g.W) {                          \
        raw[sl] = RW::load(xt_ + (d_ >> 10));                                                   \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  } else {                                                                                      \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const int i_ = tid + sl * 256;                                                            \
      const int p_ = i_ >> 2;                                                                   \
      const int ix_ = p_ % V2_IW, q_ = p_ / V2_IW;                                              \
      const int iy_ = q_ % V2_IH, t_ = q_ / V2_IH;                                              \
      const int gy_ = ty_ * V2_TH - 1 + iy_, gx_ = tx_ * V2_TW - 1 + ix_;                       \
      if (i_ < NI && s_ok && t_ < g.T && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) {     \
        raw[sl] = RW::load(x + ((((size_t)b * g.T + t_) * g.H + gy_) * g.W + gx_) * g.Cp + sbase); \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) V2_ISSUE(tl0)
  // BatchNorm scale / shift of this chunk: given, or rebuilt from the producer's completed sums while the first
  // tile's loads are in flight (csrc/bn_fin.h; the workgroup of (sample 0, walk 0) owns the chunk's global outputs)
  float sc[8], sh[8];
  if (fin.sums) {
    if (tid == 0 && co.chunk == 0 && co.group == 0 && fin.training && fin.nbt) *fin.nbt += 1;
    c3dfin::bn_consume(fin, g.C, g.Cp, c0, 32, co.group == 0, fss, fss + 32, tid, 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? fss[scv * 8 + j] : 0.f; sh[j] = s_ok ? fss[32 + scv * 8 + j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? ss[sbase + j] : 0.f; sh[j] = s_ok ? ss[g.Cp + sbase + j] : 0.f; }
  }
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i = tid + sl * 256;
      if (i < NI) {
        float f[8];
        if ((vmask >> sl) & 1u) {
          RW::cvt(raw[sl], f);
#pragma unroll
          for (int j = 0; j < 8; j += 2) {   // v_pk_fma_f32: the same fused multiply-add, two channels per instruction
            const f32x2_t r = __builtin_elementwise_fma(f32x2_t{f[j], f[j + 1]}, f32x2_t{sc[j], sc[j + 1]}, f32x2_t{sh[j], sh[j + 1]});
            f[j] = fmaxf(r[0], 0.f); f[j + 1] = fmaxf(r[1], 0.f);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        const int p = i >> 2;  // = (t*IH + iy)*IW + ix
        tile[(scv * 2 + 0) * PLANE + p] = make_float4(f[0], f[1], f[2], f[3]);
        tile[(scv * 2 + 1) * PLANE + p] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
    if (tl + 1 < tl1) V2_ISSUE(tl + 1)
    __syncthreads();

    if constexpr (HV) {
      static_assert(TT == 3 && PYR == 2, "half-vector lanes: three frames, 8 x 16 tiles");
      const int hv_half = lane >> 5, hv_yp = (lane >> 4) & 1;
      // columns of the second row group rotated by 8: the four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...) mix
      // lanes of both row groups, 4 rows x 18 float4 = 128 B (mod 256 B) apart -- unrotated, x = 4..11 of the one falls on the
      // bank quads of x = 12..15, 0..3 of the other
      const int hv_x = (lane + 8 * hv_yp) & 15;
      float acc4[TT][4][4];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc4[t][r][j] = 0.f;
      const float4* pl = tile + (wcv * 2 + hv_half) * PLANE + (4 * hv_yp) * V2_IW + hv_x;
      const float4* wl4 = reinterpret_cast<const float4*>(wl) + wcv * 2 + hv_half;   // tap k at wl4[8 k]
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        float4 wk[3][3];   // [kt][ky] of this column tap
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) wk[kt][ky] = wl4[(kt * 9 + ky * 3 + kx) * 8];
#pragma unroll
        for (int ti = 0; ti < TT; ++ti) {
#pragma unroll
          for (int iy = 0; iy < 6; ++iy) {
            const float4 v = pl[(ti * V2_IH + iy) * V2_IW + kx];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const int oy = iy - ky;
              if (oy >= 0 && oy < 4) {
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                  const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
                  if (to >= 0 && to < TT) {
                    acc4[to][oy][0] = fmaf(v.x, wk[kt][ky].x, acc4[to][oy][0]);
                    acc4[to][oy][1] = fmaf(v.y, wk[kt][ky].y, acc4[to][oy][1]);
                    acc4[to][oy][2] = fmaf(v.z, wk[kt][ky].z, acc4[to][oy][2]);
                    acc4[to][oy][3] = fmaf(v.w, wk[kt][ky].w, acc4[to][oy][3]);
                  }
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // one frame's six reads in flight, not all eighteen (registers)
        }
      }
      const int ox = tx * V2_TW + hv_x;
      T* const dst0 = y + (((size_t)b * g.T * g.H + (ty * V2_TH + 4 * hv_yp)) * g.W + ox) * g.Cp + cbase + hv_half * 4;
      const int64_t row_st = (int64_t)g.W * g.Cp, frm_st = (int64_t)g.H * g.W * g.Cp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oy = ty * V2_TH + 4 * hv_yp + r;
        if (c_ok && oy < g.H && ox < g.W) {
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            if (t < g.T) {
              T* dst = dst0 + (t * frm_st + r * row_st);
              if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(acc4[t][r][0], acc4[t][r][1]), pack_bf16x2(acc4[t][r][2], acc4[t][r][3]));
              } else {
                *reinterpret_cast<float4*>(dst) = make_float4(acc4[t][r][0], acc4[t][r][1], acc4[t][r][2], acc4[t][r][3]);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float rr = round_as<T>(acc4[t][r][j]);
                s1[j] += rr; s2[j] = fmaf(rr, rr, s2[j]);
              }
            }
          }
        }
      }
    } else {
    float acc[TT][PYR][8];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int py = 0; py < PYR; ++py)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][py][j] = 0.f;
    const float4* pl0 = tile + (wcv * 2 + 0) * PLANE;
    const float4* pl1 = tile + (wcv * 2 + 1) * PLANE;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        float wk[3][8];  // wave-uniform: broadcast LDS reads
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) lds_ld8v2(wl + (kt * 9 + ky * 3 + kx) * 32 + wcv * 8, wk[kt]);
#pragma unroll
        for (int ti = 0; ti < TT; ++ti) {
          float in[PYR][8];
#pragma unroll
          for (int py = 0; py < PYR; ++py) {
            const int p = (ti * V2_IH + PYR * yp + py + ky) * V2_IW + lx + kx;
            const float4 h0 = pl0[p], h1 = pl1[p];
            in[py][0] = h0.x; in[py][1] = h0.y; in[py][2] = h0.z; in[py][3] = h0.w;
            in[py][4] = h1.x; in[py][5] = h1.y; in[py][6] = h1.z; in[py][7] = h1.w;
          }
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT) {
#pragma unroll
              for (int py = 0; py < PYR; ++py)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[to][py][j] = fmaf(in[py][j], wk[kt][j], acc[to][py][j]);
            }
          }
        }
      }
    }
    const int ox = tx * V2_TW + lx;
    // one 64-bit lane address per tile (row 0 of the lane, frame 0); rows and frames are wave-uniform strides
    T* const dst0 = y + (((size_t)b * g.T * g.H + (ty * V2_TH + PYR * yp)) * g.W + ox) * g.Cp + cbase;
    const int64_t row_st = (int64_t)g.W * g.Cp, frm_st = (int64_t)g.H * g.W * g.Cp;
#pragma unroll
    for (int py = 0; py < PYR; ++py) {
      const int oy = ty * V2_TH + PYR * yp + py;
      if (c_ok && oy < g.H && ox < g.W) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < g.T) {
            T* dst = dst0 + (t * frm_st + py * row_st);
            Vec8<T>::store(dst, acc[t][py]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float r = round_as<T>(acc[t][py][j]);
              s1[j] += r; s2[j] = fmaf(r, r, s2[j]);
            }
          }
        }
      }
    }
      }
  }
#undef V2_ISSUE
  if (nc == nullptr) return;
  if constexpr (HV) {   // a lane holds four channels: the 32 lanes of each half of the wave are summed
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r1 = s1[j], r2 = s2[j];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); }
      const int c = cbase + (lane >> 5) * 4 + j;
      if ((lane & 31) == 0 && c < g.C) {
        atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)r1);
        atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)r2);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r1 = wave_sum(s1[j]), r2 = wave_sum(s2[j]);
    const int c = cbase + j;
    if (lane == 0 && c < g.C) {
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)r1);
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)r2);
    }
  }
}

template <typename T, int TT>
int launch_fwd_v2(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                  hipStream_t stream, const c3d_bn_fin* fin = nullptr) {
  const size_t lds = (27 * 32 + 64) * sizeof(float) + (size_t)DW_CV * 2 * V2Geo<TT>::PLANE * sizeof(float4);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2_kernel<T, TT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  constexpr int V2_TH = V2Geo<TT>::TH;
  const int ntiles = ((g.W + V2_TW - 1) / V2_TW) * ((g.H + V2_TH - 1) / V2_TH);
  int tpw = 16 * 2 / V2Geo<TT>::PYR;  // swept on MI355X: 1:427us 4:255 8:240 16:233 32:250 (stage-1 shape)
  // ...but a walk is a serial chain (~5.5 us per tile): keep ~2 workgroups per CU in the grid
  // (the 64x64 / 32x32 stages launched 256 / 224 workgroups of 16 / 8 tiles: one per CU, 98 / 52 us)
  const int chunks_ = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  // measured best: 16 / 8 / 4 tiles for the 128x128 / 64x64 / 32x32 stages = ~2 workgroups per CU, and
  // never fewer than 4 tiles (the prefetch pipeline needs a walk)
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks_ * g.B < 2L * device_cus()) tpw >>= 1;
  if (const char* e = c3d_env("C3D_DW_TPW")) tpw = atoi(e) > 0 ? atoi(e) : tpw;  // tuning knob
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), (long)((ntiles + tpw - 1) / tpw) * g.B));
  c3d_bn_fin f0;
  std::memset(&f0, 0, sizeof(f0));
  // packed slot descriptors (bf16): the largest tile-relative element offset in 22 bits, the tensor in 2^31 elements
  const size_t rel_max = (((size_t)(g.T - 1) * g.H + V2Geo<TT>::IH) * g.W + V2_IW) * g.Cp + g.Cp;
  const bool pk = sizeof(T) == 2 && rel_max < ((size_t)1 << 22) && (size_t)g.B * g.T * g.H * g.W * g.Cp < ((size_t)1 << 31);
  // half-vector lanes (C3D_OPT_DW_FWD_HV: bit 0 bf16 storage, bit 1 f32 storage), three frames
  constexpr bool HVT = TT == 3;
  const bool hv = HVT && (c3d_option_dw_fwd_hv & (sizeof(T) == 2 ? 1 : 2)) != 0;
  if (hv) {
    static bool attr_hv = false;
    if (!attr_hv) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2_kernel<T, TT, false, HVT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2_kernel<T, TT, sizeof(T) == 2, HVT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_hv = true;
    }
    if (pk)
      dw_fwd_v2_kernel<T, TT, sizeof(T) == 2, HVT><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                                    reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
    else
      dw_fwd_v2_kernel<T, TT, false, HVT><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                            reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  } else if (pk) {
    static bool attr_pk = false;
    if (!attr_pk) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2_kernel<T, TT, sizeof(T) == 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_pk = true;
    }
    dw_fwd_v2_kernel<T, TT, sizeof(T) == 2><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                             reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  } else {
    dw_fwd_v2_kernel<T, TT, false><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                     reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  }
  C3D_CHECK_LAUNCH();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// Forward v2, stride 2 (the first block of every stage).  Same roles as v2 -- wave = one 8-channel vector with
// wave-uniform weights, lane = an output pixel, walking workgroup with the next tile's raw rows prefetched -- over a
// POLYPHASE tile: the 9 x 33 input pixels a 4 x 16 output tile reads are staged as four parity planes
// [t][iy & 1][ix & 1][iy >> 1][ix >> 1], so tap (ky, kx) of output (oy, ox) is plane (ky & 1, kx & 1) at
// (oy + (ky == 2), ox + (kx == 2)): the lanes of a wave (consecutive ox) read consecutive 16-byte pixels, as in the
// stride-1 kernel (reading every second pixel of a dense tile is a two-way bank conflict on every ds_read_b128).
// The v1 kernel this replaces (thread = pixel x channel vector, 4 x 8 tiles, no prefetch) ran the three stride-2 blocks
// of the BCD step at 0.42 TB/s.
// Empty asm that "uses" the 24 accumulators: keeps the scheduler from hoisting the LDS reads of all nine (ky, kx) steps
// above the first FMA (432 live registers, 836 B of scratch per lane without it).
__device__ __forceinline__ void pin_acc3(float (&a)[3][8]) {
  asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]),
                    "+v"(a[0][7]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]),
                    "+v"(a[1][6]), "+v"(a[1][7]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[2][4]),
                    "+v"(a[2][5]), "+v"(a[2][6]), "+v"(a[2][7]));
}
template <int TT> __device__ __forceinline__ void pin_acc_tt(float (&a)[TT][8]) {
  if constexpr (TT == 3) pin_acc3(a);
}
constexpr int S2_TH = 4, S2_TW = 16, S2_IH = 2 * S2_TH + 1, S2_IW = 2 * S2_TW + 1, S2_HY = S2_TH + 1, S2_HX = S2_TW + 1;
template <int TT> struct V2S2Geo {
  static constexpr int PAR = S2_HY * S2_HX;                     // float4 units per (frame, parity) plane
  static constexpr int PLANE = TT * 4 * PAR + 1;                // per (channel vector, half vector)
  static constexpr int NI = TT * S2_IH * S2_IW * DW_CV;         // staged 8-channel vectors per tile
  static constexpr int SL = (NI + 255) / 256;
};

// PK: packed slot descriptors, as in dw_fwd_v2_kernel (14 slots here: their per-tile decode was more VALU instructions than
// the tile's 252 packed FMAs)
// W8 (round 6): 512 threads on the same tile -- two waves per channel vector, each on one HALF vector (4 channels) of the 64
// output pixels, the two half-vector planes of the vector being separate LDS planes anyway.  The tile is 131 KB (four parity
// planes of 9 x 33 pixels x 3 frames x 32 channels in f32): ONE workgroup per CU, i.e. with 256 threads one wave per SIMD --
// nothing ran beside a wave that waited for its rows or for LDS (2.2-2.5 TB/s on rows that stream from HBM).  Same LDS
// reads per FMA, half the staging items and half the tap walk per wave, twice the waves.
template <typename T, int TT, bool PK, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void dw_fwd_v2s2_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                          const float* __restrict__ w, T* __restrict__ y,
                                                          double* __restrict__ nc, const DwGeom g,
                                                          const int tiles_per_wg, const c3d_bn_fin fin) {
  typedef RawD<T> RW;
  typedef V2S2Geo<TT> G;
  constexpr int NTHR = W8 ? 512 : 256;
  constexpr int NI = G::NI, SL = (G::NI + NTHR - 1) / NTHR, PLANE = G::PLANE, PAR = G::PAR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);                  // [27][32]
  float* fss = wl + 27 * 32;                                   // [2][32] scale | shift of this chunk (fin.sums mode)
  float4* tile = reinterpret_cast<float4*>(fss + 64);          // [4 cv][2 halves][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wcv = wave_ & 3;                                   // this wave's channel vector
  [[maybe_unused]] const int whalf = wave_ >> 2;               // W8: its half of the vector (0 with four waves)
  // lane = output pixel (ly, lx) of the tile; the columns of odd rows are rotated by one so that the two half rows a
  // ds_read_b128 lane group joins (rows S2_HX = 17 float4 apart) fall on 16 distinct bank quads (see dw_fwd_v2_kernel)
  const int ly = lane >> 4;
  const int lx = (lane + (ly & 1) * ((16 - S2_HX % 16) & 15)) & 15;
  const int tiles_x = (g.Wo + S2_TW - 1) / S2_TW, tiles_y = (g.Ho + S2_TH - 1) / S2_TH;
  const int ntiles = tiles_x * tiles_y;
  const int gx = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  const ChunkOrder co = chunk_order((g.Cp + DW_CV * 8 - 1) / (DW_CV * 8), gx * g.B);
  if (co.group < 0) return;
  const int chunk = co.chunk, b = co.group / gx, tg = co.group % gx;
  const int c0 = chunk * DW_CV * 8;
  const int scv = tid & 3;              // staging role: item i = tid + 256*slot -> (cv = i & 3, pixel = i >> 2)
  const int sbase = c0 + scv * 8;
  const bool s_ok = sbase < g.Cp;
  const int cbase = c0 + wcv * 8;       // compute role
  const bool c_ok = cbase < g.Cp;

  for (int i = tid; i < 27 * 32; i += NTHR) {
    const int tap = i / 32, c = c0 + (i & 31);
    wl[i] = (c < g.C) ? w[(size_t)c * 27 + tap] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  // per-slot constants of the staging role: offset inside the input tile (global) and inside the parity planes (LDS)
  typename RW::type raw[SL];
  unsigned vmask = 0;
  unsigned dsc[PK ? SL : 1];   // (rel / 8) << 11 | valid << 10 | ix << 4 | iy   (rel is a multiple of 8 elements)
  if constexpr (PK) {
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i_ = tid + sl * NTHR;
      const int p_ = i_ >> 2;
      const int ix_ = p_ % S2_IW, q_ = p_ / S2_IW;
      const int iy_ = q_ % S2_IH, t_ = q_ / S2_IH;
      const bool ok_ = i_ < NI && s_ok && t_ < g.T;
      const unsigned rel_ = (unsigned)(((t_ * g.H + iy_) * g.W + ix_) * g.Cp + sbase);
      dsc[sl] = ok_ ? ((rel_ >> 3) << 11) | 1024u | ((unsigned)ix_ << 4) | (unsigned)iy_ : 0u;
    }
  }
#define S2_ISSUE(TL)                                                                            \
  if constexpr (PK) {                                                                           \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    const int by_ = ty_ * (2 * S2_TH) - 1, bx_ = tx_ * (2 * S2_TW) - 1;                         \
    const T* xt_ = x + (ptrdiff_t)(((b * g.T * g.H + by_) * g.W + bx_) * g.Cp);   /* wave-uniform */ \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const unsigned d_ = dsc[sl];                                                              \
      const unsigned gy_ = (unsigned)(by_ + (int)(d_ & 15u)), gx_ = (unsigned)(bx_ + (int)((d_ >> 4) & 63u)); \
      if ((d_ & 1024u) && gy_ < (unsigned)g.H && gx_ < (unsigned)g.W) {                         \
        raw[sl] = RW::load(xt_ + ((d_ >> 11) << 3));                                            \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  } else {                                                                                      \
    const int tx_ = (TL) % tiles_x, ty_ = (TL) / tiles_x;                                       \
    vmask = 0;                                                                                  \
    _Pragma("unroll") for (int sl = 0; sl < SL; ++sl) {                                         \
      const int i_ = tid + sl * NTHR;                                                           \
      const int p_ = i_ >> 2;                                                                   \
      const int ix_ = p_ % S2_IW, q_ = p_ / S2_IW;                                              \
      const int iy_ = q_ % S2_IH, t_ = q_ / S2_IH;                                              \
      const int gy_ = ty_ * (2 * S2_TH) - 1 + iy_, gx_ = tx_ * (2 * S2_TW) - 1 + ix_;           \
      if (i_ < NI && s_ok && t_ < g.T && gy_ >= 0 && gy_ < g.H && gx_ >= 0 && gx_ < g.W) {     \
        raw[sl] = RW::load(x + ((((size_t)b * g.T + t_) * g.H + gy_) * g.W + gx_) * g.Cp + sbase); \
        vmask |= 1u << sl;                                                                      \
      }                                                                                         \
    }                                                                                           \
  }

  const int tl0 = tg * tiles_per_wg;
  int tl1 = tl0 + tiles_per_wg;
  if (tl1 > ntiles) tl1 = ntiles;
  if (tl0 < tl1) S2_ISSUE(tl0)
  float sc[8], sh[8];
  if (fin.sums) {   // BatchNorm_a scale / shift rebuilt from conv_a's completed sums (csrc/bn_fin.h), as in the stride-1 kernel
    if (tid == 0 && co.chunk == 0 && co.group == 0 && fin.training && fin.nbt) *fin.nbt += 1;
    c3dfin::bn_consume(fin, g.C, g.Cp, c0, 32, co.group == 0, fss, fss + 32, tid, NTHR);
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? fss[scv * 8 + j] : 0.f; sh[j] = s_ok ? fss[32 + scv * 8 + j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_ok ? ss[sbase + j] : 0.f; sh[j] = s_ok ? ss[g.Cp + sbase + j] : 0.f; }
  }
  for (int tl = tl0; tl < tl1; ++tl) {
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int i = tid + sl * NTHR;
      if (i < NI) {
        float f[8];
        if ((vmask >> sl) & 1u) {
          RW::cvt(raw[sl], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = 0.f;
        }
        const int p = i >> 2;
        const int ix = p % S2_IW, q = p / S2_IW;
        const int iy = q % S2_IH, t = q / S2_IH;
        const int d = ((t * 2 + (iy & 1)) * 2 + (ix & 1)) * PAR + (iy >> 1) * S2_HX + (ix >> 1);
        tile[(scv * 2 + 0) * PLANE + d] = make_float4(f[0], f[1], f[2], f[3]);
        tile[(scv * 2 + 1) * PLANE + d] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
    if (tl + 1 < tl1) S2_ISSUE(tl + 1)
    __syncthreads();

    if constexpr (W8) {
      float acc4[TT][4];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[t][j] = 0.f;
      const float4* pl = tile + (wcv * 2 + whalf) * PLANE + ly * S2_HX + lx;
      const float4* wl4 = reinterpret_cast<const float4*>(wl) + wcv * 2 + whalf;   // tap k at wl4[8 k]
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float4 wk[3];
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) wk[kt] = wl4[(kt * 9 + ky * 3 + kx) * 8];
          const int off = ((ky & 1) * 2 + (kx & 1)) * PAR + (ky >> 1) * S2_HX + (kx >> 1);   // compile-time immediate
#pragma unroll
          for (int ti = 0; ti < TT; ++ti) {
            const float4 v = pl[ti * 4 * PAR + off];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
              const int to = ti - kt + 1;
              if (to >= 0 && to < TT) {
                acc4[to][0] = fmaf(v.x, wk[kt].x, acc4[to][0]); acc4[to][1] = fmaf(v.y, wk[kt].y, acc4[to][1]);
                acc4[to][2] = fmaf(v.z, wk[kt].z, acc4[to][2]); acc4[to][3] = fmaf(v.w, wk[kt].w, acc4[to][3]);
              }
            }
          }
          if constexpr (TT == 3)   // (keeps the LDS reads of all nine steps from being hoisted above the first FMA: see pin_acc3)
            asm volatile("" : "+v"(acc4[0][0]), "+v"(acc4[0][1]), "+v"(acc4[0][2]), "+v"(acc4[0][3]), "+v"(acc4[1][0]), "+v"(acc4[1][1]),
                              "+v"(acc4[1][2]), "+v"(acc4[1][3]), "+v"(acc4[2][0]), "+v"(acc4[2][1]), "+v"(acc4[2][2]), "+v"(acc4[2][3]));
        }
      }
      const int ox = tx * S2_TW + lx, oy = ty * S2_TH + ly;
      if (c_ok && oy < g.Ho && ox < g.Wo) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          if (t < g.T) {
            T* dst = y + ((((size_t)b * g.T + t) * g.Ho + oy) * g.Wo + ox) * g.Cp + cbase + whalf * 4;
            if constexpr (sizeof(T) == 2) {
              *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(acc4[t][0], acc4[t][1]), pack_bf16x2(acc4[t][2], acc4[t][3]));
            } else {
              *reinterpret_cast<float4*>(dst) = make_float4(acc4[t][0], acc4[t][1], acc4[t][2], acc4[t][3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float r = round_as<T>(acc4[t][j]);
              s1[j] += r; s2[j] = fmaf(r, r, s2[j]);
            }
          }
        }
      }
    } else {
    float acc[TT][8];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const float4* pl0 = tile + (wcv * 2 + 0) * PLANE + ly * S2_HX + lx;
    const float4* pl1 = tile + (wcv * 2 + 1) * PLANE + ly * S2_HX + lx;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float wk[3][8];  // wave-uniform: broadcast LDS reads
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) lds_ld8v2(wl + (kt * 9 + ky * 3 + kx) * 32 + wcv * 8, wk[kt]);
        const int off = ((ky & 1) * 2 + (kx & 1)) * PAR + (ky >> 1) * S2_HX + (kx >> 1);   // compile-time immediate
#pragma unroll
        for (int ti = 0; ti < TT; ++ti) {
          const float4 h0 = pl0[ti * 4 * PAR + off], h1 = pl1[ti * 4 * PAR + off];
          const float in[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) {
            const int to = ti - kt + 1;  // out[to] += in[to + kt - 1] * w[kt]
            if (to >= 0 && to < TT) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[to][j] = fmaf(in[j], wk[kt][j], acc[to][j]);
            }
          }
        }
        pin_acc_tt<TT>(acc);
      }
    }
    const int ox = tx * S2_TW + lx, oy = ty * S2_TH + ly;
    if (c_ok && oy < g.Ho && ox < g.Wo) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (t < g.T) {
          T* dst = y + ((((size_t)b * g.T + t) * g.Ho + oy) * g.Wo + ox) * g.Cp + cbase;
          Vec8<T>::store(dst, acc[t]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float r = round_as<T>(acc[t][j]);
            s1[j] += r; s2[j] = fmaf(r, r, s2[j]);
          }
        }
      }
    }
      }
  }
#undef S2_ISSUE
  if (nc == nullptr) return;
#pragma unroll
  for (int j = 0; j < (W8 ? 4 : 8); ++j) {
    const float r1 = wave_sum(s1[j]), r2 = wave_sum(s2[j]);
    const int c = cbase + (W8 ? whalf * 4 : 0) + j;
    if (lane == 0 && c < g.C) {
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2, (double)r1);
      atomicAdd(nc + ((size_t)b * g.Cp + c) * 2 + 1, (double)r2);
    }
  }
}

template <typename T, int TT>
int launch_fwd_v2s2(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                    hipStream_t stream, const c3d_bn_fin* fin = nullptr) {
  const size_t lds = (27 * 32 + 64) * sizeof(float) + (size_t)DW_CV * 2 * V2S2Geo<TT>::PLANE * sizeof(float4);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;   // (five frames: 218 KB -- the v1 kernel)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2s2_kernel<T, TT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2s2_kernel<T, TT, sizeof(T) == 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.Wo + S2_TW - 1) / S2_TW) * ((g.Ho + S2_TH - 1) / S2_TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  int tpw = c3d_knob("C3D_DWF2_TPW", 16);   // one workgroup per CU (LDS): ~2 rounds of workgroups, >= 4 tiles for the prefetch
  while (tpw > 4 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 2L * device_cus()) tpw >>= 1;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  c3d_bn_fin f0;
  std::memset(&f0, 0, sizeof(f0));
  // packed slot descriptors (bf16): the largest tile-relative element offset / 8 in 21 bits, the tensor in 2^31 elements
  const size_t rel_max = (((size_t)(g.T - 1) * g.H + S2_IH) * g.W + S2_IW) * g.Cp + g.Cp;
  const bool pk = sizeof(T) == 2 && (rel_max >> 3) < ((size_t)1 << 21) && (size_t)g.B * g.T * g.H * g.W * g.Cp < ((size_t)1 << 31);
  // eight waves on the tile (C3D_OPT_DW_FWD_HV bit 2), bf16 storage
  constexpr bool W8T = sizeof(T) == 2;
  if (W8T && (c3d_option_dw_fwd_hv & 4)) {
    static bool attr_w8 = false;
    if (!attr_w8) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2s2_kernel<T, TT, false, W8T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_v2s2_kernel<T, TT, W8T, W8T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_w8 = true;
    }
    if (pk)
      dw_fwd_v2s2_kernel<T, TT, W8T, W8T><<<grid, dim3(512), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                           reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
    else
      dw_fwd_v2s2_kernel<T, TT, false, W8T><<<grid, dim3(512), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                             reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  } else if (pk)
    dw_fwd_v2s2_kernel<T, TT, sizeof(T) == 2><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                               reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  else
    dw_fwd_v2s2_kernel<T, TT, false><<<grid, dim3(256), lds, stream>>>(reinterpret_cast<const T*>(x), ss, w,
                                                                       reinterpret_cast<T*>(y), nc, g, tpw, fin ? *fin : f0);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int S> struct DwTile;  // forward / wgrad output tile per workgroup
template <typename T> struct DwTile<T, 1> { static constexpr int TH = 8, TW = 8; };
template <typename T> struct DwTile<T, 2> { static constexpr int TH = 4, TW = 8; };

template <typename T, int S, int TT>
int launch_fwd_t(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
                 hipStream_t stream) {
  constexpr int TH = DwTile<T, S>::TH, TW = DwTile<T, S>::TW;
  constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, NTHR = TH * TW * DW_CV;
  const size_t lds = (27 * 32 + (NTHR / 64) * DW_CV * 16) * sizeof(float) +
                     (size_t)g.T * IH * IW * 32 * sizeof(typename LdsStore<T>::type);
  if (lds > 160 * 1024) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fwd_kernel<T, S, TH, TW, TT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((g.Wo + TW - 1) / TW) * ((g.Ho + TH - 1) / TH);
  const int chunks = (g.Cp + DW_CV * 8 - 1) / (DW_CV * 8);
  static const int env_tpw = c3d_env("C3D_DWF1_TPW") ? atoi(c3d_env("C3D_DWF1_TPW")) : 0;
  int tpw = 8;   // no prefetch in this kernel: the walk only amortises the weight staging and the statistics flush
  while (tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * chunks * g.B < 2L * device_cus()) tpw >>= 1;   // swept: 8 is best
  if (env_tpw > 0) tpw = env_tpw;
  if (tpw > ntiles) tpw = ntiles;
  dim3 grid(chunk_order_grid(chunks, (long)((ntiles + tpw - 1) / tpw) * g.B));
  dw_fwd_kernel<T, S, TH, TW, TT><<<grid, dim3(NTHR), lds, stream>>>(
      reinterpret_cast<const T*>(x), ss, w, reinterpret_cast<T*>(y), nc, g, tpw);
  C3D_CHECK_LAUNCH();
  return 0;
}

template <typename T, int S>
int launch_fwd(const void* x, const float* ss, const float* w, void* y, double* nc, const DwGeom& g,
               hipStream_t stream) {
  if (g.T <= 3) return launch_fwd_t<T, S, 3>(x, ss, w, y, nc, g, stream);
  return launch_fwd_t<T, S, 5>(x, ss, w, y, nc, g, stream);
}

}  // namespace

extern "C" int c3d_dw333_fwd(const void* x, const float* ss, const float* w, void* y, double* nc_sums, int32_t B,
                             int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                             void* stream) {
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!x || !ss || !w || !y || !geom_ok(g)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef C3D_TUNING
  if (stride == 1 && dtype == C3D_DT_BF16 && T <= 3 && c3d_dw_toeplitz_enabled()) {   // matrix-core kernel (dw_toeplitz.hip)
    const int rc = c3d_dw333_fwd_toeplitz(x, ss, w, y, nc_sums, B, T, H, W, C, Cp, s);
    if (rc != C3D_E_UNSUPPORTED) { if (rc == 0) C3D_CHECK_LAUNCH(); return rc; }
  }
#endif
  if (stride == 1) {  // v2 mapping (wave = channel vector, lane = x-strip)
    int rc = C3D_E_UNSUPPORTED;
    if (dtype == C3D_DT_F32) rc = T <= 3 ? launch_fwd_v2<float, 3>(x, ss, w, y, nc_sums, g, s)
                                         : launch_fwd_v2<float, 5>(x, ss, w, y, nc_sums, g, s);
    else if (dtype == C3D_DT_BF16) rc = T <= 3 ? launch_fwd_v2<bf16_t, 3>(x, ss, w, y, nc_sums, g, s)
                                               : launch_fwd_v2<bf16_t, 5>(x, ss, w, y, nc_sums, g, s);
    else return C3D_E_BADARG;
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (stride == 2 && T <= 3 && c3d_knob("C3D_DWF2_V2", 1)) {   // polyphase v2 (three frames: the tile fits LDS)
    int rc = C3D_E_UNSUPPORTED;
    if (dtype == C3D_DT_F32) rc = launch_fwd_v2s2<float, 3>(x, ss, w, y, nc_sums, g, s);
    else if (dtype == C3D_DT_BF16) rc = launch_fwd_v2s2<bf16_t, 3>(x, ss, w, y, nc_sums, g, s);
    else return C3D_E_BADARG;
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (dtype == C3D_DT_F32) return stride == 1 ? launch_fwd<float, 1>(x, ss, w, y, nc_sums, g, s)
                                              : launch_fwd<float, 2>(x, ss, w, y, nc_sums, g, s);
  if (dtype == C3D_DT_BF16) return stride == 1 ? launch_fwd<bf16_t, 1>(x, ss, w, y, nc_sums, g, s)
                                               : launch_fwd<bf16_t, 2>(x, ss, w, y, nc_sums, g, s);
  return C3D_E_BADARG;
}

extern "C" int c3d_dw333_fwd_fin(const void* x, const c3d_bn_fin* fin, const float* w, void* y, double* nc_sums,
                                 int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride,
                                 int32_t dtype, void* stream) {
  if (!fin || !fin->sums || !fin->ss || !fin->gamma || !fin->beta || !fin->training) return C3D_E_BADARG;
  DwGeom g{B, T, H, W, (H - 1) / (stride > 0 ? stride : 1) + 1, (W - 1) / (stride > 0 ? stride : 1) + 1, C, Cp, stride};
  if (!x || !w || !y || !geom_ok(g)) return C3D_E_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef C3D_TUNING
  const bool tz = stride == 1 && dtype == C3D_DT_BF16 && T <= 3 && c3d_dw_toeplitz_enabled();
#else
  const bool tz = false;
#endif
  if (!tz && stride == 1 && (dtype == C3D_DT_F32 || dtype == C3D_DT_BF16)) {
    int rc;
    if (dtype == C3D_DT_F32) rc = T <= 3 ? launch_fwd_v2<float, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                                         : launch_fwd_v2<float, 5>(x, fin->ss, w, y, nc_sums, g, s, fin);
    else rc = T <= 3 ? launch_fwd_v2<bf16_t, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                     : launch_fwd_v2<bf16_t, 5>(x, fin->ss, w, y, nc_sums, g, s, fin);
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  if (stride == 2 && T <= 3 && (dtype == C3D_DT_F32 || dtype == C3D_DT_BF16) && c3d_knob("C3D_DWF2_V2", 1)) {
    const int rc = dtype == C3D_DT_F32 ? launch_fwd_v2s2<float, 3>(x, fin->ss, w, y, nc_sums, g, s, fin)
                                       : launch_fwd_v2s2<bf16_t, 3>(x, fin->ss, w, y, nc_sums, g, s, fin);
    if (rc != C3D_E_UNSUPPORTED) return rc;
  }
  // no folded kernel for this shape: the separate launch, then the plain kernel
  const int rc = c3d_bn_finalize(fin->sums, C3D_STAT_STRIPES, fin->count, fin->gamma, fin->beta, fin->running_mean,
                                 fin->running_var, fin->nbt, fin->momentum, fin->eps, C, Cp, fin->training, fin->ss,
                                 fin->mr, stream);
  if (rc) return rc;
  return c3d_dw333_fwd(x, fin->ss, w, y, nc_sums, B, T, H, W, C, Cp, stride, dtype, stream);
}
