// Pointwise-convolution row GEMM on MFMA (gfx950), forward and data-gradient: kernel, planner and dispatch
// templates.  Included by pw_gemm.hip (bf16 instantiations + the C entry point) and pw_gemm_f32.hip (f32 parity
// instantiations) so the two halves compile in parallel.
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
#pragma once
#include "common.h"
#include "../../include/change3d_hip.h"
#include "pw_common.h"
#include "bn_fin.h"
#include <cstdlib>

#ifdef C3D_PW_CLOCK
// Debug build only (tools/pw_phase_clock.py): per-phase shader-clock sums over all waves.
constexpr int CLK_WAVES = 8192;
static __device__ unsigned long long c3d_pw_clk[CLK_WAVES][16];   // per-wave slots (atomics on shared slots stall the launch); one per translation unit
#define CLK_DECL unsigned long long clk_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long clk_last_ = __builtin_amdgcn_s_memtime();
#define CLK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk_[i] += t_ - clk_last_; clk_last_ = t_; }
#define CLK_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define CLK_FLUSH if (lane == 0) { const int w_ = (blockIdx.x * WAVES + wave) % CLK_WAVES; for (int i_ = 0; i_ < 15; ++i_) c3d_pw_clk[w_][i_] += clk_[i_]; c3d_pw_clk[w_][15] += 1ull; }
#else
#define CLK_DECL
#define CLK(i)
#define CLK_WAITVM
#define CLK_FLUSH
#endif

int c3d_detail_pw_wgrad_reduce(const float* ws, float* dw, int N, int K, int parts, int sn, int sk, hipStream_t stream);   // pw_wgrad.hip

namespace {

// ---- hand-scheduled weight-fragment pipeline (bf16) ------------------------------------------------------
// U ds_read_b128 in flight, then the U (or 2U) MFMAs behind counted lgkmcnt waits, in ONE asm statement: the
// compiler interleaves read -> wait -> MFMA with at most two reads in flight (LDS latency ~130 clk against 16-32
// clk of MFMA per fragment), and nothing (in particular no scalar load, which shares lgkmcnt and returns out of
// order) can be scheduled into the counted region.  Operand A = weight fragment, B = data rows.  The trailing
// s_nop covers the MFMA -> VALU read hazard the compiler cannot see through inline asm.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int U, bool PAIR> struct MfmaBatch;
template <> struct MfmaBatch<1, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "=&v"(b0), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<1, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %6, %1\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a2[0]), "=&v"(b0), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<2, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %2, %4\n\tv_add_u32 %4, %5, %4\n\tds_read_b128 %3, %4\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %6, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %6, %1\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "=&v"(b0), "=&v"(b1), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<2, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %4, %6\n\tv_add_u32 %6, %7, %6\n\tds_read_b128 %5, %6\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\tv_mfma_f32_16x16x32_bf16 %2, %4, %9, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\tv_mfma_f32_16x16x32_bf16 %3, %5, %9, %3\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a2[0]), "+v"(a2[1]), "=&v"(b0), "=&v"(b1), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<3, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %3, %6\n\tv_add_u32 %6, %7, %6\n\tds_read_b128 %4, %6\n\tv_add_u32 %6, %7, %6\n\tds_read_b128 %5, %6\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %0, %3, %8, %0\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %8, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %2, %5, %8, %2\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<3, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %6, %9\n\tv_add_u32 %9, %10, %9\n\tds_read_b128 %7, %9\n\tv_add_u32 %9, %10, %9\n\tds_read_b128 %8, %9\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %0, %6, %11, %0\n\tv_mfma_f32_16x16x32_bf16 %3, %6, %12, %3\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %1, %7, %11, %1\n\tv_mfma_f32_16x16x32_bf16 %4, %7, %12, %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %2, %8, %11, %2\n\tv_mfma_f32_16x16x32_bf16 %5, %8, %12, %5\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<4, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %4, %8\n\tv_add_u32 %8, %9, %8\n\tds_read_b128 %5, %8\n\tv_add_u32 %8, %9, %8\n\tds_read_b128 %6, %8\n\tv_add_u32 %8, %9, %8\n\tds_read_b128 %7, %8\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %10, %0\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %10, %1\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %2, %6, %10, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %3, %7, %10, %3\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<4, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %8, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %9, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %10, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %11, %12\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %0, %8, %14, %0\n\tv_mfma_f32_16x16x32_bf16 %4, %8, %15, %4\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %1, %9, %14, %1\n\tv_mfma_f32_16x16x32_bf16 %5, %9, %15, %5\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %2, %10, %14, %2\n\tv_mfma_f32_16x16x32_bf16 %6, %10, %15, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %3, %11, %14, %3\n\tv_mfma_f32_16x16x32_bf16 %7, %11, %15, %7\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<5, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %5, %10\n\tv_add_u32 %10, %11, %10\n\tds_read_b128 %6, %10\n\tv_add_u32 %10, %11, %10\n\tds_read_b128 %7, %10\n\tv_add_u32 %10, %11, %10\n\tds_read_b128 %8, %10\n\tv_add_u32 %10, %11, %10\n\tds_read_b128 %9, %10\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %12, %0\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %1, %6, %12, %1\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %2, %7, %12, %2\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %3, %8, %12, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %4, %9, %12, %4\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<5, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %10, %15\n\tv_add_u32 %15, %16, %15\n\tds_read_b128 %11, %15\n\tv_add_u32 %15, %16, %15\n\tds_read_b128 %12, %15\n\tv_add_u32 %15, %16, %15\n\tds_read_b128 %13, %15\n\tv_add_u32 %15, %16, %15\n\tds_read_b128 %14, %15\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %0, %10, %17, %0\n\tv_mfma_f32_16x16x32_bf16 %5, %10, %18, %5\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %1, %11, %17, %1\n\tv_mfma_f32_16x16x32_bf16 %6, %11, %18, %6\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %2, %12, %17, %2\n\tv_mfma_f32_16x16x32_bf16 %7, %12, %18, %7\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %3, %13, %17, %3\n\tv_mfma_f32_16x16x32_bf16 %8, %13, %18, %8\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %4, %14, %17, %4\n\tv_mfma_f32_16x16x32_bf16 %9, %14, %18, %9\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<6, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4; u32x4_t b5;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %6, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %7, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %8, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %9, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %10, %12\n\tv_add_u32 %12, %13, %12\n\tds_read_b128 %11, %12\n\ts_waitcnt lgkmcnt(5)\n\tv_mfma_f32_16x16x32_bf16 %0, %6, %14, %0\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %1, %7, %14, %1\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %2, %8, %14, %2\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %3, %9, %14, %3\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %4, %10, %14, %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %5, %11, %14, %5\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<6, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4; u32x4_t b5;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %12, %18\n\tv_add_u32 %18, %19, %18\n\tds_read_b128 %13, %18\n\tv_add_u32 %18, %19, %18\n\tds_read_b128 %14, %18\n\tv_add_u32 %18, %19, %18\n\tds_read_b128 %15, %18\n\tv_add_u32 %18, %19, %18\n\tds_read_b128 %16, %18\n\tv_add_u32 %18, %19, %18\n\tds_read_b128 %17, %18\n\ts_waitcnt lgkmcnt(5)\n\tv_mfma_f32_16x16x32_bf16 %0, %12, %20, %0\n\tv_mfma_f32_16x16x32_bf16 %6, %12, %21, %6\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %1, %13, %20, %1\n\tv_mfma_f32_16x16x32_bf16 %7, %13, %21, %7\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %2, %14, %20, %2\n\tv_mfma_f32_16x16x32_bf16 %8, %14, %21, %8\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %3, %15, %20, %3\n\tv_mfma_f32_16x16x32_bf16 %9, %15, %21, %9\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %4, %16, %20, %4\n\tv_mfma_f32_16x16x32_bf16 %10, %16, %21, %10\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %5, %17, %20, %5\n\tv_mfma_f32_16x16x32_bf16 %11, %17, %21, %11\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
template <> struct MfmaBatch<7, false> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4; u32x4_t b5; u32x4_t b6;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %7, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %8, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %9, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %10, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %11, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %12, %14\n\tv_add_u32 %14, %15, %14\n\tds_read_b128 %13, %14\n\ts_waitcnt lgkmcnt(6)\n\tv_mfma_f32_16x16x32_bf16 %0, %7, %16, %0\n\ts_waitcnt lgkmcnt(5)\n\tv_mfma_f32_16x16x32_bf16 %1, %8, %16, %1\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %2, %9, %16, %2\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %3, %10, %16, %3\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %4, %11, %16, %4\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %5, %12, %16, %5\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %6, %13, %16, %6\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "+v"(addr)
                 : "s"(str), "v"(xb));
  }
};
template <> struct MfmaBatch<7, true> {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    u32x4_t b0; u32x4_t b1; u32x4_t b2; u32x4_t b3; u32x4_t b4; u32x4_t b5; u32x4_t b6;
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %14, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %15, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %16, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %17, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %18, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %19, %21\n\tv_add_u32 %21, %22, %21\n\tds_read_b128 %20, %21\n\ts_waitcnt lgkmcnt(6)\n\tv_mfma_f32_16x16x32_bf16 %0, %14, %23, %0\n\tv_mfma_f32_16x16x32_bf16 %7, %14, %24, %7\n\ts_waitcnt lgkmcnt(5)\n\tv_mfma_f32_16x16x32_bf16 %1, %15, %23, %1\n\tv_mfma_f32_16x16x32_bf16 %8, %15, %24, %8\n\ts_waitcnt lgkmcnt(4)\n\tv_mfma_f32_16x16x32_bf16 %2, %16, %23, %2\n\tv_mfma_f32_16x16x32_bf16 %9, %16, %24, %9\n\ts_waitcnt lgkmcnt(3)\n\tv_mfma_f32_16x16x32_bf16 %3, %17, %23, %3\n\tv_mfma_f32_16x16x32_bf16 %10, %17, %24, %10\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x32_bf16 %4, %18, %23, %4\n\tv_mfma_f32_16x16x32_bf16 %11, %18, %24, %11\n\ts_waitcnt lgkmcnt(1)\n\tv_mfma_f32_16x16x32_bf16 %5, %19, %23, %5\n\tv_mfma_f32_16x16x32_bf16 %12, %19, %24, %12\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_16x16x32_bf16 %6, %20, %23, %6\n\tv_mfma_f32_16x16x32_bf16 %13, %20, %24, %13\n\ts_nop 15\n\ts_nop 2"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "+v"(addr)
                 : "s"(str), "v"(xb), "v"(xb2));
  }
};
// all NT output tiles of one K step, UB fragments per batch
template <int NT, int UB, int N0, bool PAIR> struct MfmaSeq {
  static __device__ __forceinline__ void run(f32x4_t* a, f32x4_t* a2, const uint32_t addr, const uint32_t str, const u32x4_t xb, const u32x4_t xb2) {
    constexpr int U = NT - N0 < UB ? NT - N0 : UB;
    MfmaBatch<U, PAIR>::run(a + N0, a2 + N0, addr + N0 * str, str, xb, xb2);
    if constexpr (N0 + U < NT) MfmaSeq<NT, UB, N0 + U, PAIR>::run(a, a2, addr, str, xb, xb2);
  }
};

constexpr bool PROis(int pro) { return pro == C3D_PRO_AFFINE2; }

__device__ __forceinline__ int64_t row_offset(const c3d_pw_args& a, uint32_t um) {
  if (a.row_mode == C3D_ROWS_DENSE) return (int64_t)um * a.Kp;
  if (a.row_mode == C3D_ROWS_FRAME) {
    const uint32_t g = um / (uint32_t)a.rpg;
    const uint32_t r = um - g * (uint32_t)a.rpg;
    return (int64_t)g * a.gstride + (int64_t)r * a.Kp;
  }
  // STRIDE2: m = (bt, ho, wo) over output [BT][H/2][W/2]
  const uint32_t Wo = (uint32_t)a.W >> 1, Ho = (uint32_t)a.H >> 1;
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

// raw (unconverted) 8-element vectors kept in registers while a prefetch is in flight
template <typename T> struct Raw;
template <> struct Raw<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ type zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct Raw<float> {
  struct type { float4 a, b; };
  static __device__ __forceinline__ type load(const float* p) {
    type t; t.a = *reinterpret_cast<const float4*>(p); t.b = *reinterpret_cast<const float4*>(p + 4); return t;
  }
  static __device__ __forceinline__ type zero() { type t; t.a = make_float4(0, 0, 0, 0); t.b = t.a; return t; }
  static __device__ __forceinline__ void cvt(const type& v, float (&f)[8]) {
    f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
  }
};

// Bounds-checked buffer addressing for every global access of the tile loop (round 5).  The loop used to guard its loads
// and stores with exec-masked branches (`if (row exists) x = load; else x = 0`); the compiler's s_waitcnt insertion then
// fell back to vmcnt(0) at the joins -- in the ISA of the round-4 kernels EVERY prefetch slot of the Swish forward waited for
// the slot re-issued just before it (one memory round trip per 1 KB: 7 per 16 x 216 tile), the burst prefetch of the other
// variants went out behind vmcnt(1) (two loads in flight), and each epilogue pass waited for the previous pass's store and the
// whole next-tile prefetch.  With raw buffer instructions a lane that must not touch memory gets a byte offset past
// num_records instead of a cleared exec bit: loads return 0 (exactly the RW::zero() the branches produced), stores are
// dropped, and the code is straight-line, so the counted waits are exact.  Offsets are 32-bit: the entry point refuses
// tensors of 2 GiB or more (PW_OOB = 2^31 is the "nowhere" offset).
typedef uint32_t pw_u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t PW_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pw_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
template <typename T> struct BufIO;
template <> struct BufIO<bf16_t> {
  static __device__ __forceinline__ uint4 load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const pw_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ void store_raw(__amdgpu_buffer_rsrc_t r, uint32_t off, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(pw_u32x4_t{v.x, v.y, v.z, v.w}, r, off, 0, 0);
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, const float (&f)[8]) {
    __builtin_amdgcn_raw_buffer_store_b128(pw_u32x4_t{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                                                      pack_bf16x2(f[6], f[7])}, r, off, 0, 0);
  }
};
template <> struct BufIO<float> {
  static __device__ __forceinline__ Raw<float>::type load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const pw_u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    const pw_u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u, 0, 0);   // PW_OOB + 16 is still out of range
    Raw<float>::type t;
    t.a = make_float4(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]));
    t.b = make_float4(__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3]));
    return t;
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, const float (&f)[8]) {
    __builtin_amdgcn_raw_buffer_store_b128(pw_u32x4_t{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])}, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(pw_u32x4_t{__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7])}, r, off + 16u, 0, 0);
  }
};

__device__ __forceinline__ void lds_ld8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

typedef __attribute__((address_space(3))) void* pw_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pw_glb_ptr_t;

// Weight matrix (f32, any strides) -> LDS image Ws[n][k] in the MFMA operand type.  VW = floats per global load
// along the contiguous dimension, KC = that dimension is K (then the VW elements are adjacent in LDS too: one
// 8-byte / 16-byte write).  All of a thread's loads are issued before the first LDS write: one round trip.
template <typename T, int VW, bool KC, int WAVES>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, typename Mma<T>::lds_t* Ws, int CL, int OLn,
                                              int ostride, int KL, int NR, int tid) {
  typedef Mma<T> MM;
  typedef typename MM::lds_t lds_t;
  constexpr int WB = 8;
  const int vpr = CL / VW, total = OLn * vpr;
  const float inv = 1.0f / (float)vpr;
  for (int base = tid; base < total; base += WAVES * 64 * WB) {
    float v[WB][VW];
    int oo[WB], ii[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int idx = base + u * WAVES * 64;
      oo[u] = -1;
      if (idx < total) {
        const int o = __float2int_rz(((float)idx + 0.5f) * inv);
        const int i = idx - o * vpr;
        oo[u] = o; ii[u] = i * VW;
        const float* src = w + (size_t)o * ostride + i * VW;
        if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(src); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
        else if constexpr (VW == 2) { const float2 t = *reinterpret_cast<const float2*>(src); v[u][0] = t.x; v[u][1] = t.y; }
        else v[u][0] = src[0];
      }
    }
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      if (oo[u] >= 0) {
        if constexpr (KC && VW == 4) {
          lds_t* dst = Ws + MM::widx(oo[u], ii[u], KL, NR);
          if (sizeof(lds_t) == 2) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], v[u][3]));
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            const int n = KC ? oo[u] : ii[u] + e;
            const int k = KC ? ii[u] + e : oo[u];
            Ws[MM::widx(n, k, KL, NR)] = MM::cvt(v[u][e]);
          }
        }
      }
    }
  }
}

// raw 8-channel vectors per lane per iteration (~8 KB per wave in flight).  The two-tensor prologue doubles the
// registers per slot; its heaviest forms (the Swish/SE-backward epilogue's 64 per-lane parameters and sums, or
// 14 accumulator tiles) get 6 slots so that 8 waves per CU fit 256 VGPRs without spilling.
// sub-tiles per weight-fragment read (2 where acc[2][NT] fits next to the prefetch and epilogue registers)
template <typename T, int NT, int PRO, int EPI> struct PwPair {
  static constexpr int value = (sizeof(T) == 2 && ((NT == 7 && !(PRO == C3D_PRO_AFFINE2 && EPI == C3D_EPI_SWISH_SE_BWD)) ||
                                                   (NT == 14 && PRO == C3D_PRO_NONE))) ? 2 : 1;
};
// weight fragments in flight per batch of the hand-scheduled MFMA pipeline (4 VGPRs each); 0 = leave the loop to
// the compiler (the widest Swish/SE-backward variant sits at 256 VGPRs already)
template <int NT, int PRO, int EPI> struct PwFragBatch {
  static constexpr int value = (NT == 14 && PRO == C3D_PRO_AFFINE2 && EPI == C3D_EPI_SWISH_SE_BWD) ? 0 : (NT < 7 ? NT : 7);
};
template <int NT, int PRO, int EPI, int WAVES, int WG = 0> struct PwSlots {
  // (the fused weight gradient of the Swish/SE-backward variant needs ~16 more registers: 4 slots keep it out of scratch)
  static constexpr int value = (WG == C3D_WG_SWISH) ? 4 :
                               (PRO == C3D_PRO_AFFINE2 && WAVES == 8 && (NT == 14 || EPI == C3D_EPI_SWISH_SE_BWD)) ? 6 : 8;
};

// Output staging type: plain-store / statistics epilogues round once to the storage type anyway,
// the arithmetic epilogues (Swish/SE backward, residual add) keep the f32 accumulator.
template <typename T, int EPI> struct OutStage { typedef float type; };
template <int EPI> struct OutStage<bf16_t, EPI> { typedef bf16_t type; };  // halves the result tile: 8 waves/CU fit

struct PwLaunch {
  int tiles_per_wave;  // 16-row sub-tiles per wave (contiguous range)
  int tpi;             // sub-tiles per iteration (prefetch batch)
  int xs_rows;         // rows of the wave's X region (= tpi*16)
  int w_off, p_off, wave_off, wave_bytes, os_off, gs_off;  // byte offsets in dynamic LDS
  int flush_shuffle_max;  // row-lanes per channel vector up to which the final sums are shuffled instead of dumped
  int se_off;             // consumer-side SqueezeExcitation gate (a.se_w1): byte offset of the [PW_SE_NS][Kp] gates + [PW_SE_NS][32] hidden units
  int wg_pair;            // fused weight gradient: 1 = a second result-tile buffer per wave (os_off2), tiles are taken in pairs
  int os_off2;
  int dw_off, ldw;        // fused weight gradient (WG != 0): byte offset and row stride (floats) of the workgroup's dW accumulators
};

// ---- weight gradient fused into the data-gradient launch (c3d_pw_args.wg_mode) -------------------------------------
// dW[k][n] = sum_rows P[row][k] * Q[row][n]: both operands of a 16-row tile sit in this wave's LDS regions in
// [row][channel] layout (P = the prologue'd rows staged for the GEMM, Q = written over the result tile by the epilogue
// passes); gfx950's transposing LDS read (ds_read_b64_tr_b16: lane l, element j <- img[4*(l/16) + j][c0 + l%16] when
// lane l addresses the 8-byte chunk (row 4*(l/16) + (l%16)/4, columns c0 + 4*(l%4) ..)) delivers them as the A / B
// fragments of v_mfma_f32_16x16x16_bf16 with the ROWS as the contraction index.  The products are added to a
// workgroup-shared F64 accumulator image in LDS (ds_add_f64): the data-gradient variants sit at 230-256 VGPRs, per-wave
// register accumulators (32 / 84 per lane on res2 / res3) do not fit beside them, 12 transient registers do.
// (tools/micro/lds_atomic_rate.hip on MI355X: ds_add_f32 takes ~190 clocks per wave-instruction -- 48x ds_add_u32, the first
// version of this kernel ran 10x slower than the pair it replaces -- ds_add_f64 takes ~8, ds_add_u32 ~4.  With f64 sums the
// order in which the waves' contributions land no longer shows in the f32 result.)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;
constexpr int PW_SE_NS = 4;     // samples a workgroup may span with the consumer-side SE gate (else: separate c3d_bn_se_finalize)
constexpr int PW_SE_CR = 32;    // hidden units (se_cr) at most
constexpr int PW_E_SE_FALLBACK = -1000;   // internal: launch_pw_d -> c3d_pw_gemm
constexpr int WG_NTP_MAX = 7;   // K tiles (P channels) held as fragments: Kp <= 112 (C3D_WG_ROWS: conv_a, K = inner channels)
constexpr int WG_NTP_MAX_SWISH = 3;   // C3D_WG_SWISH: conv_c, K = the block's output channels (<= 48 on res2 / res3)


// DENSE = rows are consecutive in memory (row_mode C3D_ROWS_DENSE): a 16-row tile is one contiguous span, so
// every lane's load address is (wave-uniform tile base) + lane*16 B + constant -- no per-slot index arithmetic,
// no 64-bit vector multiplies (the generic path's row_offset() code cost ~50 VGPRs and spilled).
template <typename T, int NT, int PRO, int EPI, int WAVES, bool DENSE, int WG = 0>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 2 : 1, 2))) void pw_gemm_kernel(const c3d_pw_args a, const PwLaunch L) {
  static_assert(WG == 0 || (sizeof(T) == 2 && PRO == C3D_PRO_AFFINE2 && DENSE &&
                            ((WG == C3D_WG_SWISH && EPI == C3D_EPI_SWISH_SE_BWD) ||
                             ((WG == C3D_WG_ROWS || WG == C3D_WG_MASKSUM) && EPI == C3D_EPI_ADD))),
                "fused weight gradient / fused block-output backward: bf16 data-gradient variants only");
  // WGRAD: the weight gradient proper.  X3F: the previous block's output rows (wg_x3) ride along one pass ahead -- as the Q
  // operand of the weight gradient (C3D_WG_ROWS), as the ReLU mask of the stored output, and for the BatchNorm_c-backward
  // sums of that block (c3d_pw_args.add_sums: c3d_block_out_bwd folded into this epilogue)
  constexpr bool WGRAD = WG == C3D_WG_SWISH || WG == C3D_WG_ROWS;
  constexpr bool X3F = WG == C3D_WG_ROWS || WG == C3D_WG_MASKSUM;
  typedef Mma<T> MM;
  typedef typename MM::lds_t lds_t;
  typedef Raw<T> RW;
  typedef typename OutStage<T, EPI>::type os_t;
  constexpr int PW_SLOTS = PwSlots<NT, PRO, EPI, WAVES, WG>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile indices stay in SGPRs
  CLK_DECL
  const int Kp = a.Kp, Np = a.Np;
  const int Kpad = (Kp + MM::KSTEP - 1) / MM::KSTEP * MM::KSTEP;
  const int KL = Kpad + MM::KPAD;
  constexpr int NL = NT * 16 + (sizeof(os_t) == 4 ? 4 : 8);
  const int KS = Kpad / MM::KSTEP;

  lds_t* Ws = reinterpret_cast<lds_t*>(smem + L.w_off);
  float* Pp = reinterpret_cast<float*>(smem + L.p_off);  // prologue parameters [3][Kp]
  unsigned char* wreg = smem + L.wave_off + (size_t)wave * L.wave_bytes;
  lds_t* Xs = reinterpret_cast<lds_t*>(wreg);
  os_t* const OsA = reinterpret_cast<os_t*>(wreg + L.os_off);
  os_t* const OsB = (WGRAD && L.wg_pair) ? reinterpret_cast<os_t*>(wreg + L.os_off2) : OsA;   // odd tiles of an iteration (pairs)
  float* Gs = reinterpret_cast<float*>(wreg + L.gs_off);  // gate of the current sample [Kp]
  double* dWs = reinterpret_cast<double*>(smem + L.dw_off);  // WG: f64 dW accumulators [ceil(Kp/16)*16][L.ldw], shared by the waves
  const int wg_ntp = (Kp + 15) >> 4, wg_ntq = (Np + 15) >> 4;

  // The first tile's rows (and the per-lane epilogue parameters) are requested BEFORE the weights are
  // staged: a launch is a chain of dependent global round trips (parameters, weights, first tile,
  // epilogue operands: ~16 us even for a 4-workgroup grid), so the independent ones must overlap.
  // ---- lane maps -----------------------------------------------------------------------------
  const int Gi = Kp >> 3;                       // input: flat map, slot i = lane + 64*q over 16*Gi vectors
  const int Q = (Gi + 3) >> 2;                  // vector slots per lane per 16-row sub-tile
  const float invGi = 1.0f / (float)Gi;
  const int Go = Np >> 3, RPo = 64 / Go;        // output: (row-in-pass, fixed channel vector) map
  const bool act_o = lane < Go * RPo;
  const int rr_o = lane / Go, v_o = lane - rr_o * Go;
  int slot_s[PW_SLOTS], slot_q[PW_SLOTS];
  {
    int s_ = 0, q_ = 0;
#pragma unroll
    for (int j = 0; j < PW_SLOTS; ++j) { slot_s[j] = s_; slot_q[j] = q_; if (++q_ == Q) { q_ = 0; ++s_; } }
  }
  const int nslots = L.tpi * Q;

  // per-lane epilogue parameters / accumulators
  // (round 5: the Swish/SE-backward epilogue's per-channel scale | shift | mean | rstd live in LDS behind the prologue
  // parameters, [4][Np], and are read per pass: as 32 registers per lane they pushed the wide variants into scratch -- and every
  // scratch reload is a VMEM access behind s_waitcnt vmcnt(0), in the middle of the prefetch burst)
  float* const Ep = Pp + 3 * Kp;
  float eG[8];
  float s0[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { eG[j] = 1.f; s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }

  const int M32 = (int)a.M;                    // M < 2^31 (checked by the entry point)
  const int tiles = (M32 + 15) >> 4;
  const int gw = (int)blockIdx.x * WAVES + wave;
  int t0 = gw * L.tiles_per_wave;
  if (t0 > tiles) t0 = tiles;
  int t1 = t0 + L.tiles_per_wave;
  if (t1 > tiles) t1 = tiles;
  const uint32_t rps32 = a.rows_per_sample > 0 ? (uint32_t)a.rows_per_sample : 1u;
  const int64_t nmax = (int64_t)(((uint32_t)a.M - 1u) / rps32);
  // first sample of this workgroup's rows (consumer-side SE gates are indexed from it)
  const int se_n_lo = (int)((uint32_t)(((int)blockIdx.x * WAVES * L.tiles_per_wave < tiles ? (int)blockIdx.x * WAVES * L.tiles_per_wave : tiles) << 4) / rps32);
  int cur_n = -1;       // sample whose partial sums are accumulated (SWISH_SE_BWD epilogue)
  int gate_n = -1;      // sample whose gate is cached in Gs (BN_SE_SWISH prologue)

  const T* X = reinterpret_cast<const T*>(a.x);
  const T* X2 = reinterpret_cast<const T*>(a.x2);
  const T* E1 = reinterpret_cast<const T*>(a.e1);
  T* Y = reinterpret_cast<T*>(a.y);
  // Buffer resources (wave-uniform: scalar registers).  The row streams are bounded by THIS wave's rows: a tile past t1
  // (another wave's) or a row past M reads zeros, which is what the exec-masked loads assigned by hand.
  constexpr uint32_t ES = (uint32_t)sizeof(T);
  const uint32_t lim_rows = (uint32_t)(t1 * 16 < M32 ? t1 * 16 : M32);
  const __amdgpu_buffer_rsrc_t rX = pw_rsrc(X, lim_rows * (uint32_t)Kp * ES);
  const __amdgpu_buffer_rsrc_t rX2 = pw_rsrc(PRO == C3D_PRO_AFFINE2 ? X2 : nullptr, lim_rows * (uint32_t)Kp * ES);
  const __amdgpu_buffer_rsrc_t rY = pw_rsrc(Y, (uint32_t)M32 * (uint32_t)Np * ES);
  const __amdgpu_buffer_rsrc_t rPO = pw_rsrc(PRO == C3D_PRO_AFFINE2 ? a.pro_out : nullptr, (uint32_t)M32 * (uint32_t)Kp * ES);
  const uint32_t lane_b = (uint32_t)lane * 8u * ES;   // this lane's 8-element vector inside a 64-vector (1 KB / 2 KB) piece

  typename RW::type xr[PW_SLOTS];
  typename RW::type x2r[PROis(PRO) ? PW_SLOTS : 1];

#define PW_ISSUE_GENERIC(TILE0)                                                                \
  _Pragma("unroll") for (int j = 0; j < PW_SLOTS; ++j) {                                       \
    if (j < nslots) {                                                                          \
      const int i_ = lane + 64 * slot_q[j];                                                    \
      const int row_ = __float2int_rz(((float)i_ + 0.5f) * invGi);                             \
      const int v_ = i_ - row_ * Gi;                                                           \
      const int64_t tl_ = (TILE0) + slot_s[j];                                                 \
      const int64_t m_ = (tl_ << 4) + row_;                                                    \
      if (row_ < 16 && tl_ < t1 && m_ < a.M) {                                                 \
        const int64_t off_ = row_offset(a, (uint32_t)m_) + v_ * 8;                             \
        xr[j] = RW::load(X + off_);                                                            \
        if (PRO == C3D_PRO_AFFINE2) x2r[PROis(PRO) ? j : 0] = RW::load(X2 + off_);             \
      } else {                                                                                 \
        xr[j] = RW::zero();                                                                    \
        if (PRO == C3D_PRO_AFFINE2) x2r[PROis(PRO) ? j : 0] = RW::zero();                      \
      }                                                                                        \
    }                                                                                          \
  }
  // byte offset of slot J of the iteration starting at tile TILE0 (wave-uniform part in scalar registers + lane_b); slots
  // the plan does not use go nowhere.  No branch, no exec mask: see BufIO.
#define PW_SLOT_OFF(TILE0, J)                                                                  \
  ((J) < nslots ? ((uint32_t)((TILE0) + slot_s[J]) * 16u * (uint32_t)Kp + 512u * (uint32_t)slot_q[J]) * ES + lane_b : PW_OOB)
#define PW_ISSUE_DENSE(TILE0)                                                                  \
  {                                                                                            \
    _Pragma("unroll") for (int j = 0; j < PW_SLOTS; ++j) {                                     \
      const uint32_t o_ = PW_SLOT_OFF(TILE0, j);                                               \
      xr[j] = BufIO<T>::load(rX, o_);                                                          \
      if (PRO == C3D_PRO_AFFINE2) x2r[PROis(PRO) ? j : 0] = BufIO<T>::load(rX2, o_);           \
    }                                                                                          \
  }
#define PW_ISSUE(TILE0) if constexpr (DENSE) PW_ISSUE_DENSE(TILE0) else { PW_ISSUE_GENERIC(TILE0) }
  // One slot of PW_ISSUE_DENSE (the Swish-prologue forward re-requests a slot as soon as its registers are free: see the
  // convert loop)
#define PW_ISSUE_DENSE_SLOT(TILE0, J, ON) xr[J] = BufIO<T>::load(rX, (ON) ? PW_SLOT_OFF(TILE0, J) : PW_OOB);
  // conv_c forward (BatchNorm x SE x Swish on load: 13 VALU units per element, as long as the memory time of the tile): the
  // next iteration's rows are requested slot by slot, each right after the prologue that consumed its registers, so the
  // requests of the first slots are in flight under the activation arithmetic of the others
  constexpr bool SLOT_REISSUE = DENSE && PRO == C3D_PRO_BN_SE_SWISH && WG == 0 && sizeof(T) == 2;

  PW_ISSUE(t0)   // (a wave without tiles reads zeros: its row bound is empty)
  CLK(9)

  // ---- stage weights: zero fill, then vectorised fill along W's contiguous dimension -------
  {
    const int wtot = NT * 16 * KL;  // elements, multiple of 8
    const bool img = a.w_img != nullptr;
    if (img) {
      // Packed image (c3d_pw_pack_weights): the LDS layout itself, copied by LDS-DMA in 1 KB chunks per wave -- no
      // zero fill, no f32 -> operand-type conversion, no scattered 2-byte LDS writes for the transposed (data-gradient)
      // orientation, half the bytes.  Every workgroup of the launch reads the same image at the same moment: the
      // chunk order is rotated by the workgroup index so that the requests spread over the L2 channels.
      const int wbytes = (sizeof(T) == 2 ? NT * 16 * Kpad : wtot) * (int)sizeof(lds_t);   // (chunk-major bf16 image: no padding chunk read)
      const int nchunk = (wbytes + 1023) >> 10;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w_img);
      const int rot = (int)blockIdx.x % nchunk;
      for (int c = wave; c < nchunk; c += WAVES) {
        int r = c + rot;
        if (r >= nchunk) r -= nchunk;
        const int off = r * 1024 + lane * 16;
        int goff = off;
        if constexpr (sizeof(T) == 2) {
          // the image was packed for c3d_pw_pack_weights' row bucket (2 / 4 / 7 / 14 tiles of 16 rows per k-chunk); a kernel on
          // fewer output tiles (conv_a of res3: NT = 3 of the NT = 4 image) copies the first NT * 16 rows of every chunk
          const int ntn = (Np + 15) >> 4;
          const int img_rows = (ntn <= 2 ? 2 : ntn <= 4 ? 4 : ntn <= 7 ? 7 : 14) * 16;
          if (img_rows != NT * 16) {
            const int ch = off / (NT * 16 * 16);
            goff = ch * img_rows * 16 + (off - ch * (NT * 16 * 16));
          }
        }
        if (off < wbytes)
          __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(src + goff), (pw_lds_ptr_t)(smem + L.w_off + r * 1024), 16, 0, 0);
      }
    } else {
      for (int i = tid * 8; i < wtot; i += WAVES * 64 * 8) {
        float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        MM::store8(Ws + i, z);
      }
    }
    if (PRO == C3D_PRO_AFFINE2 && a.fin.sums && a.fin.training) {
      // forward residual add fused into this GEMM's prologue (pro_out): A|B = scale|shift of the previous block's
      // BatchNorm_c rebuilt from its completed sums (csrc/bn_fin.h bn_consume: workgroup 0 owns the saved vectors and
      // the running statistics), C = 1 for the shortcut operand
      if (blockIdx.x == 0 && tid == 0 && a.fin.nbt) *a.fin.nbt += 1;
      c3dfin::bn_consume(a.fin, a.K, Kp, 0, Kp, blockIdx.x == 0, Pp, Pp + Kp, tid, WAVES * 64);
      for (int c = tid; c < Kp; c += WAVES * 64) Pp[2 * Kp + c] = 1.f;
    } else if (PRO == C3D_PRO_AFFINE2 && a.fin.sums) {
      // BatchNorm-backward coefficients rebuilt from the producer's completed sums (no c3d_bn_bwd_coef launch in
      // front of this kernel); workgroup 0 also accumulates dgamma / dbeta and writes the vector for other readers
      for (int c = tid; c < Kp; c += WAVES * 64) {
        float cA, cB, cC;
        c3dfin::bn_bwd_coef_consume(a.fin, a.K, Kp, c, blockIdx.x == 0, cA, cB, cC);
        Pp[c] = cA; Pp[Kp + c] = cB; Pp[2 * Kp + c] = cC;
      }
    } else if (PRO == C3D_PRO_BN_SE_SWISH && a.fin.sums) {
      // BatchNorm_b of a block without SqueezeExcitation, finalised here from the depthwise kernel's per-sample sums (no
      // c3d_bn_se_finalize launch between conv_b and conv_c); workgroup 0 owns ss / mr / the running statistics
      if (blockIdx.x == 0 && tid == 0 && a.fin.nbt) *a.fin.nbt += 1;
      c3dfin::bn_consume_nc(a.fin, a.K, Kp, blockIdx.x == 0, Pp, Pp + Kp, tid, WAVES * 64);
      if (a.se_w1) {
        // SE gate of the samples this workgroup's rows belong to (at most PW_SE_NS: the launcher checked), while the first
        // tile's rows are in flight; the workgroup holding a sample's first row writes gate / hid for the backward pass
        const int tiles_ = ((int)a.M + 15) >> 4;
        int wt0 = (int)blockIdx.x * WAVES * L.tiles_per_wave, wt1 = wt0 + WAVES * L.tiles_per_wave;
        if (wt0 > tiles_) wt0 = tiles_;
        if (wt1 > tiles_) wt1 = tiles_;
        if (wt0 < wt1) {
          const uint32_t rps_ = (uint32_t)a.rows_per_sample;
          const int r0 = wt0 << 4, r1 = ((wt1 << 4) < (int)a.M ? (wt1 << 4) : (int)a.M) - 1;
          const int n_lo = (int)((uint32_t)r0 / rps_), n_hi = (int)((uint32_t)r1 / rps_);
          const int own_lo = (int)(((uint32_t)r0 + rps_ - 1) / rps_);   // first sample whose row 0 is >= r0
          float* zg = reinterpret_cast<float*>(smem + L.se_off);
          c3dfin::se_gate_consume(a.fin.sums, (double)a.rows_per_sample, a.K, Kp, a.se_w1, a.se_b1, a.se_w2, a.se_b2, a.se_cr, n_lo,
                                  n_hi - n_lo + 1, own_lo, n_hi, Pp, Pp + Kp, zg, zg + PW_SE_NS * Kp,
                                  const_cast<float*>(a.pro_gate), a.se_hid, tid, WAVES * 64);
        }
      }
    } else if (PRO != C3D_PRO_NONE) {
      const int np = (PRO == C3D_PRO_AFFINE2 ? 3 : 2) * Kp;
      for (int i = tid; i < np; i += WAVES * 64) Pp[i] = a.pro_p[i];
    }
    if constexpr (EPI == C3D_EPI_SWISH_SE_BWD) {
      for (int i = tid; i < 2 * Np; i += WAVES * 64) { Ep[i] = a.epi_p[i]; Ep[2 * Np + i] = a.epi_q[i]; }
    }
    if constexpr (X3F) {
      if (a.add_sums)   // mean | rstd of the previous block's BatchNorm_c, read per epilogue pass
        for (int i = tid; i < 2 * Np; i += WAVES * 64) Ep[i] = a.add_mr[i];
    }
    CLK(10)
    if (img) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA chunks (and the first tile's rows, issued earlier)
    } else {
      __syncthreads();
      CLK(11)
      const bool kc = (a.w_sk == 1);
      const int CL = kc ? a.K : a.N, OLn = kc ? a.N : a.K;
      const int ostride = kc ? a.w_sn : a.w_sk;
      if ((CL & 3) == 0 && (ostride & 3) == 0 && ((uintptr_t)a.w & 15) == 0) {
        if (kc) stage_weights<T, 4, true, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
        else stage_weights<T, 4, false, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
      } else if ((CL & 1) == 0 && (ostride & 1) == 0 && ((uintptr_t)a.w & 7) == 0) {
        if (kc) stage_weights<T, 2, true, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
        else stage_weights<T, 2, false, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
      } else {
        if (kc) stage_weights<T, 1, true, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
        else stage_weights<T, 1, false, WAVES>(a.w, Ws, CL, OLn, ostride, KL, NT * 16, tid);
      }
    }
    if constexpr (WGRAD) {
      for (int i = tid; i < wg_ntp * 16 * L.ldw; i += WAVES * 64) dWs[i] = 0.0;
    }
    CLK(12)
    __syncthreads();
    CLK(13)
  }

  // each wave zeroes its X region once: the K-padding columns [Kp, Kpad) are never written later
  for (int i = lane * 8; i < L.xs_rows * KL; i += 64 * 8) {
    float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MM::store8(Xs + i, z);
  }

  // Companion rows (E1) of the arithmetic epilogues are requested ONE PASS AHEAD, the first pass of a tile before
  // its MFMAs (and before the next iteration's prefetch burst: vmcnt retires in order).  The pass loop used to
  // issue load -> s_waitcnt vmcnt(0) -> ~140 VALU per pass: one exposed memory round trip per pass, 8 per tile.
  constexpr bool E1_PIPE = EPI == C3D_EPI_SWISH_SE_BWD || EPI == C3D_EPI_ADD;
  const bool e1_rows = E1_PIPE && (EPI == C3D_EPI_SWISH_SE_BWD || a.res_mode == 0);   // E1 has Y's row layout
  const int npass = (16 + RPo - 1) / RPo;
  constexpr int RPO_MIN = 64 / (2 * NT), NPASS_MAX = (16 + RPO_MIN - 1) / RPO_MIN;
  const int v_oc = act_o ? v_o : 0;             // clamped: inactive lanes load a valid address (result unused)
  typename RW::type e1n = RW::zero();
  typename RW::type x3n = RW::zero();   // X3F: this lane's row vector of wg_x3 (Y's row layout), one pass ahead like e1n
  typename RW::type c1n = RW::zero();   // X3F with add_sums: the same lane vector of add_c (the previous block's conv_c output)
  const T* X3 = reinterpret_cast<const T*>(a.wg_x3);
  // this lane's E1 vector for pass P of the tile starting at ROW0: wave-uniform row base (scalar registers) + a
  // loop-invariant 32-bit lane offset; lanes without a row in this pass read the pass's first row (result unused)
  // (byte offsets into the bounds-checked resources rE1 / rX3: the requests are unconditional buffer loads, a resource
  // without rows -- res_mode != 0, or no companion at all -- answers zeros)
  const uint32_t e1_lane_b = (uint32_t)(rr_o * Np + v_oc * 8) * ES, e1_v_b = (uint32_t)(v_oc * 8) * ES;
#define PW_E1_OFF(ROW0, P)                                                                                          \
  ((uint32_t)((ROW0) + (P) * RPo < M32 ? (ROW0) + (P) * RPo : M32 - 1) * ((uint32_t)Np * ES) +                       \
   (((P) * RPo + rr_o < 16 && (ROW0) + (P) * RPo + rr_o < M32) ? e1_lane_b : e1_v_b))
  const __amdgpu_buffer_rsrc_t rE1 = pw_rsrc(e1_rows ? E1 : nullptr, (uint32_t)M32 * (uint32_t)Np * ES);
  const __amdgpu_buffer_rsrc_t rX3 = pw_rsrc(X3F ? X3 : nullptr, (uint32_t)M32 * (uint32_t)Np * ES);
  const bool add_sums = X3F && a.add_sums != nullptr;   // kernel-uniform
  const __amdgpu_buffer_rsrc_t rC1 = pw_rsrc(add_sums ? reinterpret_cast<const T*>(a.add_c) : nullptr, (uint32_t)M32 * (uint32_t)Np * ES);

  // Weight fragments: narrow outputs (NT <= 4) with K <= 64 keep ALL of them in registers (the per-tile MFMA phase
  // was LDS-read latency: X fragment, then each weight fragment, serially); wide outputs run two sub-tiles per
  // weight-fragment read when the register budget allows (the stage-3 MFMA phase was LDS-bandwidth bound: 8 waves
  // x 1 KB per MFMA).
  constexpr bool WREG = NT <= 4 && sizeof(T) == 2 && !(PRO == C3D_PRO_AFFINE2 && EPI == C3D_EPI_SWISH_SE_BWD);
  // (C3D_WG_MASKSUM serves the wide-K conv_a data gradients -- res4: Kp = 216 = 7 slots per tile, one tile per iteration --
  // where a second accumulator set would never be used: its 4 NT registers go to the epilogue's sums instead)
  constexpr int MT = WG == C3D_WG_MASKSUM ? 1 : PwPair<T, NT, PRO, EPI>::value;
  typename MM::frag_t wr[WREG ? 2 * NT : 1];
  if (WREG && KS <= 2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wr[WREG ? nt : 0] = MM::loadw(Ws, nt * 16 + (lane & 15), 0, KL, NT * 16, lane);
      wr[WREG ? NT + nt : 0] = MM::loadw(Ws, nt * 16 + (lane & 15), KS == 2 ? 1 : 0, KL, NT * 16, lane);
    }
  }

  CLK(0)
  for (int it0 = t0; it0 < t1; it0 += L.tpi) {
    CLK_WAITVM
    CLK(1)
    const bool reissue = SLOT_REISSUE && it0 + L.tpi < t1;
    // ---------------- convert + prologue -> LDS (all sub-tiles of this iteration) ------------
#pragma unroll
    for (int j = 0; j < PW_SLOTS; ++j) {
      const int i_ = lane + 64 * slot_q[j];
      const int row_ = __float2int_rz(((float)i_ + 0.5f) * invGi);
      const int v_ = i_ - row_ * Gi;
      const int tl_ = it0 + slot_s[j];
      [[maybe_unused]] uint32_t po_off = PW_OOB;   // where this slot's fused residual output goes (nowhere unless a real row)
      [[maybe_unused]] float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (j < nslots && row_ < 16) {
        if constexpr (sizeof(T) == 2 && PRO == C3D_PRO_NONE) {   // no prologue: the raw bf16 vector IS the operand
          *reinterpret_cast<uint4*>(Xs + (slot_s[j] * 16 + row_) * KL + v_ * 8) = xr[j];
        } else {
          RW::cvt(xr[j], f);
          if (PRO == C3D_PRO_BN_SE_SWISH) {
            float sc[8], sh[8], g[8];
            lds_ld8(Pp + v_ * 8, sc);
            lds_ld8(Pp + Kp + v_ * 8, sh);
            if (a.se_w1 && a.fin.sums) {   // gates of this workgroup's samples were computed in the prologue (LDS)
              int n_ = (int)((uint32_t)(tl_ << 4) / rps32);
              if (n_ > (int)nmax) n_ = (int)nmax;
              lds_ld8(reinterpret_cast<const float*>(smem + L.se_off) + (n_ - se_n_lo) * Kp + v_ * 8, g);
            } else if (a.pro_gate) {
              int n_ = (int)((uint32_t)(tl_ << 4) / rps32);
              if (n_ > (int)nmax) n_ = (int)nmax;
              if (n_ != gate_n) {  // wave-uniform: a sub-tile never straddles two samples
                for (int c = lane; c < Kp; c += 64) Gs[c] = a.pro_gate[(int64_t)n_ * Kp + c];
                gate_n = n_;
              }
              lds_ld8(Gs + v_ * 8, g);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] = 1.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float q = g[e] * fmaf(f[e], sc[e], sh[e]);
              f[e] = q * sigmoid_t<T>(q);
            }
          } else if (PRO == C3D_PRO_AFFINE2) {
            float f2[8], cA[8], cB[8], cC[8];
            RW::cvt(x2r[PROis(PRO) ? j : 0], f2);
            lds_ld8(Pp + v_ * 8, cA);
            lds_ld8(Pp + Kp + v_ * 8, cB);
            lds_ld8(Pp + 2 * Kp + v_ * 8, cC);
            const bool real = tl_ < t1 && ((tl_ << 4) + row_) < M32;
            if (a.pro_out) {
              // y = relu(bn_c(c) + shortcut) of the previous residual block, in the association of c3d_block_out_fwd
              // (bit-identical), also written out: the next block's shortcut and the backward pass read it
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] = real ? fmaxf(fmaf(f[e], cA[e], cB[e]) + fmaf(f2[e], cC[e], 0.f), 0.f) : 0.f;
              if (real) po_off = ((uint32_t)((tl_ << 4) + row_) * (uint32_t)Kp + (uint32_t)v_ * 8u) * ES;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] = real ? fmaf(cA[e], f[e], fmaf(cC[e], f2[e], cB[e])) : 0.f;
            }
          }
          MM::store8(Xs + (slot_s[j] * 16 + row_) * KL + v_ * 8, f);
        }
      }
      if constexpr (PRO == C3D_PRO_AFFINE2) {
        if (a.pro_out) BufIO<T>::store(rPO, po_off, f);   // (kernel-uniform condition; lanes without a real row store nowhere)
      }
      // conv_c forward: the slot's registers are free -- request the same slot of the next iteration now
      if constexpr (SLOT_REISSUE) PW_ISSUE_DENSE_SLOT(it0 + L.tpi, j, reissue)
    }
    CLK(2)
    if constexpr (E1_PIPE) e1n = BufIO<T>::load(rE1, PW_E1_OFF(it0 << 4, 0));
    if constexpr (X3F) { x3n = BufIO<T>::load(rX3, PW_E1_OFF(it0 << 4, 0)); c1n = BufIO<T>::load(rC1, PW_E1_OFF(it0 << 4, 0)); }
    // ---------------- prefetch the next iteration's rows -------------------------------------
    // (DENSE: unconditionally -- past this wave's last tile the row bound answers zeros without touching memory; a prefetch
    // under `if (more tiles)` left the epilogue's counted waits with two histories to cover: they came out as vmcnt(1))
    if constexpr (!SLOT_REISSUE) {
      if constexpr (DENSE) { PW_ISSUE_DENSE(it0 + L.tpi) }
      else if (it0 + L.tpi < t1) { PW_ISSUE_GENERIC(it0 + L.tpi) }
    }
    CLK(3)

    // stage one 16-row result tile to Os, run the fused epilogue over it, store
    // wg_prev: the first tile of a weight-gradient PAIR (its P rows and its Q tile in the other result buffer), or nullptr;
    // wg_defer: this tile's weight-gradient step is left to the next tile of the iteration (which passes it as wg_prev)
    auto finish_tile = [&](const f32x4_t (&acc)[NT], const int row0, const bool has_next, const lds_t* xt, os_t* const Os,
                           const lds_t* const wg_prev_x, const os_t* const wg_prev_q, const bool wg_defer) {
      // ---------------- stage result tile: Os[row = lane&15][channel] -------------------------
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        os_t* dst = Os + (lane & 15) * NL + nt * 16 + (lane >> 4) * 4;
        if (sizeof(os_t) == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(acc[nt][0], acc[nt][1]), pack_bf16x2(acc[nt][2], acc[nt][3]));
        }
      }
      CLK(5)
      // ---------------- epilogue + store ---------------------------------------------------
      if (EPI == C3D_EPI_SWISH_SE_BWD) {
        const int n_tile = (int)((uint32_t)row0 / rps32);
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
          if (a.epi_gate && act_o) {
            const float* gp = a.epi_gate + (int64_t)cur_n * Np + v_o * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gp);
            const float4 g1 = *reinterpret_cast<const float4*>(gp + 4);
            eG[0] = g0.x; eG[1] = g0.y; eG[2] = g0.z; eG[3] = g0.w; eG[4] = g1.x; eG[5] = g1.y; eG[6] = g1.z; eG[7] = g1.w;
          }
          // a sample changes once per few hundred tiles: let its gate loads (and the flush atomics) land HERE, so that no
          // pending memory operation of this rare branch reaches the join below -- the compiler would otherwise guard the
          // first use of eG in EVERY tile with s_waitcnt vmcnt(0), which also drains the next tile's prefetch
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(eG[j]));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      // fully unrolled over the most passes this NT can need (RPo >= 64 / (2 NT) row-lanes per channel vector): a pass past
      // npass has no rows (row >= 16 for every lane), its load and store go nowhere.  As a runtime loop the companion-row
      // pipeline (e1n) was a loop-carried register holding a pending load: the compiler's copies of it at the back edge waited
      // for the load -- and, vmcnt retiring in order, for the prefetch burst behind it
      auto epi_pass = [&](const int p) {
        const int row = p * RPo + rr_o;
        const int m = row0 + row;
        typename RW::type e1c = e1n;
        // next pass's companion rows (or the first pass of the next tile of this iteration; else nowhere), unconditionally
        // (a pass past npass -- the unroll covers the most passes this NT can need -- asks for the next tile's first rows AGAIN:
        // with `p + 1 == npass` it overwrote them with the zeros of "nowhere", and the next tile of the iteration added no
        // companion rows to its first pass: output widths of 72 / 80 and 136..168 channels, two or more tiles per iteration)
        [[maybe_unused]] const uint32_t nxt_off = p + 1 < npass ? PW_E1_OFF(row0, p + 1) : (has_next ? PW_E1_OFF(row0 + 16, 0) : PW_OOB);
        if constexpr (E1_PIPE) e1n = BufIO<T>::load(rE1, nxt_off);
        typename RW::type x3c = x3n;
        typename RW::type c1c = c1n;
        if constexpr (X3F) { x3n = BufIO<T>::load(rX3, nxt_off); c1n = BufIO<T>::load(rC1, nxt_off); }
        if constexpr (sizeof(T) == 2 && (EPI == C3D_EPI_STORE || EPI == C3D_EPI_STATS)) {
          // bf16 plain-store / statistics epilogue: the staged tile already holds the stored bits -- copy the 16 bytes
          // as they are (the f32 round trip cost 8 unpack + 8 re-round + 4 pack VALU per vector for an identity)
          const bool ok = act_o && row < 16 && m < M32;
          uint4 raw = make_uint4(0u, 0u, 0u, 0u);
          if (ok) {
            raw = *reinterpret_cast<const uint4*>(Os + row * NL + v_o * 8);
            if (EPI == C3D_EPI_STATS) {
              float f[8];
              RW::cvt(raw, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) { s0[j] += f[j]; s1[j] += f[j] * f[j]; }
            }
          }
          if constexpr (sizeof(T) == 2) BufIO<bf16_t>::store_raw(rY, ok ? ((uint32_t)m * (uint32_t)Np + (uint32_t)v_o * 8u) * ES : PW_OOB, raw);
          return;
        }
        const bool ok = act_o && row < 16 && m < M32;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
          Vec8<os_t>::load(Os + row * NL + v_o * 8, f);
          if (EPI == C3D_EPI_STATS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float r = round_as<T>(f[j]);
              s0[j] += r; s1[j] += r * r;
            }
          } else if (EPI == C3D_EPI_SWISH_SE_BWD) {
            float bv[8], qs[8], eS[8], eB[8], eM[8], eR[8];
            RW::cvt(e1c, bv);
            lds_ld8(Ep + v_o * 8, eS); lds_ld8(Ep + Np + v_o * 8, eB);
            lds_ld8(Ep + 2 * Np + v_o * 8, eM); lds_ld8(Ep + 3 * Np + v_o * 8, eR);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float pb = fmaf(bv[j], eS[j], eB[j]);
              const float q = eG[j] * pb;
              const float sg = sigmoid_t<T>(q);
              const float dq = f[j] * sg * (1.f + q * (1.f - sg));
              const float t = round_as<T>(dq * eG[j]);
              s0[j] += dq * pb;                        // d gate
              s1[j] += t;                              // sum t1
              s2[j] += t * ((bv[j] - eM[j]) * eR[j]);  // sum t1*bhat (centred)
              f[j] = t;
              qs[j] = q * sg;                          // the forward operand of this convolution (WG: Q of the weight gradient)
            }
            if constexpr (WG == C3D_WG_SWISH) Vec8<os_t>::store(Os + row * NL + v_o * 8, qs);   // over the staged result vector this lane just read
          } else if (EPI == C3D_EPI_ADD) {
            if (a.res_mode == 0) {
              float rv[8];
              RW::cvt(e1c, rv);
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
          if constexpr (X3F) {
            if (WG == C3D_WG_MASKSUM || a.wg_mask_out) {   // g = dx * (y > 0) of the previous block's output y = wg_x3 (relu output: > 0 <=> nonzero bits)
              const uint32_t xw[4] = {x3c.x, x3c.y, x3c.z, x3c.w};
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] = ((xw[j >> 1] >> ((j & 1) * 16)) & 0x7fffu) != 0u && !((xw[j >> 1] >> ((j & 1) * 16)) & 0x8000u) ? f[j] : 0.f;
            }
            if (add_sums) {   // that block's BatchNorm_c-backward sums over g AS STORED (what c3d_block_out_bwd read back)
              float cv[8], eM[8], eR[8];
              RW::cvt(c1c, cv);
              lds_ld8(Ep + v_o * 8, eM); lds_ld8(Ep + Np + v_o * 8, eR);
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float gq = round_as<T>(f[j]);
                s0[j] += gq; s1[j] = fmaf(gq, (cv[j] - eM[j]) * eR[j], s1[j]);
              }
            }
          }
          if constexpr (WG == C3D_WG_ROWS) *reinterpret_cast<typename RW::type*>(Os + row * NL + v_o * 8) = x3c;
        }
        BufIO<T>::store(rY, ok ? ((uint32_t)m * (uint32_t)Np + (uint32_t)v_o * 8u) * ES : PW_OOB, f);   // outside the branch: no exec-masked memory operation in the loop
      };
#pragma unroll
      for (int p = 0; p < NPASS_MAX; ++p) epi_pass(p);
      if constexpr (WGRAD) {
        CLK(6)
        // ---------------- fused weight gradient ------------------------------------------------------
        // Os now holds Q (rows past M and the padding columns kept the zero accumulators of zero operand rows / zero weight
        // rows); a wave's LDS writes are ordered before its later LDS reads: no barrier.  With a second result buffer the
        // tiles of an iteration are taken in PAIRS (32 rows = one k-step of v_mfma_f32_16x16x32_bf16): half the ds_add_f64 per
        // row -- the LDS atomic rate (8 clocks per wave-instruction per CU) is what this step runs at.
        if (!wg_defer) {
          const int g4 = lane >> 4, li = lane & 15;
          const int poff = (4 * g4 + (li >> 2)) * KL + 4 * (li & 3), qoff = (4 * g4 + (li >> 2)) * NL + 4 * (li & 3);
          const lds_s16x4_ptr_t pp = (lds_s16x4_ptr_t)(xt + poff);
          const lds_s16x4_ptr_t qp = (lds_s16x4_ptr_t)(Os + qoff);
          const int ldw = L.ldw;
          double* dwl = dWs + (4 * g4) * ldw + li;
          // the smaller operand side is held as fragments, the other is walked by a real loop (fully unrolled, the compiler
          // keeps every zero-initialised MFMA result in flight: 200 registers, 800 B of scratch): conv_c (C3D_WG_SWISH) holds
          // its <= 3 P tiles (K = block output channels), conv_a (C3D_WG_ROWS) its <= NT Q tiles (N = block input channels)
          constexpr bool HOLD_P = WG == C3D_WG_SWISH;
          constexpr int NH = HOLD_P ? WG_NTP_MAX_SWISH : NT;
          const int n_hold = HOLD_P ? wg_ntp : wg_ntq, n_walk = HOLD_P ? wg_ntq : wg_ntp;
          const lds_s16x4_ptr_t hp = HOLD_P ? pp : qp, wp = HOLD_P ? qp : pp;
          // dW[p-channel tile][q-channel tile]: element (4 g4 + r, li) of tile (ip, iq) at dwl + ip*16*ldw + iq*16 + r*ldw
          const int hstep = HOLD_P ? 16 * ldw : 16, wstep = HOLD_P ? 16 : 16 * ldw;
          if (wg_prev_x) {          // pair: k = 8 rows per lane, 4 of the previous tile then 4 of this one (same map for P and Q)
            const lds_s16x4_ptr_t pp0 = (lds_s16x4_ptr_t)(wg_prev_x + poff);
            const lds_s16x4_ptr_t qp0 = (lds_s16x4_ptr_t)(wg_prev_q + qoff);
            const lds_s16x4_ptr_t hp0 = HOLD_P ? pp0 : qp0, wp0 = HOLD_P ? qp0 : pp0;
            typedef short s16x8_t __attribute__((ext_vector_type(8)));
            s16x8_t hf[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i)
              if (i < n_hold) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(hp0 + i * 4), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(hp + i * 4);
                hf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              }
#pragma unroll 1
            for (int j = 0; j < n_walk; ++j) {
              const s16x4_t wlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(wp0 + j * 4), whi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(wp + j * 4);
              const s16x8_t wf = __builtin_shufflevector(wlo, whi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
              for (int i = 0; i < NH; ++i) {
                if (i < n_hold) {
                  const bf16x8_t pa = __builtin_bit_cast(bf16x8_t, HOLD_P ? hf[i] : wf), qb = __builtin_bit_cast(bf16x8_t, HOLD_P ? wf : hf[i]);
                  const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, qb, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                  double* dst = dwl + i * hstep + j * wstep;
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    __hip_atomic_fetch_add(dst + r * ldw, (double)d[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          } else {                  // single tile (k = 16 rows)
            s16x4_t hf[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i)
              if (i < n_hold) hf[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(hp + i * 4);   // 16 channels = 4 chunks of 8 bytes on
#pragma unroll 1
            for (int j = 0; j < n_walk; ++j) {
              const s16x4_t wf = __builtin_amdgcn_ds_read_tr16_b64_v4i16(wp + j * 4);
#pragma unroll
              for (int i = 0; i < NH; ++i) {
                if (i < n_hold) {
                  const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(HOLD_P ? hf[i] : wf, HOLD_P ? wf : hf[i], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                  double* dst = dwl + i * hstep + j * wstep;
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    __hip_atomic_fetch_add(dst + r * ldw, (double)d[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __builtin_amdgcn_sched_barrier(0);   // one product in flight: 7 hoisted MFMAs + their f64 conversions spilled 160 B
                }
              }
            }
          }
        }
        CLK(14)
      }
    };

    // (no `break` out of this loop: the compiler materialised the exit's undefined next-`sub` from a register with a load
    // pending -- s_waitcnt vmcnt(0) in front of every tile, i.e. behind the prefetch burst just issued)
    const int nsub = t1 - it0 < L.tpi ? t1 - it0 : L.tpi;
    auto do_sub = [&](const int sub) {
      const int tile = it0 + sub;
      const int row0 = tile << 4;
      const lds_t* Xt = Xs + sub * 16 * KL;
      const bool two = MT == 2 && sub + 1 < L.tpi && tile + 1 < t1;          // wave-uniform
      // ---------------- MFMA: MT sub-tiles share every weight fragment read -----------------
      f32x4_t acc[NT], acc2[NT];   // acc2 is dead code (no registers) when MT == 1
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (MT == 2) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      if (WREG && KS <= 2) {
        const typename MM::frag_t xb0 = MM::load(Xt, lane & 15, 0, KL, lane);
        typename MM::frag_t xb1 = xb0;
        if (KS == 2) xb1 = MM::load(Xt, lane & 15, 1, KL, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = MM::mma(wr[WREG ? nt : 0], xb0, acc[nt]);
        if (KS == 2) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = MM::mma(wr[WREG ? NT + nt : 0], xb1, acc[nt]);
        }
      } else {
        // (bf16 only below: the chunk-major weight image, Mma<bf16_t>::widx)
        constexpr uint32_t wstr = 16 * 8 * 2;                                            // bytes between output tiles
        constexpr uint32_t wks = 4 * NT * 16 * 8 * 2;                                    // bytes between k steps of 32
        const uint32_t wlane = (uint32_t)(uintptr_t)Ws + (uint32_t)((((lane >> 4) * NT * 16 + (lane & 15)) * 8) * sizeof(lds_t));
        constexpr int UB = PwFragBatch<NT, PRO, EPI>::value;
        if (MT == 2 && two) {
          for (int ks = 0; ks < KS; ++ks) {
            const typename MM::frag_t xb = MM::load(Xt, lane & 15, ks, KL, lane);
            const typename MM::frag_t xb2 = MM::load(Xt + 16 * KL, lane & 15, ks, KL, lane);
            if constexpr (sizeof(T) == 2 && UB > 0) {
              const u32x4_t xv = __builtin_bit_cast(u32x4_t, xb), xv2 = __builtin_bit_cast(u32x4_t, xb2);
              MfmaSeq<NT, (UB > 0 ? UB : 1), 0, true>::run(acc, acc2, wlane + ks * wks, wstr, xv, xv2);
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const typename MM::frag_t wa = MM::loadw(Ws, nt * 16 + (lane & 15), ks, KL, NT * 16, lane);
                acc[nt] = MM::mma(wa, xb, acc[nt]);
                acc2[nt] = MM::mma(wa, xb2, acc2[nt]);
              }
            }
          }
        } else {
          for (int ks = 0; ks < KS; ++ks) {
            const typename MM::frag_t xb = MM::load(Xt, lane & 15, ks, KL, lane);
            if constexpr (sizeof(T) == 2 && UB > 0) {
              const u32x4_t xv = __builtin_bit_cast(u32x4_t, xb);
              MfmaSeq<NT, (UB > 0 ? UB : 1), 0, false>::run(acc, acc, wlane + ks * wks, wstr, xv, xv);
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const typename MM::frag_t wa = MM::loadw(Ws, nt * 16 + (lane & 15), ks, KL, NT * 16, lane);
                acc[nt] = MM::mma(wa, xb, acc[nt]);
              }
            }
          }
        }
      }
      CLK(4)
      if constexpr (WGRAD && MT == 1) {
        // weight-gradient pairs: the even tile of an iteration defers its step when the odd one follows in the same iteration
        const bool nxt = sub + 1 < L.tpi && tile + 1 < t1;
        const bool odd = (sub & 1) != 0;
        const bool pair = L.wg_pair != 0;
        finish_tile(acc, row0, nxt, Xt, odd ? OsB : OsA, (pair && odd) ? Xt - 16 * KL : nullptr, (pair && odd) ? OsA : nullptr,
                    pair && !odd && nxt);
      } else {
        finish_tile(acc, row0, MT == 2 ? two : (sub + 1 < L.tpi && tile + 1 < t1), Xt, OsA, nullptr, nullptr, false);
        if (MT == 2 && two) finish_tile(acc2, row0 + 16, sub + 2 < L.tpi && tile + 2 < t1, Xt + 16 * KL, OsA, nullptr, nullptr, false);
      }
      CLK(6)
    };
    if constexpr (E1_PIPE) {
      // The first tile of an iteration is its own copy of the code: its first companion row was requested BEFORE the prefetch
      // burst (17+ younger requests when it is consumed), a later tile's by the last pass of the tile before it (2 younger).
      // Sharing one loop body made the compiler wait for the stricter of the two at every tile: vmcnt(1..2) in front of the
      // first epilogue pass = the whole burst landed before the first store.
      do_sub(0);
      for (int sub = MT; sub < nsub; sub += MT) do_sub(sub);
    } else {
      for (int sub = 0; sub < nsub; sub += MT) do_sub(sub);
    }
  }
#undef PW_E1_OFF
#undef PW_ISSUE
#undef PW_ISSUE_DENSE
#undef PW_ISSUE_DENSE_SLOT
#undef PW_ISSUE_GENERIC
#undef PW_SLOT_OFF
  CLK(7)
  if constexpr (WGRAD) {
    // this workgroup's dW partial -> wg_ws[blockIdx.x][K][N] (the reducer launched behind this kernel adds the partials in
    // fixed order into wg_dw)
    __syncthreads();
    float* wsb = a.wg_ws + (size_t)blockIdx.x * a.K * a.N;
    for (int i = tid; i < a.K * a.N; i += WAVES * 64) {
      const int k = i / a.N, n = i - k * a.N;
      wsb[i] = (float)dWs[k * L.ldw + n];
    }
  }

  // ---- final flush of per-lane partial sums -------------------------------------------------
  // lanes -> LDS ([value][lane] per wave, the X regions are dead now) -> one thread per output sums the
  // RPo row-lanes of every wave -> ONE f64 atomic per value and workgroup.  (The lane step used to be
  // RPo dependent ds_bpermute shuffles per value: 7-14 us per launch on the narrow layers, Go = 3..7.)
  constexpr int NV = EPI == C3D_EPI_SWISH_SE_BWD ? 24 : 16;   // partial sums per lane
  const int flush_shuffle_max = L.flush_shuffle_max;
  if (EPI == C3D_EPI_STATS || EPI == C3D_EPI_SWISH_SE_BWD || add_sums) {
    // wide layers (2-3 row-lanes per channel vector): one or two shuffles per value beat the [NV][64] lane dump and
    // leave the cross-wave sum 8 reads per value instead of 16-24
    const bool dump = RPo > flush_shuffle_max && (size_t)L.wave_bytes >= (size_t)64 * NV * sizeof(float) + WAVES * sizeof(int);
    float* mine = reinterpret_cast<float*>(smem + L.wave_off + (size_t)wave * L.wave_bytes);   // [NV][64]
    int* ncur = reinterpret_cast<int*>(smem + L.wave_off + (size_t)WAVES * L.wave_bytes - WAVES * sizeof(int));
    __syncthreads();
    if (dump) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mine[j * 64 + lane] = s0[j];
        mine[(8 + j) * 64 + lane] = s1[j];
        if (EPI == C3D_EPI_SWISH_SE_BWD) mine[(16 + j) * 64 + lane] = s2[j];
      }
    } else {   // tiny LDS regions: shuffle the row-lanes together, result in lanes < Go, stored at [value][v_o]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float r0 = strided_lane_sum(s0[j], lane, Go, RPo);
        const float r1 = strided_lane_sum(s1[j], lane, Go, RPo);
        const float r2 = EPI == C3D_EPI_SWISH_SE_BWD ? strided_lane_sum(s2[j], lane, Go, RPo) : 0.f;
        if (lane < Go) {
          mine[j * Go + lane] = r0;
          mine[(8 + j) * Go + lane] = r1;
          if (EPI == C3D_EPI_SWISH_SE_BWD) mine[(16 + j) * Go + lane] = r2;
        }
      }
    }
    if (EPI == C3D_EPI_SWISH_SE_BWD && lane == 0) ncur[wave] = (int)cur_n;
    __syncthreads();
    // value (which, channel c = v*8 + j) of wave wv
    auto wave_value = [&](const int wv, const int which, const int c) -> float {
      const float* base = reinterpret_cast<const float*>(smem + L.wave_off + (size_t)wv * L.wave_bytes);
      const int v = c >> 3, j = c & 7;
      if (!dump) return base[(which * 8 + j) * Go + v];
      float acc = 0.f;
      for (int rr = 0; rr < RPo; ++rr) acc += base[(which * 8 + j) * 64 + rr * Go + v];
      return acc;
    };
    if (EPI == C3D_EPI_ADD) {
      // BatchNorm_c-backward sums of the previous block (single set, f64 [2][N]: the layout c3d_block_out_bwd fills)
      for (int i = tid; i < 2 * a.N; i += WAVES * 64) {
        const int which = i / a.N, c = i - which * a.N;
        float acc = 0.f;
        for (int wv = 0; wv < WAVES; ++wv) acc += wave_value(wv, which, c);
        atomicAdd(a.add_sums + which * a.N + c, (double)acc);
      }
    } else if (EPI == C3D_EPI_STATS) {
      // into one of C3D_STAT_STRIPES accumulator sets (keeps same-address atomic contention low)
      double* dst = a.stats + (size_t)(blockIdx.x % C3D_STAT_STRIPES) * 2 * a.N;
      for (int i = tid; i < 2 * a.N; i += WAVES * 64) {
        const int which = i / a.N, c = i - which * a.N;
        float acc = 0.f;
        for (int wv = 0; wv < WAVES; ++wv) acc += wave_value(wv, which, c);
        atomicAdd(dst + which * a.N + c, (double)acc);
      }
      if (a.fin.ticket) {   // last workgroup: BatchNorm scale/shift + running statistics (no separate finalize launch)
        int* flag = reinterpret_cast<int*>(smem + L.p_off);   // prologue-parameter region: dead after the tile loop
        if (c3dfin::last_workgroup(a.fin.ticket, gridDim.x, flag))
          c3dfin::bn_forward(a.fin, a.stats, C3D_STAT_STRIPES, a.N, a.Np, tid, WAVES * 64);
      }
    } else {
      // Per-(sample, channel) sums.  The waves of a workgroup almost always end inside the same sample:
      // combine them (the per-wave flush was 660 k same-address atomics = 40 % of the stage-3 kernel).
      int n_all = -1;
      bool uniform = true;
      for (int wv = 0; wv < WAVES; ++wv) {
        const int nw = ncur[wv];
        if (nw < 0) continue;          // a wave without tiles contributes nothing
        if (n_all < 0) n_all = nw;
        else if (nw != n_all) uniform = false;
      }
      if (uniform) {
        if (n_all >= 0) {
          for (int i = tid; i < 3 * Np; i += WAVES * 64) {
            const int which = i / Np, c = i - which * Np;
            float acc = 0.f;
            for (int wv = 0; wv < WAVES; ++wv)
              if (ncur[wv] >= 0) acc += wave_value(wv, which, c);
            atomicAdd(a.stats + ((int64_t)n_all * Np + c) * 3 + which, (double)acc);
          }
        }
      } else if (cur_n >= 0) {
        for (int i = lane; i < 3 * Np; i += 64) {
          const int which = i / Np, c = i - which * Np;
          atomicAdd(a.stats + ((int64_t)cur_n * Np + c) * 3 + which, (double)wave_value(wave, which, c));
        }
      }
    }
  }
  CLK(8)
  CLK_FLUSH
}

inline size_t al16(size_t v) { return (v + 15) / 16 * 16; }

template <typename T, int NT, int PRO, int EPI, int WAVES, int WG = 0>
bool plan_pw(const c3d_pw_args& a, PwLaunch& L, size_t& lds) {
  typedef Mma<T> MM;
  typedef typename OutStage<T, EPI>::type os_t;
  constexpr int PW_SLOTS = PwSlots<NT, PRO, EPI, WAVES, WG>::value;
  const int Kpad = (a.Kp + MM::KSTEP - 1) / MM::KSTEP * MM::KSTEP;
  const int KL = Kpad + MM::KPAD;
  const int NL = NT * 16 + (sizeof(os_t) == 4 ? 4 : 8);
  const int Q = ((a.Kp >> 3) + 3) >> 2;
  const size_t w_bytes = al16((size_t)NT * 16 * KL * sizeof(typename MM::lds_t));
  const size_t p_bytes = al16(((size_t)3 * a.Kp + (EPI == C3D_EPI_SWISH_SE_BWD ? (size_t)4 * a.Np : (WG == C3D_WG_ROWS || WG == C3D_WG_MASKSUM) ? (size_t)2 * a.Np : 0)) * sizeof(float));   // prologue [3][Kp] | epilogue [4][Np] / [2][Np]
  const size_t os_bytes = al16((size_t)16 * NL * sizeof(os_t));
  const size_t gs_bytes = PRO == C3D_PRO_BN_SE_SWISH ? al16((size_t)a.Kp * sizeof(float)) : 0;   // per-wave gate copy
  // fused weight gradient: workgroup-shared f64 accumulators [ceil(Kp/16)*16][ceil(Np/16)*16 + 4] (row stride = 4 mod 8
  // doubles: the four 16-lane groups of a ds_add_f64 -- rows 4g + r -- fall on the two halves of the 64 banks alternately)
  const int ldw = ((a.Np + 15) / 16) * 16 + 4;
  const size_t dw_bytes = (WG == C3D_WG_SWISH || WG == C3D_WG_ROWS) ? al16((size_t)((a.Kp + 15) / 16) * 16 * ldw * sizeof(double)) : 0;
  const size_t se_bytes = (PRO == C3D_PRO_BN_SE_SWISH && a.se_w1) ? al16((size_t)PW_SE_NS * (a.Kp + PW_SE_CR) * sizeof(float)) : 0;
  for (int tpi = PW_SLOTS / Q; tpi >= 1; --tpi) {
    const size_t xs_bytes = al16((size_t)tpi * 16 * KL * sizeof(typename MM::lds_t));
    size_t wave_bytes = xs_bytes + os_bytes + gs_bytes;
    // fused weight gradient: a second result-tile buffer per wave (tile pairs) when the iteration has >= 2 tiles and it fits
    const bool pair = (WG == C3D_WG_SWISH || WG == C3D_WG_ROWS) && tpi >= 2 && w_bytes + p_bytes + WAVES * (wave_bytes + os_bytes) + dw_bytes + se_bytes <= 160 * 1024;
    L.wg_pair = pair ? 1 : 0;
    L.os_off2 = (int)(xs_bytes + os_bytes + gs_bytes);
    if (pair) wave_bytes += os_bytes;
    const size_t total = w_bytes + p_bytes + WAVES * wave_bytes + dw_bytes + se_bytes;
    if (total <= 160 * 1024) {
      L.tpi = tpi; L.xs_rows = tpi * 16;
      L.w_off = 0; L.p_off = (int)w_bytes; L.wave_off = (int)(w_bytes + p_bytes);
      L.wave_bytes = (int)wave_bytes; L.os_off = (int)xs_bytes; L.gs_off = (int)(xs_bytes + os_bytes);
      L.dw_off = (int)(w_bytes + p_bytes + WAVES * wave_bytes); L.ldw = ldw;
      L.se_off = (int)(w_bytes + p_bytes + WAVES * wave_bytes + dw_bytes);
      lds = total;
      return true;
    }
  }
  return false;
}

constexpr int PW_WG_MAX_PARTS = 512;   // workgroups (= dW partials) of a launch with the fused weight gradient

template <typename T, int NT, int PRO, int EPI, int WAVES, bool DENSE, int WG = 0>
int launch_pw_d(const c3d_pw_args& a, hipStream_t stream) {
  PwLaunch L;
  size_t lds = 0;
  if (!plan_pw<T, NT, PRO, EPI, WAVES, WG>(a, L, lds)) return C3D_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<T, NT, PRO, EPI, WAVES, DENSE, WG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int64_t tiles = (a.M + 15) >> 4;
  int occ = (int)((160 * 1024) / lds);
  if (occ > 32 / WAVES) occ = 32 / WAVES;
  if (occ < 1) occ = 1;
  int64_t max_blocks = (int64_t)device_cus() * occ;
  if ((WG == C3D_WG_SWISH || WG == C3D_WG_ROWS) && max_blocks > PW_WG_MAX_PARTS) max_blocks = PW_WG_MAX_PARTS;
  int64_t blocks = (tiles + (int64_t)WAVES * L.tpi - 1) / ((int64_t)WAVES * L.tpi);  // >= one iteration per wave
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  int64_t tpw = (tiles + blocks * WAVES - 1) / (blocks * WAVES);
  // Whole iterations only when that does not idle CUs: 6144 tiles over 2048 waves is 3 per wave; rounding
  // to 4 (tpi = 2) left 64 of 256 CUs without a workgroup (stage-3 K=96 layers, -20 % per launch).
  static const int round_iters = c3d_env("C3D_PW_ROUND") ? atoi(c3d_env("C3D_PW_ROUND")) : 0;
  const int64_t tpw_r = (tpw + L.tpi - 1) / L.tpi * L.tpi;
  const int64_t blocks_r = (tiles + tpw_r * WAVES - 1) / (tpw_r * WAVES);
  if (round_iters || blocks_r * 16 >= blocks * 15) tpw = tpw_r;
  blocks = (tiles + tpw * WAVES - 1) / (tpw * WAVES);
  L.tiles_per_wave = (int)tpw;
  if (PRO == C3D_PRO_BN_SE_SWISH && a.se_w1) {   // consumer-side SE gate: a workgroup's rows may span at most PW_SE_NS samples
    const int64_t rows_wg = tpw * WAVES * 16;
    if (a.rows_per_sample <= 0 || (rows_wg + a.rows_per_sample - 2) / a.rows_per_sample + 1 > PW_SE_NS || a.se_cr > PW_SE_CR)
      return PW_E_SE_FALLBACK;
  }
  static const int fsm = c3d_env("C3D_PW_FLUSH_SHFL") ? atoi(c3d_env("C3D_PW_FLUSH_SHFL")) : 0;   // tuning knob (measured: no gain)
  L.flush_shuffle_max = fsm;
  pw_gemm_kernel<T, NT, PRO, EPI, WAVES, DENSE, WG><<<dim3((unsigned)blocks), dim3(WAVES * 64), lds, stream>>>(a, L);
  C3D_CHECK_LAUNCH();
  if (WG == C3D_WG_SWISH || WG == C3D_WG_ROWS)   // partials [blocks][K][N] -> dw[n*w_sn + k*w_sk] (+=), fixed order
    return c3d_detail_pw_wgrad_reduce(a.wg_ws, a.wg_dw, a.K, a.N, (int)blocks, a.w_sk, a.w_sn, stream);
  return 0;
}

template <typename T, int NT, int PRO, int EPI, int WAVES>
int launch_pw_w(const c3d_pw_args& a, hipStream_t stream) {
  // the dense-row specialisation exists for the throughput (bf16) path only; f32 (parity) uses the generic code
  if (a.row_mode == C3D_ROWS_DENSE) return launch_pw_d<T, NT, PRO, EPI, WAVES, true>(a, stream);   // (f32 too since round 5: the straight-line buffer-addressed loop)
  return launch_pw_d<T, NT, PRO, EPI, WAVES, false>(a, stream);
}

template <typename T, int NT, int PRO, int EPI>
int launch_pw(const c3d_pw_args& a, hipStream_t stream) {
  // prefer 8 waves per workgroup (one weight copy per 8 waves) when LDS allows it
  PwLaunch L;
  size_t lds = 0;
  static const int force8 = c3d_env("C3D_PW_FORCE8") ? atoi(c3d_env("C3D_PW_FORCE8")) : 0;  // tuning knob
  const bool e1_epi = EPI == C3D_EPI_SWISH_SE_BWD || EPI == C3D_EPI_ADD;  // epilogues with exposed companion loads
  if (force8 >= 0 && plan_pw<T, NT, PRO, EPI, 8>(a, L, lds) &&
      (L.tpi * ((((a.Kp >> 3) + 3) >> 2)) >= 4 || lds <= 80 * 1024 || force8 == 2 || (force8 == 1 && e1_epi)))
    return launch_pw_w<T, NT, PRO, EPI, 8>(a, stream);
  if (plan_pw<T, NT, PRO, EPI, 4>(a, L, lds)) return launch_pw_w<T, NT, PRO, EPI, 4>(a, stream);
  if (sizeof(T) == 4) return launch_pw_w<float, NT, PRO, EPI, 2>(a, stream);  // f32 parity path only
  return C3D_E_UNSUPPORTED;
}

template <typename T, int PRO, int EPI>
int dispatch_nt(const c3d_pw_args& a, hipStream_t stream) {
  const int nt = (a.Np + 15) / 16;
  if (nt <= 2) return launch_pw<T, 2, PRO, EPI>(a, stream);
  if (nt <= 4) return launch_pw<T, 4, PRO, EPI>(a, stream);
  if (nt <= 7) return launch_pw<T, 7, PRO, EPI>(a, stream);
  if (nt <= 14) return launch_pw<T, 14, PRO, EPI>(a, stream);
  return C3D_E_UNSUPPORTED;
}

template <typename T>
int dispatch_mode(const c3d_pw_args& a, hipStream_t s) {
  const int pro = a.pro_mode, epi = a.epi_mode;
  if (a.wg_mode != C3D_WG_NONE) return C3D_E_UNSUPPORTED;   // routed to pw_gemm_wg.hip by the entry point
  if (pro == C3D_PRO_NONE && epi == C3D_EPI_STORE) return dispatch_nt<T, C3D_PRO_NONE, C3D_EPI_STORE>(a, s);
  if (pro == C3D_PRO_NONE && epi == C3D_EPI_STATS) return dispatch_nt<T, C3D_PRO_NONE, C3D_EPI_STATS>(a, s);
  if (pro == C3D_PRO_BN_SE_SWISH && epi == C3D_EPI_STORE) return dispatch_nt<T, C3D_PRO_BN_SE_SWISH, C3D_EPI_STORE>(a, s);
  if (pro == C3D_PRO_BN_SE_SWISH && epi == C3D_EPI_STATS) return dispatch_nt<T, C3D_PRO_BN_SE_SWISH, C3D_EPI_STATS>(a, s);
  if (pro == C3D_PRO_AFFINE2 && epi == C3D_EPI_STORE) return dispatch_nt<T, C3D_PRO_AFFINE2, C3D_EPI_STORE>(a, s);
  if (pro == C3D_PRO_AFFINE2 && epi == C3D_EPI_STATS) return dispatch_nt<T, C3D_PRO_AFFINE2, C3D_EPI_STATS>(a, s);
  if (pro == C3D_PRO_AFFINE2 && epi == C3D_EPI_SWISH_SE_BWD) return dispatch_nt<T, C3D_PRO_AFFINE2, C3D_EPI_SWISH_SE_BWD>(a, s);
  if (pro == C3D_PRO_AFFINE2 && epi == C3D_EPI_ADD) return dispatch_nt<T, C3D_PRO_AFFINE2, C3D_EPI_ADD>(a, s);
  if (pro == C3D_PRO_NONE && epi == C3D_EPI_ADD) return dispatch_nt<T, C3D_PRO_NONE, C3D_EPI_ADD>(a, s);
  return C3D_E_UNSUPPORTED;
}

}  // namespace
