// VALU issue-rate probe (gfx950): v_fma_f32 vs v_pk_fma_f32 vs v_dot2c_f32_bf16, 16 independent chains; modes 16 / 17: a whole
// swish(x) = x * sigmoid(x) per chain element -- v_mul + v_exp + v_add + v_rcp + v_mul against a clamped odd polynomial of the
// same 2^-10 accuracy (v_med3 + 9 full-rate FMA / MUL): the round-4 review's "packed-FMA sigmoid" priced in instructions.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int NCH = 16, ITERS = 4096;

template <int MODE> __global__ __launch_bounds__(256) void probe(float* out, uint32_t seed) {
  float acc[NCH];
  f32x2_t acc2[NCH];
  const uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
  const float fa = __uint_as_float((a & 0x7fffff) | 0x3f800000), fb = __uint_as_float((b & 0x7fffff) | 0x3c800000);
#pragma unroll
  for (int i = 0; i < NCH; ++i) { acc[i] = (float)i; acc2[i] = f32x2_t{(float)i, 1.f}; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "v"(f32x2_t{fa, fb}), "v"(f32x2_t{fb, fa}));
      if (MODE == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(acc[i]) : "v"(a));
      if (MODE == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 5) asm volatile("v_rcp_f32 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 6) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 7) asm volatile("v_fract_f32 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 8) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
      if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc2[i]) : "v"(f32x2_t{fa, fb}));
      if (MODE == 10) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(fa));
      if (MODE == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(fa), "v"(fb));
      if (MODE == 12) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(acc[i]) : "v"(a));
      if (MODE == 13) asm volatile("v_exp_f16 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 14) asm volatile("v_rcp_f16 %0, %1" : "=v"(acc[i]) : "v"(fa));
      if (MODE == 15) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
      if (MODE == 16) {   // x * rcp(1 + exp2(-log2e x))
        float t, x = acc[i];
        asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\tv_mul_f32 %1, %1, %0" : "=&v"(t), "+v"(x));
        acc[i] = x + fa;
      }
      if (MODE == 17) {   // x * (0.5 + c P(c^2)), c = clamp(x, -7, 7), P of degree 6 in c^2 (7 odd terms): |error| < 1e-3
        float c, s2, p, x = acc[i];
        asm volatile("v_med3_f32 %0, %3, %5, %6\n\tv_mul_f32 %1, %0, %0\n\t"
                     "v_mov_b32 %2, 0x2f800000\n\tv_fma_f32 %2, %2, %1, %4\n\tv_fma_f32 %2, %2, %1, %4\n\tv_fma_f32 %2, %2, %1, %4\n\t"
                     "v_fma_f32 %2, %2, %1, %4\n\tv_fma_f32 %2, %2, %1, %4\n\tv_fma_f32 %2, %2, %1, %4\n\t"
                     "v_fma_f32 %2, %2, %0, 0.5\n\tv_mul_f32 %3, %3, %2"
                     : "=&v"(c), "=&v"(s2), "=&v"(p), "+v"(x) : "v"(fb), "v"(-7.0f), "v"(7.0f));
        acc[i] = x + fa;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += acc[i] + acc2[i][0] + acc2[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* out, int waves_per_simd) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  probe<MODE><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, 2);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)ITERS * NCH * waves_per_simd;
  printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", out, w);
    run<1>("v_pk_fma_f32", out, w);
    run<2>("v_dot2c_f32_bf16", out, w);
    run<3>("v_lshlrev_b32", out, w);
    run<4>("v_exp_f32", out, w);
    run<5>("v_rcp_f32", out, w);
    run<6>("v_cvt_u32_f32", out, w);
    run<7>("v_fract_f32", out, w);
    run<8>("v_med3_f32", out, w);
    run<9>("v_pk_mul_f32", out, w);
    run<10>("v_mul_f32", out, w);
    run<11>("v_cvt_pk_bf16_f32", out, w);
    run<12>("v_and_b32 literal", out, w);
    run<13>("v_exp_f16", out, w);
    run<14>("v_rcp_f16", out, w);
    run<15>("v_pk_fma_f16", out, w);
    run<16>("swish exp+rcp (5 instr)", out, w);
    run<17>("swish poly7 (11 instr)", out, w);
  }
  return 0;
}
