// LDS atomic issue rates on gfx950: clocks per wave-instruction of ds_add_f32 / ds_add_u32 / ds_write_b32 / ds_add_rtn_f32,
// 8 waves per workgroup, (a) every wave its own 64-float row, (b) all waves the same row, (c) the fused-wgrad pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int SHARED>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  __shared__ float buf[8 * 4096];
  for (int i = threadIdx.x; i < 8 * 4096; i += 512) buf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* p = buf + (SHARED ? 0 : wave * 4096) + (SHARED == 2 ? ((lane >> 4) * 4) * 116 + (lane & 15) : lane * ((MODE == 4 || MODE == 5) ? 2 : 1));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float* q = p + (SHARED == 2 ? (u & 3) * 116 + (u >> 2) * 16 : u * 64 * ((MODE == 4 || MODE == 5) ? 2 : 1));
      if (MODE == 0) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 1) __hip_atomic_fetch_add((unsigned*)q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) *(volatile float*)q = 1.0f;
      else if (MODE == 4) __hip_atomic_fetch_add((unsigned long long*)q + 0, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 5) __hip_atomic_fetch_add((double*)q, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 6) { float r = *(volatile float*)q; *(volatile float*)q = r + 1.0f; }
      else { float r = __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" :: "v"(r)); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int SHARED> void run(const char* name, unsigned long long* d) {
  const int iters = 200;
  k<MODE, SHARED><<<1, 512>>>(d, iters);
  k<MODE, SHARED><<<1, 512>>>(d, iters);
  unsigned long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0; for (int i = 0; i < 8; ++i) mx = h[i] > mx ? h[i] : mx;
  // s_memtime ticks at 100 MHz on this part: convert with the ~2.1 GHz shader clock
  printf("%-34s %8.2f memtime-ticks per wave-instruction (8 waves)  ~%7.1f shader clk per instruction per CU\n", name, mx / (iters * 16.0), mx / (iters * 16.0) * 21.0 / 8.0);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  run<2, 0>("ds_write_b32 private rows", d);
  run<1, 0>("ds_add_u32 private rows", d);
  run<0, 0>("ds_add_f32 private rows", d);
  run<0, 1>("ds_add_f32 shared row", d);
  run<0, 2>("ds_add_f32 shared, wgrad pattern", d);
  run<3, 0>("ds_add_rtn_f32 private rows", d);
  run<1, 1>("ds_add_u32 shared row", d);
  run<4, 0>("ds_add_u64 private rows", d);
  run<4, 1>("ds_add_u64 shared row", d);
  run<5, 0>("ds_add_f64 private rows", d);
  run<6, 0>("ds_read+ds_write f32 private", d);
  return 0;
}
