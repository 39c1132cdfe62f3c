// What a cooperative (persistent) per-block kernel would pay per phase boundary vs what the launch sequence pays today:
//   (a) grid barrier of 256 workgroups x 512 threads, flat (one device-scope counter) and XCD-hierarchical (8 XCD-local
//       counters + 1 global), in a persistent kernel;
//   (b) the boundary between two dependent kernel launches on one stream (empty kernels and kernels that end with a
//       device-scope atomic flush, as every statistics producer does).
// Build: hipcc --offload-arch=gfx950 -O2 grid_barrier_cost.hip -o grid_barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void flat_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * nblocks) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
// workgroup b runs on XCD b % 8 (round-robin dispatch): 8 local counters (one cache line each) + one global
__device__ __forceinline__ void xcd_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    const unsigned x = blockIdx.x & 7u, per = nblocks / 8u;
    unsigned* loc = ctr + 64 + x * 64;
    const unsigned prev = __hip_atomic_fetch_add(loc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == epoch * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // last of this XCD
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8u) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(512) void persistent(unsigned* ctr, int iters, float* sink) {
  unsigned epoch = 0;
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    v = v * 1.0001f + 1.f;
    if (MODE == 0) flat_barrier(ctr, gridDim.x, epoch); else xcd_barrier(ctr, gridDim.x, epoch);
  }
  if (v == 12345.f) sink[0] = v;
}
__global__ __launch_bounds__(512) void empty_kernel(float* sink) { if (sink[0] == 12345.f) sink[1] = 1.f; }
__global__ __launch_bounds__(512) void flush_kernel(double* acc, float* sink) {
  __shared__ float s[512];
  s[threadIdx.x] = sink[threadIdx.x & 7];
  __syncthreads();
  if (threadIdx.x < 96) atomicAdd(acc + (blockIdx.x % 16) * 96 + threadIdx.x, (double)s[threadIdx.x]);
}
int main() {
  unsigned* ctr; float* sink; double* acc;
  CHECK(hipMalloc(&ctr, 4096 * 4)); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&acc, 16 * 96 * 8));
  CHECK(hipMemset(sink, 0, 4096)); CHECK(hipMemset(acc, 0, 16 * 96 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(ctr, 0, 4096 * 4));
      CHECK(hipEventRecord(e0));
      if (mode == 0) persistent<0><<<256, 512>>>(ctr, iters, sink); else persistent<1><<<256, 512>>>(ctr, iters, sink);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("grid barrier, %s, 256 workgroups x 512 threads: %.2f us per barrier\n", mode ? "XCD-hierarchical (8 + 1 counters)" : "flat (one counter)", ms * 1e3 / iters);
    }
  }
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) { if (kind == 0) empty_kernel<<<256, 512>>>(sink); else flush_kernel<<<256, 512>>>(acc, sink); }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("dependent launches on one stream, %s: %.2f us per launch\n", kind ? "256 x 512 kernel ending in 96 f64 atomics per workgroup (16 stripes)" : "empty 256 x 512 kernel", ms * 1e3 / iters);
    }
  }
  return 0;
}
