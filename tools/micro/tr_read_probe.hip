// Probe of gfx950's LDS transpose read (ds_read_b64_tr_b16) and of v_mfma_f32_16x16x16_bf16 operand layouts:
// prints, for an LDS image img[row][col] = row * 256 + col (u16), what lane l receives when each lane of a 16-lane
// group g addresses the 8-byte chunk (row 4g + (l%16)/4, cols 4*(l%4) .. +3).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint16_t img[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) img[i] = 0xffff;
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * 16; i += 64) { const int r = i / 16, c = i % 16; img[r * stride + c] = (uint16_t)(r * 256 + c); }
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const uint32_t addr = (uint32_t)(uintptr_t)(img + (4 * g + (i >> 2)) * stride + 4 * (i & 3));
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 40, 72}) {
    probe<<<1, 64>>>(d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { const int want = (4 * (l >> 4) + j) * 256 + (l & 15); if (h[l * 4 + j] != want) ok = 0; }
    printf("stride %d: lane l elem j == img[4*(l/16)+j][l%%16] ? %s\n", stride, ok ? "YES" : "NO");
    if (!ok) for (int l = 0; l < 64; l += 1) printf("  lane %2d: %04x %04x %04x %04x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
