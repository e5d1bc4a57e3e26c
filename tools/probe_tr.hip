// Probe: lane/element semantics of ds_read_b64_tr_b16 on gfx950 (prints, for every lane, which LDS halfwords it got).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int lane = threadIdx.x;
  // every lane points at its own 8-byte chunk: lane L -> halfwords [4L, 4L+4)
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + (mode == 0 ? lane * 8 : (lane & 15) * 8 + (lane >> 4) * 512);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  out[lane * 2] = (uint32_t)v;
  out[lane * 2 + 1] = (uint32_t)(v >> 32);
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 8);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane: 4 halfword indices)\n", mode);
    for (int l = 0; l < 64; ++l) printf("L%02d: %4u %4u %4u %4u\n", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16);
  }
  return 0;
}
