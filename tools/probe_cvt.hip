// Probe: rounding / saturation of v_cvt_pk_u8_f32 and the byte of the magic-add phase encoder on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const float* in, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i < n) {
    out[2 * i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
    out[2 * i + 1] = __builtin_bit_cast(uint32_t, in[i] + 49152.0f) & 0xffu;
  }
}
int main() {
  const float h[] = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 255.4f, 255.6f, 256.f, 300.f, -0.4f, -1.f, 127.999f, 0.00195f, 0.998f, -0.002f, 3.25f, -3.25f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; uint32_t* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 8);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, n);
  uint32_t r[2 * 64]; hipMemcpy(r, o, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("x=%10.5f  cvt_pk_u8=%3u   phase8(x)=%3u (x*256 mod 256 = %.2f)\n", h[i], r[2 * i], r[2 * i + 1], (h[i] - floorf(h[i])) * 256.f);
  return 0;
}
