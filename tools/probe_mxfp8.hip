// Probe of the gfx950 MX conversions used for the dpre workspace (hipcc --offload-arch=gfx950 tools/probe_mxfp8.hip -o /tmp/probe_mxfp8):
// v_cvt_scalef32_pk_fp8_f32 (two floats -> two e4m3 bytes, divided by the scale) and v_cvt_scalef32_pk_bf16_fp8 / _f32_fp8 (back, times the scale).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float scale, unsigned* enc, float* dec, unsigned* decbf) {
  const int i = threadIdx.x;
  const float a = in[2 * i], b = in[2 * i + 1];
  s16x2 w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, a, b, scale, false);  // low half
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, b, a, scale, true);   // high half
  const unsigned u = __builtin_bit_cast(unsigned, w);
  enc[i] = u;
  const f32x2 lo = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(u, scale, false), hi = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(u, scale, true);
  dec[4 * i] = lo.x, dec[4 * i + 1] = lo.y, dec[4 * i + 2] = hi.x, dec[4 * i + 3] = hi.y;
  const bf16x2 bl = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(u, scale, false);
  decbf[i] = __builtin_bit_cast(unsigned, bl);
}
int main() {
  const int n = 16;
  float h[2 * n] = {1.0f, -1.0f, 0.5f, 3.0f, 448.0f, 500.0f, 1000.0f, -2000.0f, 0.001f, 0.01f, 1.0625f, 1.1875f, 17.0f, 19.0f, 0.0f, -0.0f,
                    1.5f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f, 100.0f, 200.0f, 300.0f, 400.0f, 0.015625f, 0.001953125f, 7.0f, 9.0f, 11.0f, 13.0f};
  float *din, *ddec; unsigned *denc, *dbf;
  hipMalloc(&din, sizeof(h)); hipMalloc(&denc, n * 4); hipMalloc(&ddec, n * 16); hipMalloc(&dbf, n * 4);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, din, scale, denc, ddec, dbf);
    unsigned e[n], bf[n]; float d[4 * n];
    hipMemcpy(e, denc, sizeof(e), hipMemcpyDeviceToHost); hipMemcpy(d, ddec, sizeof(d), hipMemcpyDeviceToHost); hipMemcpy(bf, dbf, sizeof(bf), hipMemcpyDeviceToHost);
    printf("scale %g\n", scale);
    for (int i = 0; i < n; ++i) {
      unsigned lo = bf[i] << 16, hi = bf[i] & 0xffff0000u; float fl, fh; memcpy(&fl, &lo, 4); memcpy(&fh, &hi, 4);
      printf("  a=%-10g b=%-10g enc=%08x  dec: %g %g | %g %g   bf16(lo half): %g %g\n", h[2 * i], h[2 * i + 1], e[i], d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3], fl, fh);
    }
  }
  return 0;
}
