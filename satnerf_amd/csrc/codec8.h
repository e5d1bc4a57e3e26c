// 8-bit codecs of the training workspaces (mlp_layout.h, SR_FMT8): PHASE8 for sin stages, MX8 (int8 with a shared
// power-of-two scale per lane per 16 values) for identity stages and pre-activation gradients.
#pragma once
#include "common.h"

namespace sr {

// ---- PHASE8 -----------------------------------------------------------------------------------------------------------
// x = pre-activation in revolutions (|x| < 2^13).  x + 1.5*2^15 has ulp 2^-8, so the low mantissa byte of the sum is
// round-to-nearest-even(x * 256) mod 256 -- one v_add per value, no fract / scale / convert.
__device__ __forceinline__ uint32_t phase8_bits(float x) { return __builtin_bit_cast(uint32_t, x + 49152.0f); }
__device__ __forceinline__ uint32_t bytes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {  // low bytes of a, b, c, d -> one dword
  const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u);  // [a.b0, b.b0, 0, 0]
  const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
  return __builtin_amdgcn_perm(cd, ab, 0x05040100u);             // [ab.b0, ab.b1, cd.b0, cd.b1]
}
template <class V>
__device__ __forceinline__ uint4 phase8_encode(const V& x) {  // x[0..15]
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = bytes4(phase8_bits(x[4 * q]), phase8_bits(x[4 * q + 1]), phase8_bits(x[4 * q + 2]), phase8_bits(x[4 * q + 3]));
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ float ubyte_f32(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }  // v_cvt_f32_ubyte<k>
// phase byte k of w as a number of revolutions for v_sin / v_cos -- 128 + u / 256, built by ONE v_perm_b32: the byte lands in bits 8..15 of
// 0x43000000 (= 128.0f, whose mantissa lsb is 2^-16).  sin and cos have period 1 in revolutions and the hardware reduces |x| <= 256
// exactly, so this is sin / cos of u / 256 without the convert + scale pair (one instruction per value less in the dX and dW decoders).
__device__ __forceinline__ float phase8_rev(uint32_t w, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_perm(0x43000000u, w, 0x070c000cu | ((uint32_t)k << 8)));
}

// ---- MX8 --------------------------------------------------------------------------------------------------------------
// E = biased exponent of max|v| * (1 + 2^-7) (so that max|v| / 2^(E-133) <= 127.008 rounds to <= 127), clamped to >= 6.
template <class V>
__device__ __forceinline__ uint32_t mx8_exponent(const V& v) {
  float m = 0.f;
#pragma unroll
  for (int g = 0; g < 16; g += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[g]), __builtin_fabsf(v[g + 1])));
  m = __builtin_fmaf(m, 0.0078125f, m);
  uint32_t e = __builtin_bit_cast(uint32_t, m) >> 23;
  return e < 6u ? 6u : (e > 254u ? 254u : e);
}
template <class V>
__device__ __forceinline__ uint4 mx8_encode(const V& v, uint32_t e) {
  const float inv = __builtin_bit_cast(float, (260u - e) << 23);  // 2^(133 - E)
  uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int g = 0; g < 16; ++g)  // v_cvt_pk_u8_f32 rounds to nearest even and saturates (tools/probe_cvt.hip)
    w[g >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[g], inv, 128.0f), g & 3, w[g >> 2]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ float mx8_scale(uint32_t e) { return __builtin_bit_cast(float, (e - 6u) << 23); }  // 2^(E - 133)
__device__ __forceinline__ float mx8_value(uint32_t w, int k, float s, float bias) { return __builtin_fmaf(ubyte_f32(w, k), s, bias); }  // bias = -128 s

}  // namespace sr
