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
// The maximum of the 16 magnitudes is a tree of v_max3_f32 with |.| source modifiers, 8 instructions (left to the compiler: 15, it
// canonicalises every fmaxf operand) in ONE asm statement (between dependent asm statements the compiler pads with s_nop).
// `after` = any value the caller has ALREADY derived from v by an ordinary instruction (e.g. the bf16 pack of v[0], v[1]): the asm lists
// it as an input, so it is scheduled behind that instruction and inherits the wait states the compiler put in front of it -- the compiler
// does not pad an asm statement that reads a register an MFMA has just written (an identity stage feeds its accumulators straight in:
// the maximum was taken of registers still in flight and d feats came out with the wrong exponent).
template <class V>
__device__ __forceinline__ uint32_t mx8_exponent(const V& v, uint32_t after) {
  float m, t1, t2;
  asm("v_max3_f32 %0, |%3|, |%4|, |%5|\n\t"
      "v_max3_f32 %1, |%6|, |%7|, |%8|\n\t"
      "v_max3_f32 %2, |%9|, |%10|, |%11|\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_max3_f32 %1, |%12|, |%13|, |%14|\n\t"
      "v_max3_f32 %2, |%15|, |%16|, |%17|\n\t"
      "v_max3_f32 %1, %1, %2, |%18|\n\t"
      "v_max_f32 %0, %0, %1"
      : "=&v"(m), "=&v"(t1), "=&v"(t2)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
        "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(after));
  m = __builtin_fmaf(m, 0.0078125f, m);
  uint32_t e = __builtin_bit_cast(uint32_t, m) >> 23;
  return e < 6u ? 6u : (e > 254u ? 254u : e);
}
// u = RNE(v / 2^(E-133)) + 128 by ONE rounding: v * 2^(133-E) + (1.5 * 2^23 + 128) has ulp 1, so the fma rounds to the integer and the
// low mantissa byte of the sum is u (|v| 2^(133-E) <= 127.008 by the choice of E: 1 <= u <= 255, no saturation needed).  Two values per
// v_pk_fma_f32, then three v_perm_b32 per four bytes (bytes4): 20 instructions per 16 values (fma + v_cvt_pk_u8_f32 per value: 32).
// (The elements of the packed result are copied to floats before the bit cast: __builtin_bit_cast(uint32_t, r[1]) on a vector ELEMENT
// reads the first four bytes of the vector, i.e. r[0], with hipcc 7.2 / clang 22 -- tools/probe_perm_fold.hip.)
template <class V>
__device__ __forceinline__ uint4 mx8_encode(const V& v, uint32_t e) {
  const float inv = __builtin_bit_cast(float, (260u - e) << 23);  // 2^(133 - E)
  const f32x2 inv2 = {inv, inv}, magic = {12583040.0f, 12583040.0f};
  uint32_t t[16];
#pragma unroll
  for (int g = 0; g < 16; g += 2) {
    const f32x2 pair = {v[g], v[g + 1]};
    const f32x2 r = __builtin_elementwise_fma(pair, inv2, magic);
    const float r0 = r[0], r1 = r[1];
    t[g] = __builtin_bit_cast(uint32_t, r0), t[g + 1] = __builtin_bit_cast(uint32_t, r1);
  }
  return make_uint4(bytes4(t[0], t[1], t[2], t[3]), bytes4(t[4], t[5], t[6], t[7]), bytes4(t[8], t[9], t[10], t[11]),
                    bytes4(t[12], t[13], t[14], t[15]));
}
__device__ __forceinline__ float mx8_scale(uint32_t e) { return __builtin_bit_cast(float, (e - 6u) << 23); }  // 2^(E - 133)
__device__ __forceinline__ float mx8_value(uint32_t w, int k, float s, float bias) { return __builtin_fmaf(ubyte_f32(w, k), s, bias); }  // bias = -128 s

}  // namespace sr
