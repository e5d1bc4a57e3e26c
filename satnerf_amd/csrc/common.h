// Shared device/host helpers for libsatrender (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/satrender.h"

namespace sr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// hipFuncSetAttribute(max dynamic LDS) once per (kernel, device) of this process: false + sr_last_error on failure
bool ensure_dynamic_lds(const void* kernel, size_t bytes);

#define SR_REQUIRE(cond, ...)  \
  do {                         \
    if (!(cond)) {             \
      sr::set_error(__VA_ARGS__); \
      return 1;                \
    }                          \
  } while (0)

// tiles a training workspace holds for n_points points (sr_workspace_tiles): whole workgroups of 8 waves = 8 tiles
constexpr long ws_tiles(long n_points) { return ((n_points + 31) / 32 + 7) / 8 * 8; }
// bytes of the exponent-maxima table behind the dpre workspace (SR_FMT8; mlp_layout.h): 16 per 4 tiles, rounded up to whole 1-KiB units
constexpr long ws_emax_bytes(long n_points) { return (ws_tiles(n_points) / 4 * 16 + 1023) / 1024 * 1024; }

// ---- bf16 helpers -----------------------------------------------------------------------------
// Two fp32 -> one dword of two bf16 (RNE): element 0 in the low half.  Lowers to v_cvt_pk_bf16_f32.
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// Two fp32 -> one dword of two fp16 (RNE): v_cvt_pk_f16_f32.  Operand format of the SR_MODE_F16 forward kernels.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// the same with saturation to +-65504 (the fp16 weight stream: a scaled weight beyond the fp16 range must not become inf; a NaN becomes -65504 -- v_max_f32 returns the number)
__device__ __forceinline__ uint32_t pack_f16x2_sat(float a, float b) {
  return pack_f16x2(__builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f), __builtin_fminf(__builtin_fmaxf(b, -65504.f), 65504.f));
}
__device__ __forceinline__ float bf16_lo_to_f32(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// hi/lo split of a pair: hi = bf16(x), lo = bf16(x - hi).
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - bf16_lo_to_f32(hi), b - bf16_hi_to_f32(hi));
}

// sin(2*pi*x).  v_sin_f32 takes revolutions; valid for |x| <= 256.
__device__ __forceinline__ float sin_rev_fast(float x) { return __builtin_amdgcn_sinf(x); }

// sin(2*pi*x) to ~1 ulp-of-1: exact range reduction r = x - rint(x) in [-0.5,0.5], fold to
// [-0.25,0.25] by symmetry, then an odd degree-11 Taylor polynomial in t = 2*pi*r.
__device__ __forceinline__ float sin_rev_precise(float x) {
  float r = x - __builtin_rintf(x);              // exact in fp32
  float a = __builtin_fabsf(r);
  float f = (a > 0.25f) ? (0.5f - a) : a;        // sin(pi - t) = sin(t); exact subtraction
  f = __builtin_copysignf(f, r);
  const float t = f * 6.28318530717958647692f;
  const float t2 = t * t;
  float p = -2.5052108385441718775e-8f;           // -1/11!
  p = __builtin_fmaf(p, t2, 2.7557319223985890653e-6f);   // 1/9!
  p = __builtin_fmaf(p, t2, -1.9841269841269841253e-4f);  // -1/7!
  p = __builtin_fmaf(p, t2, 8.3333333333333332177e-3f);   // 1/5!
  p = __builtin_fmaf(p, t2, -1.6666666666666665741e-1f);  // -1/3!
  return __builtin_fmaf(p * t2, t, t);
}

// Streaming workspace traffic (activations / pre-activation gradients: written once, read once, hundreds of MB per step).
// Non-temporal stores / loads keep it from displacing the weight stream in L2: A/B on MI355X (profiles/r01_ab_variants.txt)
// forward+save 113 -> 98 us, dX 166 -> 153 us; the weight-gradient kernel, which re-reads through L2/MALL, prefers plain
// loads (216 vs 226 us) and uses ws_load_cached.
typedef unsigned int u32x4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ws_store(uint4* p, const uint4& v) {
#ifdef SR_ABL_NO_WS_STORE  // timing experiment (tools/ab_bwd_abl.sh): the value is computed, the store dropped
  asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(p));
  return;
#endif
#ifdef SR_WS_TEMPORAL  // A/B (tools/ab_nt.sh): default cache policy on the workspace traffic of this translation unit
  *reinterpret_cast<u32x4_native*>(p) = __builtin_bit_cast(u32x4_native, v);
  return;
#endif
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_native, v), reinterpret_cast<u32x4_native*>(p));
}
__device__ __forceinline__ uint4 ws_load(const uint4* p) {
#ifdef SR_WS_TEMPORAL
  return *p;
#endif
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_native*>(p)));
}
__device__ __forceinline__ uint4 ws_load_cached(const uint4* p) { return *p; }

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch Softplus(beta=1,threshold=20)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

}  // namespace sr
