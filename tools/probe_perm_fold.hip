// tools/probe_perm_fold.hip: a front-end trap met while packing MX8 bytes (csrc/codec8.h).  __builtin_bit_cast(uint32_t, r[1]) with r an
// ext_vector_type(2) float reads the FIRST four bytes of the vector (hipcc 7.2 / clang 22): the IR holds extractelement ..., i64 0 for both
// elements, so bytes4(t0, t1, t2, t3) below packs [t0, t0, t2, t2].  Copying the element to a float first (kernel `good`) is correct.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/probe_perm_fold.hip -o - | grep v_perm
//   bad:  v_perm_b32 v1, v2, v2, s1      good:  v_perm_b32 v1, v3, v2, s1
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bytes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u);
  const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
  return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
template <bool COPY>
__device__ __forceinline__ void body(const float* in, uint32_t* out, float inv) {
  float v[4];
  for (int i = 0; i < 4; ++i) v[i] = in[threadIdx.x * 4 + i];
  const f32x2 inv2 = {inv, inv}, magic = {12583040.0f, 12583040.0f};
  uint32_t t[4];
  for (int g = 0; g < 4; g += 2) {
    const f32x2 pair = {v[g], v[g + 1]};
    const f32x2 r = __builtin_elementwise_fma(pair, inv2, magic);
    if constexpr (COPY) {
      const float r0 = r[0], r1 = r[1];
      t[g] = __builtin_bit_cast(uint32_t, r0), t[g + 1] = __builtin_bit_cast(uint32_t, r1);
    } else {
      t[g] = __builtin_bit_cast(uint32_t, r[0]), t[g + 1] = __builtin_bit_cast(uint32_t, r[1]);
    }
  }
  out[threadIdx.x] = bytes4(t[0], t[1], t[2], t[3]);
}
__global__ void bad(const float* in, uint32_t* out, float inv) { body<false>(in, out, inv); }
__global__ void good(const float* in, uint32_t* out, float inv) { body<true>(in, out, inv); }
