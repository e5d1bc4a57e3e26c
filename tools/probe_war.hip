// Probe: is it safe to issue an LDS read INTO the A-operand registers of an MFMA right behind that MFMA (no wait, no buffer in between)?
// A 2-waves-per-SIMD weight-gradient kernel (VERDICT r05 #1a) only fits 256 registers per wave if the A operands are single-buffered: the
// read of the next k-step's A_a is issued as soon as the last MFMA that uses the current A_a has been ISSUED.  The ISA documents wait states
// for VALU writes against MFMA sources, nothing for LDS returns.  Each iteration: 4 MFMAs on A registers v[8:11] x B registers, then at once
// ds_read_b128 v[8:11] (the next iteration's A, other data) -- back to back with NO independent work -- against the same loop with a full
// drain (s_waitcnt + s_nop) before the overwrite.  Any difference in the accumulators = the hazard is real.
// hipcc --offload-arch=gfx950 -O3 tools/probe_war.hip -o build_variants/probe_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool SAFE>
__global__ void __launch_bounds__(128) k(const uint4* a_src, const uint4* b_src, float* out, int iters) {
  __shared__ uint4 lds[64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = a_src[i];
  __syncthreads();
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const uint4 b = b_src[lane];
  const u32x4 bq = {b.x, b.y, b.z, b.w};
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[lane]);
  asm volatile(
      "ds_read_b128 v[8:11], %[ad]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 s20, %[n]\n"
      "1:\n"
      "v_mfma_f32_32x32x16_bf16 %[c0], v[8:11], %[b], %[c0]\n"
      "v_mfma_f32_32x32x16_bf16 %[c1], v[8:11], %[b], %[c1]\n"
      "v_mfma_f32_32x32x16_bf16 %[c2], v[8:11], %[b], %[c2]\n"
      "v_mfma_f32_32x32x16_bf16 %[c3], v[8:11], %[b], %[c3]\n"
      ".if %c[safe]\n"
      "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
      ".endif\n"
      "v_add_u32 %[ad], 1024, %[ad]\n"
      "v_and_b32 %[ad], 0xffff, %[ad]\n"
      "ds_read_b128 v[8:11], %[ad]\n"      // overwrites the A operand of the four MFMAs just issued
      "s_waitcnt lgkmcnt(0)\n"
      "s_sub_u32 s20, s20, 1\n"
      "s_cmp_lg_u32 s20, 0\n"
      "s_cbranch_scc1 1b\n"
      "s_nop 15\n s_nop 15\n"
      : [c0] "+{v[16:31]}"(acc0), [c1] "+{v[32:47]}"(acc1), [c2] "+{v[48:63]}"(acc2), [c3] "+{v[64:79]}"(acc3), [ad] "+{v80}"(addr)
      : [b] "{v[12:15]}"(bq), [n] "s"(iters), [safe] "n"(SAFE ? 1 : 0)
      : "v8", "v9", "v10", "v11", "s20", "scc", "memory");
  float* o = out + ((size_t)blockIdx.x * 2 + wave) * 64 * 64 + lane * 64;
  for (int g = 0; g < 16; ++g) o[g] = acc0[g], o[16 + g] = acc1[g], o[32 + g] = acc2[g], o[48 + g] = acc3[g];
}
int main() {
  const int blocks = 1024, iters = 200;
  uint4 *a, *b; float *o0, *o1;
  (void)hipMalloc(&a, 64 * 64 * 16); (void)hipMalloc(&b, 64 * 16);
  (void)hipMalloc(&o0, (size_t)blocks * 2 * 64 * 64 * 4); (void)hipMalloc(&o1, (size_t)blocks * 2 * 64 * 64 * 4);
  unsigned short* h = (unsigned short*)malloc(64 * 64 * 16);
  srand(1);
  for (int i = 0; i < 64 * 64 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff)) | ((rand() & 1) << 15);   // bf16 values around +-(0.0078 .. 0.03)
  (void)hipMemcpy(a, h, 64 * 64 * 16, hipMemcpyHostToDevice);
  for (int i = 0; i < 64 * 8; ++i) h[i] = (unsigned short)(0x3f00 + (rand() & 0xff));
  (void)hipMemcpy(b, h, 64 * 16, hipMemcpyHostToDevice);
  size_t n = (size_t)blocks * 2 * 64 * 64;
  float* r0 = (float*)malloc(n * 4); float* r1 = (float*)malloc(n * 4);
  long bad_total = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(128), 0, 0, a, b, o0, iters);
    hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(128), 0, 0, a, b, o1, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(r0, o0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(r1, o1, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; double s = 0;
    for (size_t i = 0; i < n; ++i) { bad += r0[i] != r1[i]; s += r0[i]; }
    printf("rep %d: %ld of %zu accumulator values differ between the drained and the back-to-back loop (checksum %.6g)\n", rep, bad, n, s);
    bad_total += bad;
  }
  printf(bad_total == 0 ? "SAFE: an LDS read may overwrite the A operand of MFMAs already issued (2 waves per SIMD, 1024 workgroups x 200 iterations x 5)\n"
                        : "HAZARD: results differ\n");
  return 0;
}
