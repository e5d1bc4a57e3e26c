// Probe: what does each ingredient of the fused kernels cost IN ENERGY on a power-capped MI355X?  (r05)
//
// The training step runs at the 1400 W package limit (profiles/r04_power.txt): a kernel that keeps the matrix pipe busy is clocked down until
// its power fits, so its time follows its energy.  This probe runs hand-placed streams of v_mfma_f32_32x32x16 SUSTAINED (hundreds of
// back-to-back launches after a warm-up) on all-zero operands (what the stream's CYCLES cost at the un-throttled clock) and on random ones
// (what its ENERGY costs) and reports the wall time per MFMA slot of a SIMD.  One wave per SIMD (256 threads) or two (512), one workgroup per CU.
//   bf16 / f16        : MFMAs only, A / B fragments resident in registers (4 accumulators round robin)
//   lds1 / lds2       : the A fragment of every MFMA / of every second MFMA read from LDS (ds_read_b128, six / three reads ahead): 32 vs 64
//                       points per wave and weight fragment
//   dma               : lds1 + the forward's LDS-DMA refill (every wave re-loads 1 KiB of the ring per 8 MFMAs from an L2-resident stream)
//   sin, perm4, pk4   : bf16 + 1 v_sin_f32 / 4 v_perm_b32 / 4 v_pk_mul_f32 per MFMA gap
//   hbm16 / hbm8      : bf16 + 1 KiB streamed from a 1-GiB buffer per wave and 16 / 8 MFMAs (3.6 / 6 TB/s chip-wide)
//   fillers           : every VALU / LDS instruction class the kernels use, 2 and 4 copies behind every MFMA (which co-issue, which hold the
//                       matrix pipe); the same under EXEC = 0; 4- vs 8-byte encodings at 6 / 8 / 10 per MFMA (instruction fetch)
// Results: profiles/r05_probe_power.txt, read in DESIGN.md section 4.
// hipcc --offload-arch=gfx950 -O3 tools/probe_power.hip -o /tmp/probe_power && /tmp/probe_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { V_BF16 = 0, V_F16, V_LDS1, V_LDS2, V_SIN, V_PERM4, V_PK4, V_DMA, V_HBM16, V_HBM8, V_COUNT };
static const char* kNames[V_COUNT] = {"mfma bf16 (regs)", "mfma f16 (regs)", "bf16 + A from LDS per MFMA", "bf16 + A from LDS per 2 MFMAs",
                                      "bf16 + 1 v_sin per MFMA", "bf16 + 4 v_perm per MFMA", "bf16 + 4 v_pk_mul_f32 per MFMA",
                                      "bf16 + A from LDS + LDS-DMA refill", "bf16 + 1 KiB from HBM per wave and 16 MFMAs", "bf16 + 1 KiB from HBM per wave and 8 MFMAs"};
constexpr int kSlots = 64;  // MFMAs per loop iteration

// operands: `ops` holds 8 KiB per wave-lane set: fragment f (0..7) of lane l at ops[(f * 64 + l)] (uint4); LDS ring of 64 KiB filled from `ring`
template <int V>
__global__ void __launch_bounds__(512) k(const uint4* __restrict__ ops, const uint4* __restrict__ ring, float* out, int iters, const char* big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = ring[i];
  __syncthreads();
  u32x4 a[4], b[2];
#pragma unroll
  for (int f = 0; f < 4; ++f) a[f] = __builtin_bit_cast(u32x4, ops[f * 64 + lane]);
#pragma unroll
  for (int f = 0; f < 2; ++f) b[f] = __builtin_bit_cast(u32x4, ops[(4 + f) * 64 + lane]);
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float s0 = 0.3f + lane * 1e-3f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
  const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024);
  const uint64_t gsrc = (uint64_t)(uintptr_t)ring;
  uint32_t goff = lane * 16 + (threadIdx.x >> 6) * 1024;
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == V_BF16 || V == V_F16) {
#define MF(op, acc, af, bf) "v_mfma_f32_32x32x16_" op " %[c" #acc "], %[a" #af "], %[b" #bf "], %[c" #acc "]\n"
#define RR16(op) MF(op, 0, 0, 0) MF(op, 1, 1, 1) MF(op, 2, 2, 0) MF(op, 3, 3, 1) MF(op, 0, 1, 0) MF(op, 1, 2, 1) MF(op, 2, 3, 0) MF(op, 3, 0, 1) \
                 MF(op, 0, 2, 1) MF(op, 1, 3, 0) MF(op, 2, 0, 1) MF(op, 3, 1, 0) MF(op, 0, 3, 1) MF(op, 1, 0, 0) MF(op, 2, 1, 1) MF(op, 3, 2, 0)
      if constexpr (V == V_BF16)
        asm volatile(RR16("bf16") RR16("bf16") RR16("bf16") RR16("bf16") : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3) : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]));
      else
        asm volatile(RR16("f16") RR16("f16") RR16("f16") RR16("f16") : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3) : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]));
    } else if constexpr (V == V_SIN || V == V_PERM4 || V == V_PK4) {
#define FS "v_sin_f32 %[s1], %[s0]\n"
#define FP "v_perm_b32 %[s1], %[s2], %[s3], %[s0]\n v_perm_b32 %[s2], %[s3], %[s1], %[s0]\n v_perm_b32 %[s3], %[s1], %[s2], %[s0]\n v_perm_b32 %[s1], %[s3], %[s2], %[s0]\n"
#define FK "v_pk_mul_f32 %[p0], %[p1], %[p1]\n v_pk_mul_f32 %[p1], %[p0], %[p0]\n v_pk_mul_f32 %[p0], %[p1], %[p1]\n v_pk_mul_f32 %[p1], %[p0], %[p0]\n"
#define MG(F, acc, af, bf) MF("bf16", acc, af, bf) F
#define RG16(F) MG(F, 0, 0, 0) MG(F, 1, 1, 1) MG(F, 2, 2, 0) MG(F, 3, 3, 1) MG(F, 0, 1, 0) MG(F, 1, 2, 1) MG(F, 2, 3, 0) MG(F, 3, 0, 1) \
                MG(F, 0, 2, 1) MG(F, 1, 3, 0) MG(F, 2, 0, 1) MG(F, 3, 1, 0) MG(F, 0, 3, 1) MG(F, 1, 0, 0) MG(F, 2, 1, 1) MG(F, 3, 2, 0)
      if constexpr (V == V_SIN)
        asm volatile(RG16(FS) RG16(FS) RG16(FS) RG16(FS) : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [s1] "+v"(s1) : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [s0] "v"(s0));
      else if constexpr (V == V_PERM4)
        asm volatile(RG16(FP) RG16(FP) RG16(FP) RG16(FP) : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3) : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [s0] "v"(s0));
      else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 p0 = {s0, 0.99f}, p1 = {0.98f, s0};
        asm volatile(RG16(FK) RG16(FK) RG16(FK) RG16(FK) : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [p0] "+v"(p0), [p1] "+v"(p1) : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]));
        s1 += p0[0];
      }
    } else if constexpr (V == V_HBM16 || V == V_HBM8) {
      // workspace-like traffic: every wave streams its own region of a 1-GiB buffer (beyond the 256-MB Infinity Cache) with 1-KiB loads
      constexpr int EVERY = V == V_HBM16 ? 16 : 8;
      const uint64_t gb = (uint64_t)(uintptr_t)big;
      uint32_t ho = (uint32_t)(((((unsigned long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (unsigned long)iters + it) * (64 / EVERY) * 1024ul) & ((1ul << 30) - 16384ul)) + lane * 16;
      asm volatile(".set n, 0\n.rept 64\n"
                   "v_mfma_f32_32x32x16_bf16 v[32+16*(n%%4):47+16*(n%%4)], %[a0], %[b0], v[32+16*(n%%4):47+16*(n%%4)]\n"
                   ".if (n %% %c[ev]) == 0\n global_load_dwordx4 v[100+4*((n/%c[ev])%%8):103+4*((n/%c[ev])%%8)], %[ho], %[gb] offset:0\n v_add_u32 %[ho], 1024, %[ho]\n.endif\n"
                   ".set n, n+1\n.endr\n s_waitcnt vmcnt(0)\n"
                   : "+{v[32:47]}"(c0), "+{v[48:63]}"(c1), "+{v[64:79]}"(c2), "+{v[80:95]}"(c3), [ho] "+v"(ho)
                   : [a0] "v"(a[0]), [b0] "v"(b[0]), [gb] "s"(gb), [ev] "n"(EVERY)
                   : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                     "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "memory");
    } else {
      // A fragments through an 8-deep register ring v[100:131]; piece p of the LDS ring at lds + 1024 p (64 pieces).  lds1: the read of slot
      // n + 6 is issued in front of MFMA n (six ahead, in-order returns: lgkmcnt(6) = read n has landed); lds2 (64 points per wave): one read
      // per MFMA PAIR, three pairs ahead.  Accumulators in fixed registers so that the assembler can index them.
#define CLOB_RING "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
                  "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "memory"
#define ACC_IO "+{v[32:47]}"(c0), "+{v[48:63]}"(c1), "+{v[64:79]}"(c2), "+{v[80:95]}"(c3)
#define RD(n) "ds_read_b128 v[100+4*((" n ")%%8):103+4*((" n ")%%8)], %[lb] offset:((" n ")%%64)*1024\n"
#define MFN(n, bsel) "v_mfma_f32_32x32x16_bf16 v[32+16*((" n ")%%4):47+16*((" n ")%%4)], v[100+4*((" n ")%%8):103+4*((" n ")%%8)], " bsel ", v[32+16*((" n ")%%4):47+16*((" n ")%%4)]\n"
      if constexpr (V == V_LDS1) {
        asm volatile(RD("0") RD("1") RD("2") RD("3") RD("4") RD("5")
                     ".set n, 0\n.rept 64\n" RD("n+6") "s_waitcnt lgkmcnt(6)\n" MFN("n", "%[b0]") ".set n, n+1\n.endr\n s_waitcnt lgkmcnt(0)\n"
                     : ACC_IO : [b0] "v"(b[0]), [b1] "v"(b[1]), [lb] "v"(lbase) : CLOB_RING);
      } else if constexpr (V == V_DMA) {
        // every wave also refills ONE 1-KiB piece per 8 MFMAs: the forward's rate (a row of 8 pieces per 8 MFMAs and workgroup)
        uint32_t m0s;
        asm volatile("s_mov_b32 %[m0s], m0\n" RD("0") RD("1") RD("2") RD("3") RD("4") RD("5")
                     ".set n, 0\n.rept 64\n" RD("n+6") "s_waitcnt lgkmcnt(6)\n" MFN("n", "%[b0]")
                     ".if (n %% 8) == 7\n s_add_u32 m0, %[wb], ((n+1)%%64)*1024\n s_nop 0\n global_load_lds_dwordx4 %[go], %[gs]\n.endif\n"
                     ".set n, n+1\n.endr\n s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n s_mov_b32 m0, %[m0s]\n"
                     : ACC_IO, [m0s] "=&s"(m0s) : [b0] "v"(b[0]), [b1] "v"(b[1]), [lb] "v"(lbase), [wb] "s"(wbase), [go] "v"(goff), [gs] "s"(gsrc) : CLOB_RING);
        goff = (goff + 8192u) & ((1u << 20) - 1u);  // walks 1 MiB of the L2-resident stream
      } else {
#define MFP(n) "v_mfma_f32_32x32x16_bf16 v[32+32*((" n ")%%2):47+32*((" n ")%%2)], v[100+4*((" n ")%%8):103+4*((" n ")%%8)], %[b0], v[32+32*((" n ")%%2):47+32*((" n ")%%2)]\n" \
               "v_mfma_f32_32x32x16_bf16 v[48+32*((" n ")%%2):63+32*((" n ")%%2)], v[100+4*((" n ")%%8):103+4*((" n ")%%8)], %[b1], v[48+32*((" n ")%%2):63+32*((" n ")%%2)]\n"
        asm volatile(RD("0") RD("1") RD("2")
                     ".set n, 0\n.rept 32\n" RD("n+3") "s_waitcnt lgkmcnt(3)\n" MFP("n") ".set n, n+1\n.endr\n s_waitcnt lgkmcnt(0)\n"
                     : ACC_IO : [b0] "v"(b[0]), [b1] "v"(b[1]), [lb] "v"(lbase) : CLOB_RING);
      }
    }
  }
  float r = s1 + s2 + s3;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  if (r == 12345.678f) out[threadIdx.x] = r;  // never true: keeps the results alive
}


// ---- which VALU instructions run BESIDE an MFMA, and which take the matrix pipe's issue slot? ----------------------------------------
// K copies of one filler instruction behind every MFMA (independent destinations v[16:23], sources v[24:31]); reported on all-zero
// operands (cycles) and on random ones (energy).  A filler that co-issues leaves the zero-operand time at ~33 cycles per MFMA.
#define FILLERS(X)                                                                         \
  X(0, "v_mul_f32", "v_mul_f32 v[16+d], v[24+d], v[28]\n")                                \
  X(1, "v_fma_f32", "v_fma_f32 v[16+d], v[24+d], v[28], v[29]\n")                         \
  X(2, "v_pk_mul_f32", "v_pk_mul_f32 v[16+2*(d%%4):17+2*(d%%4)], v[24:25], v[26:27]\n")  \
  X(3, "v_pk_fma_f32", "v_pk_fma_f32 v[16+2*(d%%4):17+2*(d%%4)], v[24:25], v[26:27], v[28:29]\n") \
  X(4, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 v[16+d], v[24+d], v[28]\n")                \
  X(5, "v_max3_f32 |a|,|b|,|c|", "v_max3_f32 v[16+d], |v[24+d]|, |v[28]|, |v[29]|\n")     \
  X(6, "v_cos_f32", "v_cos_f32 v[16+d], v[24+d]\n")                                       \
  X(7, "v_perm_b32", "v_perm_b32 v[16+d], v[24+d], v[28], v[29]\n")                       \
  X(8, "v_pk_fma_f16", "v_pk_fma_f16 v[16+d], v[24+d], v[28], v[29]\n")                   \
  X(9, "v_pk_mul_f16", "v_pk_mul_f16 v[16+d], v[24+d], v[28]\n")                          \
  X(10, "v_lshl_or_b32", "v_lshl_or_b32 v[16+d], v[24+d], 8, v[28]\n")                    \
  X(11, "v_max_u32_sdwa", "v_max_u32_sdwa v[16+d], v[24+d], v[28] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n") \
  X(12, "v_accvgpr_write_b32", "v_accvgpr_write_b32 a[16+d], v[24+d]\n")                  \
  X(13, "v_pk_add_f32", "v_pk_add_f32 v[16+2*(d%%4):17+2*(d%%4)], v[24:25], v[26:27]\n") \
  X(14, "v_sin_f16 (sdwa)", "v_sin_f16_sdwa v[16+d], v[24+d] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n") \
  X(15, "v_add_f32 dpp row_shr:1", "v_add_f32_dpp v[16+d], v[24+d], v[28] row_shr:1 row_mask:0xf bank_mask:0xf\n") \
  X(16, "v_mov_b32", "v_mov_b32 v[16+d], v[24+d]\n")                                      \
  X(17, "v_fmac_f32", "v_fmac_f32 v[16+d], v[24+d], v[28]\n")                             \
  X(18, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 v[16+d], v[24+d], v[28]\n")                 \
  X(19, "v_exp_f32", "v_exp_f32 v[16+d], v[24+d]\n")                                      \
  X(20, "ds_read_b64_tr_b16", "ds_read_b64_tr_b16 v[16+2*(d%%4):17+2*(d%%4)], v30 offset:64*d\n") \
  X(21, "ds_write_b128", "ds_write_b128 v30, v[24:27] offset:1024*d\n")                   \
  X(22, "v_perm_b32 under EXEC = 0", "s_mov_b64 exec, 0\n v_perm_b32 v[16+d], v[24+d], v[28], v[29]\n s_mov_b64 exec, -1\n") \
  X(23, "ds_write_b128 under EXEC = 0", "s_mov_b64 exec, 0\n ds_write_b128 v30, v[24:27] offset:1024*d\n s_mov_b64 exec, -1\n") \
  X(24, "2 x s_mov_b64 exec only", "s_mov_b64 exec, 0\n s_mov_b64 exec, -1\n")
constexpr int kFillers = 25;

template <int F, int K>
__global__ void __launch_bounds__(512) kf(const uint4* __restrict__ ops, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  u32x4 a0 = __builtin_bit_cast(u32x4, ops[lane]), a1 = __builtin_bit_cast(u32x4, ops[64 + lane]);
  u32x4 b0 = __builtin_bit_cast(u32x4, ops[4 * 64 + lane]), s0 = __builtin_bit_cast(u32x4, ops[5 * 64 + lane]), s1 = __builtin_bit_cast(u32x4, ops[6 * 64 + lane]);
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const uint32_t lb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (threadIdx.x >> 6) * 8192;
  for (int it = 0; it < iters; ++it) {
#define X(i, name, text)                                                                                                               \
  if constexpr (F == i)                                                                                                                \
    asm volatile(".set n, 0\n.set m, 0\n.rept 64\n"                                                                                    \
                 "v_mfma_f32_32x32x16_bf16 v[32+16*(n%%4):47+16*(n%%4)], v[0+4*(n%%2):3+4*(n%%2)], v[8:11], v[32+16*(n%%4):47+16*(n%%4)]\n" \
                 ".rept %c[k]\n.set d, m%%8\n" text ".set m, m+1\n.endr\n.set n, n+1\n.endr\n s_waitcnt lgkmcnt(0)\n"                    \
                 : "+{v[32:47]}"(c0), "+{v[48:63]}"(c1), "+{v[64:79]}"(c2), "+{v[80:95]}"(c3)                                          \
                 : "{v[0:3]}"(a0), "{v[4:7]}"(a1), "{v[8:11]}"(b0), "{v[24:27]}"(s0), "{v[28:31]}"(s1), "{v30}"(lb), [k] "n"(K)      \
                 : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "memory");
    FILLERS(X)
#undef X
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int F, int K>
static double time_filler(int waves, const uint4* ops, float* out) {
  const int iters = 60;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)kf<F, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  auto launch = [&] { hipLaunchKernelGGL((kf<F, K>), dim3(256), dim3(64 * 4 * waves), 65536, 0, ops, out, iters); };
  for (int i = 0; i < 200; ++i) launch();
  CHECK(hipDeviceSynchronize());
  const int reps = 200;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e6 / reps / ((double)iters * 64 * waves);
}

// instruction FETCH: the same fillers at 6 / 8 / 10 per MFMA, 4-byte against 8-byte encodings (zero operands: cycles only).  Two waves per
// SIMD x 4 SIMDs fetch 8 x (8 + K x size) bytes per 32-cycle MFMA slot from the CU's instruction cache.
template <int F>
static void run_fetch(const char* name, int bytes, const uint4* zops, float* out) {
  for (int waves = 1; waves <= 2; ++waves) {
    const double z6 = time_filler<F, 6>(waves, zops, out), z8 = time_filler<F, 8>(waves, zops, out), z10 = time_filler<F, 10>(waves, zops, out);
    printf("%-26s (%d-byte encoding) %d wave/SIMD | 6 per MFMA %5.1f cycles (%4.1f code bytes per cycle and CU) | 8: %5.1f (%4.1f) | 10: %5.1f (%4.1f)\n", name, bytes, waves,
           z6 * 2.4, 4.0 * waves * (8 + 6 * bytes) / (z6 * 2.4), z8 * 2.4, 4.0 * waves * (8 + 8 * bytes) / (z8 * 2.4), z10 * 2.4, 4.0 * waves * (8 + 10 * bytes) / (z10 * 2.4));
  }
}

template <int F>
static void run_filler(const char* name, const uint4* ops, const uint4* zops, float* out) {
  for (int waves = 1; waves <= 2; ++waves) {
    const double z2 = time_filler<F, 2>(waves, zops, out), z4 = time_filler<F, 4>(waves, zops, out), r4 = time_filler<F, 4>(waves, ops, out);
    printf("%-26s %d wave/SIMD | zero operands: 2 per MFMA %5.1f cycles, 4 per MFMA %5.1f (%+5.1f per instruction beyond 33) | random, 4 per MFMA: %6.2f ns = %5.1f cycles at 2.4 GHz\n",
           name, waves, z2 * 2.4, z4 * 2.4, (z4 * 2.4 - 33.0) / 4, r4, r4 * 2.4);
  }
}

static uint16_t to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static uint16_t to_f16(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

static const char* g_big = nullptr;
template <int V>
static double time_one(int waves, const uint4* ops, const uint4* ring, float* out, int slots_per_iter) {
  const int iters = 100;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  auto launch = [&] { hipLaunchKernelGGL(k<V>, dim3(256), dim3(64 * 4 * waves), 65536, 0, ops, ring, out, iters, g_big); };
  for (int i = 0; i < 250; ++i) launch();  // >= 25 ms: the chip settles at its power limit
  CHECK(hipDeviceSynchronize());
  const int reps = 300;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e6 / reps / ((double)iters * slots_per_iter * waves);  // ns per MFMA and SIMD
}

// every stream twice: on all-zero operands (what its CYCLES cost at the un-throttled clock) and on random ones (what its ENERGY costs)
template <int V>
static void run(const char* name, int waves, const uint4* ops, const uint4* ring, const uint4* zops, const uint4* zring, float* out) {
  const double z = time_one<V>(waves, zops, zring, out, kSlots), r = time_one<V>(waves, ops, ring, out, kSlots);
  printf("%-38s %d wave/SIMD | zero operands %6.2f ns per MFMA and SIMD = %5.1f cycles at 2.4 GHz | random %6.2f ns = %5.1f (%4.0f TFLOP/s, x%.2f of zero)\n", name,
         waves, z, z * 2.4, r, r * 2.4, 1024.0 * 32768 / r / 1e3, r / z);
}

int main() {
  std::vector<uint16_t> h(8 * 64 * 8), z(8 * 64 * 8, 0), hf(8 * 64 * 8), hr(65536 / 2);
  srand(1);
  auto rnd = [] { float s = 0; for (int i = 0; i < 6; ++i) s += rand() / (float)RAND_MAX - 0.5f; return s * 0.35f; };  // ~N(0, 0.25^2)
  for (size_t i = 0; i < h.size(); ++i) { const float v = rnd(); h[i] = to_bf16(v), hf[i] = to_f16(v); }
  std::vector<uint16_t> big(1 << 20);  // 2 MiB stream for the DMA variant (its first 64 KiB also fills the LDS ring)
  for (size_t i = 0; i < big.size(); ++i) big[i] = to_bf16(rnd());
  uint4 *d_bf, *d_f16, *d_zero, *d_ring, *d_zring;
  float* d_out;
  CHECK(hipMalloc(&d_bf, h.size() * 2));
  CHECK(hipMalloc(&d_f16, h.size() * 2));
  CHECK(hipMalloc(&d_zero, h.size() * 2));
  CHECK(hipMalloc(&d_ring, big.size() * 2));
  CHECK(hipMalloc(&d_zring, big.size() * 2));
  CHECK(hipMalloc(&d_out, 4096));
  CHECK(hipMemcpy(d_bf, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_f16, hf.data(), h.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(d_zero, 0, h.size() * 2));
  CHECK(hipMemset(d_zring, 0, big.size() * 2));
  CHECK(hipMemcpy(d_ring, big.data(), big.size() * 2, hipMemcpyHostToDevice));
  {
    char* big;
    CHECK(hipMalloc(&big, 1ul << 30));
    CHECK(hipMemset(big, 0x3c, 1ul << 30));
    g_big = big;
  }
  for (int waves = 1; waves <= 2; ++waves) {
    run<V_BF16>(kNames[V_BF16], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_F16>(kNames[V_F16], waves, d_f16, d_ring, d_zero, d_zring, d_out);
    run<V_LDS1>(kNames[V_LDS1], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_LDS2>(kNames[V_LDS2], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_DMA>(kNames[V_DMA], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_SIN>(kNames[V_SIN], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_PERM4>(kNames[V_PERM4], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_PK4>(kNames[V_PK4], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_HBM16>(kNames[V_HBM16], waves, d_bf, d_ring, d_zero, d_zring, d_out);
    run<V_HBM8>(kNames[V_HBM8], waves, d_bf, d_ring, d_zero, d_zring, d_out);
  }
  printf("---- fillers beside the MFMA (bf16, operands in registers) ----\n");
#define X(i, name, text) run_filler<i>(name, d_bf, d_zero, d_out);
  FILLERS(X)
#undef X
  printf("---- dead work under EXEC = 0 (8 / 10 per MFMA, zero operands) ----\n");
  run_fetch<22>("v_perm_b32 under EXEC = 0", 8, d_zero, d_out);
  run_fetch<24>("2 x s_mov_b64 exec only", 0, d_zero, d_out);
  printf("---- instruction fetch ----\n");
  run_fetch<16>("v_mov_b32", 4, d_zero, d_out);
  run_fetch<0>("v_mul_f32", 4, d_zero, d_out);
  run_fetch<7>("v_perm_b32", 8, d_zero, d_out);
  run_fetch<1>("v_fma_f32", 8, d_zero, d_out);
  run_fetch<5>("v_max3_f32 |a|,|b|,|c|", 8, d_zero, d_out);
  return 0;
}
