// Device-side building blocks shared by the fused forward and backward (dX) MLP kernels: the LDS ring fed by
// asm LDS-DMA, the counted-vmcnt chunk protocol, MFMA wrappers and compile-time loops.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "mlp_layout.h"

namespace sr {
inline namespace SR_FEAT_NS {

constexpr int kD = 3;      // prefetch distance, chunks
constexpr int kNSLOT = 4;  // LDS ring slots (= kD + 1)
// A/B on MI355X (profiles/r01_ab_variants.txt): batching the A-fragment reads ahead of the MFMAs does not pay at two
// waves per SIMD (82.7 us un-batched vs 84.7 us with 9-deep batches), so the default leaves scheduling to hipcc.
#ifndef SR_TRUNK_BATCH
#define SR_TRUNK_BATCH 1
#endif
#ifndef SR_HEAD_BATCH
#define SR_HEAD_BATCH 1
#endif
#ifndef SR_SCHEDBAR
#define SR_SCHED_BARRIER() ((void)0)
#else
#define SR_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int kBatchTrunk = SR_TRUNK_BATCH;  // A fragments fetched per batch in the trunk
constexpr int kBatchHead = SR_HEAD_BATCH;   // ... and in the head stages (register pressure is higher there)

template <int NPASS>
struct Mode {
  // waves per workgroup: 8 (two per SIMD, <= 256 VGPRs) when a wave's activations fit 256 registers, else 4 (<= 512 VGPRs): the
  // parity mode (hi + lo planes) and the 512-wide forward (32 + 32 B fragments = 256 registers before anything else)
#ifdef SR_MODE1_NW  // a translation unit's own choice for its single-pass kernels (mlp_bwd.hip: 4 waves of two tiles each)
  static constexpr int NW = NPASS == 1 ? SR_MODE1_NW : 4;
#else
  static constexpr int NW = (NPASS == 1 && kFeat <= 256) ? 8 : 4;
#endif
  static constexpr int NPA = NPASS == 1 ? 1 : 2;  // A planes (hi[, lo])
  static constexpr int PIECE_BYTES = 1024 * NPA;
};

template <int NPASS, int N>
struct Frags {  // N B-fragments (k-steps) of the wave's 32 points
  uint4 hi[N];
  uint4 lo[NPASS == 3 ? N : 1];
};


// LDS-DMA through inline asm: hipcc then does not know an LDS write is in flight and does not drain vmcnt before
// every ds_read (it cannot tell ring slots apart); ordering is ours: counted vmcnt + s_barrier in wait_then_barrier().
// lds_addr is the wave-uniform LDS byte address (goes to M0); the 64 lanes land at lds_addr + lane*16.
__device__ __forceinline__ void glds16(const char* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// the same with the address split into a wave-uniform 64-bit base (SGPR pair: advanced by scalar adds, no VALU address
// arithmetic per piece) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t b = (uint64_t)(uintptr_t)sbase;  // readfirstlane: tells the compiler the base is uniform (folds away when it already is)
  const uint64_t ub = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(ub), "s"(lds_addr)
      : "memory");
}
// predicated form: `on` is wave-uniform; when it is 0 the request is issued with EXEC = 0 (no memory operation, no vmcnt
// increment) instead of being branched around -- a branch per potential request costs the in-order wave more than the
// dead issue slot
__device__ __forceinline__ void glds16_s_if(bool on, const char* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t ub = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t pred = __builtin_amdgcn_readfirstlane((int)on);
  uint32_t keep;
  uint64_t ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_cmp_lg_u32 %5, 0\n\ts_cselect_b64 exec, %1, 0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(ex)
      : "v"(voff), "s"(ub), "s"(lds_addr), "s"(pred)
      : "memory", "scc");
}
__device__ __forceinline__ uint32_t lds_addr_of(const char* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// ---- weight stream: L2 -> LDS ring ------------------------------------------------------------------
template <int NPASS, int NP>
__device__ __forceinline__ void issue_chunk(const char* hi, const char* lo, long off, char* slot, int wave, int lane) {
  constexpr int NW = Mode<NPASS>::NW, NT = NP * Mode<NPASS>::NPA, PER = (NT + NW - 1) / NW;
  const uint32_t slot_addr = __builtin_amdgcn_readfirstlane(lds_addr_of(slot));
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = wave + k * NW;  // wave-uniform
    const int piece = NPASS == 1 ? i : (i >> 1), plane = NPASS == 1 ? 0 : (i & 1);
#if SR_FEAT == 512
    // the 512-wide builds have no register to spare (256 VGPR + 256 AGPR): the SGPR-base form made hipcc spill (11-34 VGPRs, the
    // plain forward kernel 390 -> 837 us); they keep the per-lane 64-bit address
    if (i < NT) glds16((plane ? lo : hi) + off + piece * 1024 + lane * 16, slot_addr + piece * Mode<NPASS>::PIECE_BYTES + plane * 1024);
#elif defined(SR_DMA_BRANCH)
    if (i < NT) glds16_s((plane ? lo : hi) + off + piece * 1024, (uint32_t)lane * 16u, slot_addr + piece * Mode<NPASS>::PIECE_BYTES + plane * 1024);
#else
    if ((k + 1) * NW <= NT) {  // every wave has this piece: unconditional
      glds16_s((plane ? lo : hi) + off + piece * 1024, (uint32_t)lane * 16u, slot_addr + piece * Mode<NPASS>::PIECE_BYTES + plane * 1024);
    } else {                   // the ragged last round: predicated, not branched
      glds16_s_if(i < NT, (plane ? lo : hi) + off + piece * 1024, (uint32_t)lane * 16u, slot_addr + piece * Mode<NPASS>::PIECE_BYTES + plane * 1024);
    }
#endif
  }
}

// loads this wave has certainly issued for a chunk of NP pieces (lower bound over waves)
template <int NPASS>
constexpr int min_loads(int np) { return np * Mode<NPASS>::NPA / Mode<NPASS>::NW; }

template <int N>
__device__ __forceinline__ void wait_then_barrier() {
  // own LDS-DMA for the chunk about to be consumed has landed (all but the N newest VMEM ops done), every
  // ds_read of the slot about to be refilled has returned; then rendezvous.
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N > 63 ? 63 : N) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// single-pass operand format of this translation unit: bf16 (SR_MODE_BF16, and every backward kernel) or fp16 (SR_F16 builds of the
// forward kernel = SR_MODE_F16: same MFMA rate, 11 instead of 8 significand bits on weights and activations)
#ifdef SR_F16
__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_op2(float a, float b) { return pack_f16x2(a, b); }
#else
__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_op2(float a, float b) { return pack_bf16x2(a, b); }
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}


}  // inline namespace SR_FEAT_NS
}  // namespace sr
