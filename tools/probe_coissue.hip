// Probe: do v_mfma_f32_32x32x16_bf16 and the SIREN epilogue's VALU work (v_sin_f32, v_cvt_pk_bf16_f32) overlap on one SIMD of
// gfx950, and what does the shape of the instruction stream cost?  (task 2.)
//
// Every variant is ONE hand-placed instruction stream (a single asm statement: hipcc schedules nothing inside it) executed by every
// wave of a 256-workgroup launch (one workgroup per CU; 256 threads = one wave per SIMD, 512 threads = two waves per SIMD).
// Reported per variant: shader cycles per MFMA slot (s_memtime around the loop, wave 0 of workgroup 0 and the slowest sampled wave)
// and the wall time of the launch.  "slot" = one MFMA of the stream, or for the VALU-only streams the VALU work that the
// interleaved variants place beside one MFMA.
//
// Streams (T = one "tile" = 17 MFMAs = 16 k-steps + 1 aux k-step of a 32-row output tile, as in csrc/mlp_fwd.inc):
//   mfma_dep      T on ONE accumulator (every MFMA takes the previous one's D as C)
//   mfma_2acc/4   the same MFMAs round-robin over 2 / 4 accumulators
//   valu_*        only the fillers of the corresponding interleaved stream
//   dep_*         mfma_dep with k fillers after every MFMA        (s = v_sin, c = v_cvt_pk, f = v_fma; independent registers)
//   2acc_* 4acc_* the same on 2 / 4 accumulators
//   tile1         the forward kernel's software pipeline: 17 MFMAs on accumulator X, between them the epilogue of the PREVIOUS
//                 tile (16 v_sin + 8 v_cvt_pk READING accumulator Y), then X <-> Y
//   tile2         the same with TWO chains per tile pair (64 points per wave: accumulators X0, X1 share the A fragment; the
//                 epilogue reads Y0, Y1): 34 MFMAs + 32 sin + 16 cvt
//   serial1       17 MFMAs, THEN the 24 epilogue instructions (no interleave): what two waves per SIMD can overlap by themselves
//   tile1_lds / tile2_lds / serial1_lds   the same with the A fragment of every MFMA read from LDS (ds_read_b128, 2 reads ahead,
//                 counted lgkmcnt): 1 read per MFMA (tile1) or per 2 MFMAs (tile2)
//
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/probe_coissue.bin tools/probe_coissue.hip && tools/probe_coissue.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// register map of every stream (all named explicitly; listed as clobbers):
//   v[0:3]   A fragment (register streams)         v[4:7], v[8:11]  B fragments of chain 0 / 1
//   v[16:31] filler destinations                   v[32:47] filler sources (independent streams)
//   v[64:79] v[80:95] v[96:111] v[112:127]         accumulators 0..3
//   v[128:143] A fragments read from LDS (ring of 4)   v144 LDS address
#define CLOB                                                                                                                        \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",  \
      "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", \
      "v42", "v43", "v44", "v45", "v46", "v47", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
      "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", \
      "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",     \
      "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",       \
      "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139",       \
      "v140", "v141", "v142", "v143", "v144", "memory"

// assembler macros shared by the streams.  Symbols: n = running slot index.
#define MACROS                                                                                     \
  ".macro MF acc\n v_mfma_f32_32x32x16_bf16 v[\\acc:\\acc+15], v[0:3], v[4:7], v[\\acc:\\acc+15]\n.endm\n"     \
  ".macro MFB acc, b\n v_mfma_f32_32x32x16_bf16 v[\\acc:\\acc+15], v[0:3], v[\\b:\\b+3], v[\\acc:\\acc+15]\n.endm\n" \
  ".macro MFL acc, b, a\n v_mfma_f32_32x32x16_bf16 v[\\acc:\\acc+15], v[\\a:\\a+3], v[\\b:\\b+3], v[\\acc:\\acc+15]\n.endm\n" \
  ".macro SIN k\n v_sin_f32 v[16+((\\k)%%16)], v[32+((\\k)%%16)]\n.endm\n"                           \
  ".macro CVT k\n v_cvt_pk_bf16_f32 v[16+((\\k)%%16)], v[32+((\\k)%%16)], v[32+((\\k+1)%%16)]\n.endm\n" \
  ".macro FMA k\n v_fma_f32 v[16+((\\k)%%16)], v[32+((\\k)%%16)], v[32+((\\k+1)%%16)], v[32+((\\k+2)%%16)]\n.endm\n" \
  ".macro SINA src, k\n v_sin_f32 v[16+((\\k)%%16)], v[\\src+((\\k)%%16)]\n.endm\n"                  \
  ".macro CVTP k\n v_cvt_pk_bf16_f32 v[32+((\\k)%%8)], v[16+((2*(\\k)+14)%%16)], v[16+((2*(\\k)+15)%%16)]\n.endm\n" \
  ".macro LDA k\n ds_read_b128 v[128+4*((\\k)%%4):131+4*((\\k)%%4)], v144 offset:1024*((\\k)%%16)\n.endm\n"
#define PURGE ".purgem MF\n.purgem MFB\n.purgem MFL\n.purgem SIN\n.purgem CVT\n.purgem FMA\n.purgem SINA\n.purgem CVTP\n.purgem LDA\n"

enum Variant {
  V_MFMA_DEP, V_MFMA_2ACC, V_MFMA_4ACC,
  V_VALU_S, V_VALU_SC, V_VALU_SSC, V_VALU_5,
  V_DEP_F, V_DEP_S, V_DEP_SC, V_DEP_SSC, V_DEP_5,
  V_2ACC_F, V_2ACC_S, V_2ACC_SC, V_2ACC_SSC, V_2ACC_5,
  V_4ACC_SC, V_4ACC_SSC, V_4ACC_5,
  V_TILE1, V_TILE2, V_SERIAL1,
  V_TILE1_LDS, V_TILE2_LDS, V_SERIAL1_LDS, V_MFMA_LDS1, V_MFMA_LDS2,
  V_COUNT
};
static const char* kNames[V_COUNT] = {
  "mfma_dep", "mfma_2acc", "mfma_4acc",
  "valu_s (1 sin/slot)", "valu_sc (1 sin + .5 cvt)", "valu_ssc (2 sin + 1 cvt)", "valu_5 (3 sin + 2 cvt)",
  "dep + 1 fma", "dep + 1 sin", "dep + 1 sin + .5 cvt", "dep + 2 sin + 1 cvt", "dep + 3 sin + 2 cvt",
  "2acc + 1 fma", "2acc + 1 sin", "2acc + 1 sin + .5 cvt", "2acc + 2 sin + 1 cvt", "2acc + 3 sin + 2 cvt",
  "4acc + 1 sin + .5 cvt", "4acc + 2 sin + 1 cvt", "4acc + 3 sin + 2 cvt",
  "tile1 (1 chain, epilogue of prev tile between)", "tile2 (2 chains share A, epilogue between)", "serial1 (17 MFMA then 24 VALU)",
  "tile1_lds (+ 1 ds_read_b128 per MFMA)", "tile2_lds (+ 1 ds_read_b128 per 2 MFMA)", "serial1_lds", "mfma_dep_lds (reads only)", "mfma_2chain_lds (reads only)",
};
// MFMA slots per loop iteration (the divisor of the cycle count)
static const int kSlots[V_COUNT] = {136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136, 136,
                                    136, 136, 136, 136, 136, 136, 136, 136};

#define BODY_BEGIN asm volatile(MACROS ".set n, 0\n"
#define BODY_END PURGE ::"v"(lds_addr) : CLOB)

template <int V>
__device__ __forceinline__ void body(uint32_t lds_addr) {
  // every stream first moves the LDS address into v144 (unused by the register streams)
  if constexpr (V == V_MFMA_DEP) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n MF 64\n .endr\n" BODY_END;
  } else if constexpr (V == V_MFMA_2ACC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n MF 80\n .endr\n" BODY_END;
  } else if constexpr (V == V_MFMA_4ACC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 34\n MF 64\n MF 80\n MF 96\n MF 112\n .endr\n" BODY_END;
  } else if constexpr (V == V_VALU_S) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n SIN n\n .set n, n+1\n .endr\n" BODY_END;
  } else if constexpr (V == V_VALU_SC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n SIN n\n SIN n+1\n CVT n+2\n .set n, n+3\n .endr\n" BODY_END;
  } else if constexpr (V == V_VALU_SSC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n SIN n\n SIN n+1\n CVT n+2\n .set n, n+3\n .endr\n" BODY_END;
  } else if constexpr (V == V_VALU_5) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n SIN n\n SIN n+1\n CVT n+2\n SIN n+3\n CVT n+4\n .set n, n+5\n .endr\n" BODY_END;
  } else if constexpr (V == V_DEP_F) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n MF 64\n FMA n\n .set n, n+1\n .endr\n" BODY_END;
  } else if constexpr (V == V_DEP_S) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n MF 64\n SIN n\n .set n, n+1\n .endr\n" BODY_END;
  } else if constexpr (V == V_DEP_SC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n SIN n\n MF 64\n SIN n+1\n CVT n+2\n .set n, n+3\n .endr\n" BODY_END;
  } else if constexpr (V == V_DEP_SSC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n MF 64\n SIN n\n SIN n+1\n CVT n+2\n .set n, n+3\n .endr\n" BODY_END;
  } else if constexpr (V == V_DEP_5) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 136\n MF 64\n SIN n\n SIN n+1\n CVT n+2\n SIN n+3\n CVT n+4\n .set n, n+5\n .endr\n" BODY_END;
  } else if constexpr (V == V_2ACC_F) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n FMA n\n MF 80\n FMA n+1\n .set n, n+2\n .endr\n" BODY_END;
  } else if constexpr (V == V_2ACC_S) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n SIN n\n MF 80\n SIN n+1\n .set n, n+2\n .endr\n" BODY_END;
  } else if constexpr (V == V_2ACC_SC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n SIN n\n MF 80\n SIN n+1\n CVT n+2\n .set n, n+3\n .endr\n" BODY_END;
  } else if constexpr (V == V_2ACC_SSC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n SIN n\n SIN n+1\n CVT n+2\n MF 80\n SIN n+3\n SIN n+4\n CVT n+5\n .set n, n+6\n .endr\n" BODY_END;
  } else if constexpr (V == V_2ACC_5) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 68\n MF 64\n SIN n\n SIN n+1\n CVT n+2\n SIN n+3\n CVT n+4\n MF 80\n SIN n+5\n SIN n+6\n CVT n+7\n SIN n+8\n CVT n+9\n .set n, n+10\n .endr\n" BODY_END;
  } else if constexpr (V == V_4ACC_SC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 34\n MF 64\n SIN n\n MF 80\n SIN n+1\n CVT n+2\n MF 96\n SIN n+3\n MF 112\n SIN n+4\n CVT n+5\n .set n, n+6\n .endr\n" BODY_END;
  } else if constexpr (V == V_4ACC_SSC) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 34\n MF 64\n SIN n\n SIN n+1\n CVT n+2\n MF 80\n SIN n+3\n SIN n+4\n CVT n+5\n MF 96\n SIN n+6\n SIN n+7\n CVT n+8\n MF 112\n SIN n+9\n SIN n+10\n CVT n+11\n .set n, n+12\n .endr\n" BODY_END;
  } else if constexpr (V == V_4ACC_5) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .rept 34\n"
      " MF 64\n SIN n\n SIN n+1\n CVT n+2\n SIN n+3\n CVT n+4\n MF 80\n SIN n+5\n SIN n+6\n CVT n+7\n SIN n+8\n CVT n+9\n"
      " MF 96\n SIN n+10\n SIN n+11\n CVT n+12\n SIN n+13\n CVT n+14\n MF 112\n SIN n+15\n SIN n+16\n CVT n+17\n SIN n+18\n CVT n+19\n .set n, n+20\n .endr\n" BODY_END;
  } else if constexpr (V == V_TILE1) {
    // 8 tiles: tile on X (64 / 80 alternating) with the epilogue of the other accumulator: MFMA i is followed by sin i (i < 16) and,
    // after every second sin, the cvt_pk of that pair
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 80\n .rept 8\n .set k, 0\n"
      " MF x\n .rept 8\n MF x\n SINA y, 2*k\n CVTP k\n MF x\n SINA y, 2*k+1\n .set k, k+1\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n" BODY_END;
  } else if constexpr (V == V_TILE2) {
    // 4 tile pairs: chains X0 (B0), X1 (B1); epilogue reads Y0 then Y1: per k-step  M(X0) sin sin cvt  M(X1) sin sin cvt  (k < 8) and
    // 9 more MFMA pairs of the same shape reading Y1
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 96\n .rept 4\n .set k, 0\n"
      " MFB x, 4\n MFB x+16, 8\n"
      " .rept 8\n MFB x, 4\n SINA y, 2*k\n CVTP k\n MFB x+16, 8\n SINA y, 2*k+1\n .set k, k+1\n .endr\n"
      " .set k, 0\n .rept 8\n MFB x, 4\n SINA y+16, 2*k\n CVTP k\n MFB x+16, 8\n SINA y+16, 2*k+1\n .set k, k+1\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n" BODY_END;
  } else if constexpr (V == V_SERIAL1) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 80\n .rept 8\n"
      " .rept 17\n MF x\n .endr\n s_nop 15\n .set k, 0\n .rept 8\n SINA x, 2*k\n CVTP k\n SINA x, 2*k+1\n .set k, k+1\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n" BODY_END;
  } else if constexpr (V == V_TILE1_LDS) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 80\n LDA 0\n LDA 1\n .set n, 0\n .rept 8\n .set k, 0\n"
      " LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n .set n, n+1\n"
      " .rept 8\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n SINA y, 2*k\n CVTP k\n LDA n+3\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*((n+1)%%4)\n SINA y, 2*k+1\n"
      " .set k, k+1\n .set n, n+2\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n s_waitcnt lgkmcnt(0)\n" BODY_END;
  } else if constexpr (V == V_TILE2_LDS) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 96\n LDA 0\n LDA 1\n .set n, 0\n .rept 4\n .set k, 0\n"
      " LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n MFL x+16, 8, 128+4*(n%%4)\n .set n, n+1\n"
      " .rept 8\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n SINA y, 2*k\n CVTP k\n MFL x+16, 8, 128+4*(n%%4)\n SINA y, 2*k+1\n .set k, k+1\n .set n, n+1\n .endr\n"
      " .set k, 0\n .rept 8\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n SINA y+16, 2*k\n CVTP k\n MFL x+16, 8, 128+4*(n%%4)\n SINA y+16, 2*k+1\n .set k, k+1\n .set n, n+1\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n s_waitcnt lgkmcnt(0)\n" BODY_END;
  } else if constexpr (V == V_SERIAL1_LDS) {
    BODY_BEGIN "v_mov_b32 v144, %0\n .set x, 64\n .set y, 80\n LDA 0\n LDA 1\n .set n, 0\n .rept 8\n"
      " .rept 17\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL x, 4, 128+4*(n%%4)\n .set n, n+1\n .endr\n"
      " s_nop 15\n .set k, 0\n .rept 8\n SINA x, 2*k\n CVTP k\n SINA x, 2*k+1\n .set k, k+1\n .endr\n"
      " .set t, x\n .set x, y\n .set y, t\n .endr\n s_waitcnt lgkmcnt(0)\n" BODY_END;
  } else if constexpr (V == V_MFMA_LDS1) {
    BODY_BEGIN "v_mov_b32 v144, %0\n LDA 0\n LDA 1\n .set n, 0\n .rept 136\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL 64, 4, 128+4*(n%%4)\n .set n, n+1\n .endr\n s_waitcnt lgkmcnt(0)\n" BODY_END;
  } else if constexpr (V == V_MFMA_LDS2) {
    BODY_BEGIN "v_mov_b32 v144, %0\n LDA 0\n LDA 1\n .set n, 0\n .rept 68\n LDA n+2\n s_waitcnt lgkmcnt(2)\n MFL 64, 4, 128+4*(n%%4)\n MFL 80, 8, 128+4*(n%%4)\n .set n, n+1\n .endr\n s_waitcnt lgkmcnt(0)\n" BODY_END;
  }
}

template <int V>
__global__ void __launch_bounds__(512) probe_kernel(uint64_t* cycles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: 16 KiB of fragments per wave (lane-linear 1-KiB pieces) so the reads are real and conflict-free
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (int)(blockDim.x / 64) * 1024; i += blockDim.x)
    reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  const uint32_t lds_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + wave * 16384 + lane * 16;
  // operands: small finite numbers so the clocks see real data
  asm volatile(
      "v_mov_b32 v0, 0x3c003c00\n v_mov_b32 v1, 0x3c003c00\n v_mov_b32 v2, 0x3c003c00\n v_mov_b32 v3, 0x3c003c00\n"
      "v_mov_b32 v4, 0x3b803b80\n v_mov_b32 v5, 0x3b803b80\n v_mov_b32 v6, 0x3b803b80\n v_mov_b32 v7, 0x3b803b80\n"
      "v_mov_b32 v8, 0x3b803b80\n v_mov_b32 v9, 0x3b803b80\n v_mov_b32 v10, 0x3b803b80\n v_mov_b32 v11, 0x3b803b80\n"
      ".set r, 0\n .rept 16\n v_mov_b32 v[32+r], 0x3e000000\n v_mov_b32 v[16+r], 0\n .set r, r+1\n .endr\n"
      ".set r, 0\n .rept 64\n v_mov_b32 v[64+r], 0\n .set r, r+1\n .endr\n" ::: CLOB);
  body<V>(lds_addr);  // warm the instruction cache
  __builtin_amdgcn_s_barrier();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) body<V>(lds_addr);
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  // keep the results alive
  float sink;
  asm volatile("v_add_f32 %0, v64, v80\n v_add_f32 %0, %0, v96\n v_add_f32 %0, %0, v112\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v32" : "=v"(sink)::CLOB);
  if (sink == 12345.678f) cycles[0] = 0;
}

template <int V>
static void run_variant(uint64_t* d_cycles, FILE* out) {
  const int iters = 40, grid = 256;
  for (int threads : {256, 512}) {
    const size_t lds = 136 * 1024;  // one workgroup per CU
    CHECK(hipFuncSetAttribute((const void*)probe_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe_kernel<V>, dim3(grid), dim3(threads), lds, 0, d_cycles, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_kernel<V>, dim3(grid), dim3(threads), lds, 0, d_cycles, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    static uint64_t h[256 * 8];
    CHECK(hipMemcpy(h, d_cycles, sizeof(h), hipMemcpyDeviceToHost));
    const int nw = threads / 64;
    uint64_t mx = 0, mn = ~0ull;
    double sum = 0;
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < nw; ++w) {
        const uint64_t c = h[b * 8 + w];
        if (c > mx) mx = c;
        if (c < mn) mn = c;
        sum += (double)c;
      }
    const double slots = (double)kSlots[V] * iters;
    // wall: every SIMD runs (threads / 256) waves x slots MFMA slots
    const double us = best * 1e3;
    const double wall_cyc_per_slot_per_simd = us * 1e-6 * 2.4e9 / (slots * (threads / 256));  // at the 2.4 GHz maximum clock (upper bound)
    fprintf(out, "%-52s %d wave/SIMD  cycles/slot/wave: mean %6.1f  min %6.1f  max %6.1f | per SIMD slot %6.1f | wall %8.1f us (<= %5.1f cyc/slot/SIMD at 2.4 GHz)\n",
            kNames[V], threads / 256, sum / (grid * nw) / slots, (double)mn / slots, (double)mx / slots, sum / (grid * nw) / slots / (threads / 256), us,
            wall_cyc_per_slot_per_simd);
    fflush(out);
  }
}

template <int V>
static void run_all(uint64_t* d, FILE* out) {
  run_variant<V>(d, out);
  if constexpr (V + 1 < V_COUNT) run_all<V + 1>(d, out);
}

int main() {
  uint64_t* d;
  CHECK(hipMalloc(&d, 256 * 8 * sizeof(uint64_t)));
  CHECK(hipMemset(d, 0, 256 * 8 * sizeof(uint64_t)));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("# probe_coissue on %s (%d CUs, clock %d kHz); 256 workgroups, 136 MFMA slots x 40 iterations per wave; ideal = 32 cycles per MFMA per SIMD\n",
         p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run_all<0>(d, stdout);
  return 0;
}
