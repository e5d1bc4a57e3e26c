// Weight gradients of the fused Sat-NeRF MLP from the 8-bit training workspaces (SR_FMT8, mlp_layout.h) for gfx950:
//   dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Same job as wgrad.hip (autograd's grad_weight = grad_output^T @ input / grad_bias of every nn.Linear in SatNeRF,
// models/satnerf.py:104-153) and the same contraction machinery -- a 16-wave workgroup owns one job block of up to 256 x 256
// (+ its aux columns) over a contiguous slice of 32-point tiles, operands are staged point-major in LDS and read back transposed
// with ds_read_b64_tr_b16, fp32 partial blocks are reduced by sr_unpack_grads / sr_grad_tail -- but the operands arrive as ONE
// BYTE per value: a 1-KiB double fragment (DF) carries what two bf16 fragments carried, so a block moves 17 KiB per tile
// instead of 34 and the kernel's HBM floor halves.  Each wave fetches one DF per tile by LDS-DMA into the place of the SECOND of
// the two fragments it expands to, and decodes it in place one tile ahead of the MFMAs (PHASE8 -> bf16 sin, MX8 -> bf16 value
// with the lane's shared scale); the two single bf16 fragments of the format (d_sigma_pre, d_head) and the aux fragments are
// fetched as they are.  What a block loads is spelled out by a per-block load table built on the host (packing.wgrad8_loads).
//
// Pipeline per 32-point tile i (ring of 4 LDS slots): issue the DMA of tile i+3 -> k-step 0 of tile i -> decode tile i+1 ->
// k-step 1 -> wait for this wave's DMA of tile i+2 -> rendezvous.  The DMA of a tile is complete and visible to every wave one
// rendezvous before its decode, so a wave may read scale bytes another wave fetched.
#include <stdlib.h>

#ifndef SR_W8_ABL
#define SR_W8_ABL 0  // 2..8: timing ablations (wrong results), tools/ab_wgrad_abl.sh
#endif
#include "codec8.h"
#include "common.h"
#include "mlp_device.h"
#include "mlp_layout.h"

namespace sr {

struct Wgrad8Params {
  const uint4* dpre;
  const uint4* acts;
  const int* blocks;  // planned job table, kWgTableInts ints per block (row / column fragment counts, n_slices, first_slice)
  const int* loads;   // kWg8LoadInts ints per block
  float* partial;
  long n_tiles;
  int n_blocks;
  int ak;             // activation units per tile
  int auxs;
  int dk;  // units per tile of the dpre workspace
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// load table (host: packing.wgrad8_loads).  ints 0..15: the primary load of wave w, 0 = none, else
//   bits 0-1 source (1 dpre, 2 acts) | 2-3 codec (0 RAW16, 1 PHASE8, 2 MX8) | 4-11 unit within the tile | 12-17 operand fragment
//   (0..15 rows, 16..31 columns; a DF expands into fragment and fragment + 1) | 18-19 scale area | 20-23 byte within the lane's 16 B
// ints 16..18: the scale unit fetched into scale area 0..2 (bits 0-1 source, 4-11 unit; 0 = none), by waves 2..4.
// ints 20..99: the duty table of the 4-wave kernel (wgrad9.hip, packing.wgrad9_duties); ints 100..107: the exponent groups of its row pairs, int 108: its quadrant mask.
constexpr int kWg8LoadInts = 113;
enum { kSrcDpre = 1, kSrcActs = 2, kRaw16 = 0, kPhase8 = 1, kMx8 = 2 };

constexpr int kFragStride8 = 1088;  // as wgrad.hip: 1-KiB lane-linear fragment image + 64 B so that transposed reads spread over the banks
constexpr int kOperandFrags = 34;   // 16 row + 16 column + 2 aux
constexpr int kScaleAreas = 3;
constexpr int kSlot8Bytes = kOperandFrags * kFragStride8 + kScaleAreas * 1024;
constexpr int kSlots8 = 4;

template <int N>
__device__ __forceinline__ void wait_vm8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ void __launch_bounds__(1024) wgrad8_kernel(const Wgrad8Params prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* d = prm.blocks;
  int blk = 0;
  for (; blk + 1 < prm.n_blocks && (int)blockIdx.x >= d[kWgFirstSlice] + d[kWgSlices]; ++blk) d += kWgTableInts;
  const int nr = d[1] + d[3], nc = d[5] + d[7];
  const int* ld_tab = prm.loads + blk * kWg8LoadInts;
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = (long)((int)blockIdx.x - d[kWgFirstSlice]) * tiles_per_split;
  long t_end = t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  const int nt = t_end > t_begin ? (int)(t_end - t_begin) : 0;

  // ---- this wave's loads --------------------------------------------------------------------------------------------------
  const int prim = __builtin_amdgcn_readfirstlane(ld_tab[wave]);
  const bool has_prim = prim != 0;
  const int p_codec = (prim >> 2) & 3, p_unit = (prim >> 4) & 255, p_dst = (prim >> 12) & 63, p_area = (prim >> 18) & 3, p_byte = (prim >> 20) & 15;
  int sec_src = 0, sec_unit = 0, sec_off = 0;  // secondary: aux fragment (waves < auxs) or a scale unit (waves 2..4)
  if (wave < prm.auxs) {
    sec_src = kSrcActs, sec_unit = wave, sec_off = (32 + wave) * kFragStride8;
  } else if (wave >= 2 && wave < 2 + kScaleAreas) {
    const int sd = __builtin_amdgcn_readfirstlane(ld_tab[16 + wave - 2]);
    if (sd != 0) sec_src = sd & 3, sec_unit = (sd >> 4) & 255, sec_off = kOperandFrags * kFragStride8 + (wave - 2) * 1024;
  }
  const bool has_sec = sec_src != 0, sec_is_aux = wave < prm.auxs;
  const int n_ld = (int)has_prim + (int)has_sec;  // 0..2 DMA instructions per tile
  const int src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);  // rotated image: position `lane` holds this source lane's 16 B
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const long p_stride = (prim & 3) == kSrcDpre ? prm.dk : prm.ak, s_stride = sec_src == kSrcDpre ? prm.dk : prm.ak;
  // DMA addresses = wave-uniform base (SGPR pair, advanced per tile by scalar arithmetic) + a fixed per-lane byte offset
  const uint4* p_base = ((prim & 3) == kSrcDpre ? prm.dpre : prm.acts) + p_unit * 64;
  const uint4* s_base = (sec_src == kSrcDpre ? prm.dpre : prm.acts) + sec_unit * 64;
  const uint32_t p_voff = (uint32_t)src_unit * 16u, s_voff = (uint32_t)(sec_is_aux ? src_unit : lane) * 16u;
  const int p_off = (p_codec == kRaw16 ? p_dst : p_dst + 1) * kFragStride8;  // a DF lands where its second fragment will be
  auto issue = [&](long tile, int slot) {
#if SR_W8_ABL == 5  // timing experiment: no LDS-DMA (operands are whatever the LDS holds)
    return;
#endif
    const uint32_t base = ring + slot * kSlot8Bytes;
    if (has_prim) glds16_s(reinterpret_cast<const char*>(p_base + tile * p_stride * 64), p_voff, base + p_off);
    if (has_sec) glds16_s(reinterpret_cast<const char*>(s_base + tile * s_stride * 64), s_voff, base + sec_off);
  };
  auto wait_outstanding = [&](int tiles_in_flight) {  // all but the newest `tiles_in_flight` tiles of this wave have landed
    switch (tiles_in_flight * n_ld) {
      case 4: wait_vm8<4>(); break;
      case 3: wait_vm8<3>(); break;
      case 2: wait_vm8<2>(); break;
      case 1: wait_vm8<1>(); break;
      default: wait_vm8<0>(); break;
    }
  };
  auto decode = [&](int slot) {
#if SR_W8_ABL == 3 || SR_W8_ABL == 8  // timing experiment: no decode at all (8: LDS-DMA and rendezvous only)
    return;
#endif
    if (!has_prim || p_codec == kRaw16) return;
    char* sl = lds + slot * kSlot8Bytes;
    uint4* raw = reinterpret_cast<uint4*>(sl + (p_dst + 1) * kFragStride8 + lane * 16);
    const uint4 v = *raw;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[8];
    if (p_codec == kPhase8) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        o[q] = pack_bf16x2(__builtin_amdgcn_sinf(phase8_rev(w[q >> 1], 2 * (q & 1))), __builtin_amdgcn_sinf(phase8_rev(w[q >> 1], 2 * (q & 1) + 1)));
    } else {
      const uint32_t e = *reinterpret_cast<const uint8_t*>(sl + kOperandFrags * kFragStride8 + p_area * 1024 + src_unit * 16 + p_byte);
      const float s = mx8_scale(e), bias = -128.0f * s;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = pack_bf16x2(mx8_value(w[q >> 1], 2 * (q & 1), s, bias), mx8_value(w[q >> 1], 2 * (q & 1) + 1, s, bias));
    }
#if SR_W8_ABL == 2  // timing experiment: decode arithmetic kept, LDS writes dropped (wrong results)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(o[i]));
#else
    *reinterpret_cast<uint4*>(sl + p_dst * kFragStride8 + lane * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    *raw = make_uint4(o[4], o[5], o[6], o[7]);
#endif
  };

  // ---- MFMA side: identical to wgrad.hip --------------------------------------------------------------------------------
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  int rd_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int point = 16 * ks + 8 * hh + 4 * rd + m;
      rd_off[ks][rd] = rh * kFragStride8 + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8;
    }
  auto operand = [&](const char* buf, int frag_pair, int ks) {
#if SR_W8_ABL == 7  // timing experiment: no transposed operand reads (MFMAs on register garbage)
    return make_uint4((uint32_t)frag_pair, (uint32_t)ks, 0x3f803f80u, 0x3f803f80u);
#endif
    const char* p = buf + frag_pair * 2 * kFragStride8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + rd_off[ks][0]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + rd_off[ks][1]));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
  };
  auto mma = [](const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  const int wr = wave >> 2, wc = wave & 3;
  auto clampi = [](int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; };
  const int n_rt = clampi((16 * nr - 64 * wr + 31) / 32, 2), n_ct = clampi((16 * nc - 64 * wc + 31) / 32, 2);
  const bool aux_on = wc == 0 ? n_rt >= 1 : wc == 1 ? n_rt == 2 : false;  // aux columns of row tile 2*wr (wc 0) / 2*wr+1 (wc 1)
  const bool full = n_rt == 2 && n_ct == 2;
  auto rendezvous = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if SR_W8_ABL != 6  // (6: timing experiment without the workgroup barrier)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
  };

  auto run = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
    f32x16 acc[2][2] = {}, acc_aux = {};
    auto kstep = [&](const char* b, int ks) {
#if SR_W8_ABL == 8
      return;
#endif
      if constexpr (kFull) {
#if SR_W8_ABL == 4  // timing experiment: reads only, no MFMAs
        {
          const uint4 ra0 = operand(b, 2 * wr, ks), ra1 = operand(b, 2 * wr + 1, ks);
          const uint4 rb0 = operand(b, 8 + 2 * wc, ks), rb1 = operand(b, 8 + 2 * wc + 1, ks);
          asm volatile("" ::"v"(ra0.x), "v"(ra1.x), "v"(rb0.x), "v"(rb1.x), "v"(ra0.w), "v"(ra1.w), "v"(rb0.w), "v"(rb1.w));
          return;
        }
#endif
        const uint4 a0 = operand(b, 2 * wr, ks), a1 = operand(b, 2 * wr + 1, ks);
        const uint4 b0 = operand(b, 8 + 2 * wc, ks), b1 = operand(b, 8 + 2 * wc + 1, ks);
        acc[0][0] = mma(a0, b0, acc[0][0]), acc[1][0] = mma(a1, b0, acc[1][0]);
        acc[0][1] = mma(a0, b1, acc[0][1]), acc[1][1] = mma(a1, b1, acc[1][1]);
        if (aux_on) acc_aux = mma(wc ? a1 : a0, operand(b, 16, ks), acc_aux);
      } else {
        if (n_rt == 0) return;
        const uint4 a0 = operand(b, 2 * wr, ks);
        uint4 a1 = a0;
        if (n_rt > 1) a1 = operand(b, 2 * wr + 1, ks);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (ct < n_ct) {
            const uint4 bc = operand(b, 8 + 2 * wc + ct, ks);
            acc[0][ct] = mma(a0, bc, acc[0][ct]);
            if (n_rt > 1) acc[1][ct] = mma(a1, bc, acc[1][ct]);
          }
        }
        if (aux_on) acc_aux = mma(wc ? a1 : a0, operand(b, 16, ks), acc_aux);
      }
    };
    // prologue: tiles 0..2 in flight; tile 0 landed -> visible -> decoded; tile 1 landed -> visible
    const int pre = nt < kSlots8 - 1 ? nt : kSlots8 - 1;
    for (int i = 0; i < pre; ++i) issue(t_begin + i, i);
    if (nt > 0) {
      wait_outstanding(pre - 1);
      rendezvous();
      decode(0);
      wait_outstanding(pre > 2 ? pre - 2 : 0);
      rendezvous();
    }
    for (int i = 0; i < nt; ++i) {
      const char* cur = lds + (i & (kSlots8 - 1)) * kSlot8Bytes;
      const bool more = i + kSlots8 - 1 < nt;
      if (more) issue(t_begin + i + kSlots8 - 1, (i + kSlots8 - 1) & (kSlots8 - 1));
      kstep(cur, 0);
      if (i + 1 < nt) decode((i + 1) & (kSlots8 - 1));
      kstep(cur, 1);
      wait_outstanding(more ? 1 : 0);  // tile i+2 (if any) has landed; only tile i+3 may still be in flight
      rendezvous();
    }
    float* out = prm.partial + (long)blockIdx.x * kWgBlockFloats;
    const int n_rows = 16 * nr, n_cols = 16 * nc;
    if (n_rt > 0 && n_ct > 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int row = 64 * wr + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh;
            const int col = 64 * wc + 32 * ct + (lane & 31);
            if (row < n_rows && col < n_cols) out[row * 256 + col] = acc[rt][ct][g];
          }
    }
    if (aux_on) {
      float* oa = out + 256 * 256;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 64 * wr + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
        if (row < n_rows) oa[row * 32 + (lane & 31)] = acc_aux[g];
      }
    }
  };
  if (full) run(std::true_type{});
  else run(std::false_type{});
}

}  // namespace sr

namespace sr {
int launch_wgrad8f(const uint4* dpre, const uint4* acts, const int* blocks, const int* loads, float* partial, long n_tiles, int n_blocks,
                   int ak, int auxs, int dk, int n_slices, hipStream_t st);  // wgrad8f.hip
int launch_wgrad9(const uint4* dpre, const uint4* acts, const uint4* emax, const int* blocks, const int* loads, float* partial, long n_tiles,
                  int n_blocks, int ak, int dk, int load_ints, int n_slices, int span, hipStream_t st);  // wgrad9.hip
bool wgrad9_fits(long n_tiles, int ak, int dk);
}
using namespace sr;

extern "C" int sr_satnerf_wgrad8(int feat, int tau, int64_t n_points, const uint16_t* dpre, int64_t dpre_elems, const uint16_t* acts,
                                 const int32_t* blocks, const int32_t* loads, int n_blocks, int n_slices, int plan_span, float* partial, void* stream) {
  SR_REQUIRE(feat == 256 || feat == 512, "sr_satnerf_wgrad8: feat=%d unsupported (256, 512)", feat);
  SR_REQUIRE(dpre && acts && blocks && loads && partial, "sr_satnerf_wgrad8: null pointer argument");
  // the default kernel reads the table of exponent maxima BEHIND the last tile (a buffer of tiles x sr_dpre_elems_per_tile is too short)
  SR_REQUIRE(dpre_elems >= sr_dpre_workspace_elems(n_points, feat, SR_FMT8), "sr_satnerf_wgrad8: dpre holds %lld elements, sr_dpre_workspace_elems asks for %lld",
             (long long)dpre_elems, (long long)sr_dpre_workspace_elems(n_points, feat, SR_FMT8));
  SR_REQUIRE(n_blocks >= 1 && n_slices >= n_blocks && plan_span >= 0, "sr_satnerf_wgrad8: bad plan (%d blocks, %d slices, span %d): run sr_wgrad_plan first",
             n_blocks, n_slices, plan_span);
  Wgrad8Params p;
  p.dpre = (const uint4*)dpre, p.acts = (const uint4*)acts, p.blocks = blocks, p.loads = loads, p.partial = partial;
  p.n_tiles = (n_points + 31) / 32;
  p.n_blocks = n_blocks;
  p.auxs = aux_steps(tau);
  p.ak = (int)(sr_act_elems_per_tile(feat, SR_FMT8) / 512) - (2 - p.auxs);  // the size query assumes the 2-step aux layout
  p.dk = (int)(sr_dpre_elems_per_tile(feat, SR_FMT8) / 512);
  // default: the 4-wave kernel with 128 x 128 register tiles and the generated slice loop (wgrad9.hip), either width.  SATNERF_WGRAD_V1=1
  // keeps the r02 kernel below (A/B; it also takes the workspaces beyond the 32-bit offsets of the new one); SATNERF_WGRAD_V2=1 the r03
  // fat-wave experiment (wgrad8f.hip, width 256).
  static const bool v1 = [] { const char* e = getenv("SATNERF_WGRAD_V1"); return e && e[0] == '1'; }();
  static const bool v2 = [] { const char* e = getenv("SATNERF_WGRAD_V2"); return e && e[0] == '1'; }();
  if (!v1 && !v2 && wgrad9_fits(p.n_tiles, p.ak, p.dk))
    return launch_wgrad9(p.dpre, p.acts, p.dpre + sr::ws_tiles(n_points) * p.dk * 64 /* the exponent maxima behind the last tile */, blocks, loads,
                         partial, p.n_tiles, n_blocks, p.ak, p.dk, kWg8LoadInts, n_slices, plan_span, (hipStream_t)stream);
  SR_REQUIRE(plan_span == 0, "sr_satnerf_wgrad8: a stream-K plan (span %d) needs the 4-wave kernel: plan with SATNERF_WGRAD_STREAMK=0 for the r02 / r03 kernels", plan_span);
  if (feat == 256 && v2)
    return launch_wgrad8f(p.dpre, p.acts, blocks, loads, partial, p.n_tiles, n_blocks, p.ak, p.auxs, p.dk, n_slices, (hipStream_t)stream);
  const size_t lds = (size_t)kSlots8 * kSlot8Bytes;
  if (!ensure_dynamic_lds((const void*)wgrad8_kernel, lds)) return 1;
  hipLaunchKernelGGL(wgrad8_kernel, dim3(n_slices), dim3(1024), lds, (hipStream_t)stream, p);
  return check_launch("wgrad8_kernel");
}

extern "C" int sr_wgrad8_load_ints(void) { return kWg8LoadInts; }
