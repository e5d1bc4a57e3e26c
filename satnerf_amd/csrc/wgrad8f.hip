// Weight gradients from the 8-bit training workspaces, second generation ("fat waves"): same job as wgrad8.hip --
//   dW[row][col] = sum over sample points of dpre[row] * act[col]    (autograd's grad_weight / grad_bias of every nn.Linear in
//   SatNeRF, models/satnerf.py:104-153), same job table, load table, LDS slot layout, split-K partial blocks --
// but a workgroup is 8 waves of <= 256 VGPRs (two per SIMD) instead of 16 thin ones, and for the full 256 x 256 blocks (10 of the 14
// at width 256: 85 % of the MFMAs) the work of one 32-point tile is ONE hand-placed instruction stream per wave
// (csrc/gen/wgrad_tile.py -> wgrad8f_tile_{p,m}.inc): 18 MFMAs (4 x 2 output tiles + one aux tile, two k-steps), their 32
// transposed operand reads software-pipelined with counted lgkmcnt waits, and the ~100 VALU instructions that decode the NEXT tile's
// two double fragments (PHASE8 -> sin, MX8 -> value) spread over the MFMA gaps.  r02's kernel ran matrix work and decode one after
// the other (MFMA busy 30 %, VALU 36 %: profiles/r02_train_pmc.csv); profiles/r03_coissue.txt shows they overlap when hand-placed.
// The other blocks (narrow head layers) run the r02 contraction code, each wave playing two of the old kernel's 16 roles.
//
// Pipeline per tile i (ring of 4 LDS slots, unchanged): issue the DMA of tile i+3 -> [k-step 0, k-step 1 of tile i | decode of tile
// i+1] -> wait for this wave's DMA of tile i+2 -> rendezvous.
#include <stdlib.h>

#include "codec8.h"
#include "common.h"
#include "mlp_device.h"
#include "mlp_layout.h"

namespace sr {

struct Wgrad8Params {  // (as wgrad8.hip)
  const uint4* dpre;
  const uint4* acts;
  const int* blocks;
  const int* loads;
  float* partial;
  long n_tiles;
  int n_blocks;
  int ak;
  int auxs;
  int dk;
  long long* dbg;  // SR_W8_TIMING builds: s_memtime stamps of workgroup 0 ([wave][iteration < 32][8])
};

typedef short s16x4f __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

namespace {
constexpr int kLoadInts = 113;  // = wgrad8.hip kWg8LoadInts (ints 0..19 are this kernel's)
enum { kSrcDpreF = 1, kSrcActsF = 2, kRaw16F = 0, kPhase8F = 1, kMx8F = 2 };
constexpr int kFrag = 1088, kPair = 2 * kFrag, kOperandFragsF = 34, kScaleAreasF = 3;
constexpr int kSlotBytes = kOperandFragsF * kFrag + kScaleAreasF * 1024;
constexpr int kSlotsF = 4;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

__global__ void __launch_bounds__(512) wgrad8f_kernel(const Wgrad8Params prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7: plays the old kernel's waves `wave` and `wave + 8`
  const int* d = prm.blocks;
  int blk = 0;
  for (; blk + 1 < prm.n_blocks && (int)blockIdx.x >= d[kWgFirstSlice] + d[kWgSlices]; ++blk) d += kWgTableInts;
  const int nr = d[1] + d[3], nc = d[5] + d[7];
  const int* ld_tab = prm.loads + blk * kLoadInts;
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = (long)((int)blockIdx.x - d[kWgFirstSlice]) * tiles_per_split;
  long t_end = t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  const int nt = t_end > t_begin ? (int)(t_end - t_begin) : 0;

  // ---- this wave's loads: the primaries of both roles + role 0's secondary (aux fragment or scale unit) ----------------------------
  int prim[2], p_codec[2], p_dst[2], p_area[2], p_byte[2];
  const uint4* p_base[2];
  long p_stride[2];
  bool has_prim[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    prim[r] = __builtin_amdgcn_readfirstlane(ld_tab[wave + 8 * r]);
    has_prim[r] = prim[r] != 0;
    p_codec[r] = (prim[r] >> 2) & 3, p_dst[r] = (prim[r] >> 12) & 63, p_area[r] = (prim[r] >> 18) & 3, p_byte[r] = (prim[r] >> 20) & 15;
    p_stride[r] = (prim[r] & 3) == kSrcDpreF ? prm.dk : prm.ak;
    p_base[r] = ((prim[r] & 3) == kSrcDpreF ? prm.dpre : prm.acts) + ((prim[r] >> 4) & 255) * 64;
  }
  int sec_src = 0, sec_unit = 0, sec_off = 0;
  if (wave < prm.auxs) {
    sec_src = kSrcActsF, sec_unit = wave, sec_off = (32 + wave) * kFrag;
  } else if (wave >= 2 && wave < 2 + kScaleAreasF) {
    const int sd = __builtin_amdgcn_readfirstlane(ld_tab[16 + wave - 2]);
    if (sd != 0) sec_src = sd & 3, sec_unit = (sd >> 4) & 255, sec_off = kOperandFragsF * kFrag + (wave - 2) * 1024;
  }
  const bool has_sec = sec_src != 0, sec_is_aux = wave < prm.auxs;
  const int n_ld = (int)has_prim[0] + (int)has_prim[1] + (int)has_sec;  // 0..3 DMA instructions per tile
  const int src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);  // rotated image: position `lane` holds this source lane's 16 B
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const long s_stride = sec_src == kSrcDpreF ? prm.dk : prm.ak;
  const uint4* s_base = (sec_src == kSrcDpreF ? prm.dpre : prm.acts) + sec_unit * 64;
  const uint32_t p_voff = (uint32_t)src_unit * 16u, s_voff = (uint32_t)(sec_is_aux ? src_unit : lane) * 16u;
  int p_off[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) p_off[r] = (p_codec[r] == kRaw16F ? p_dst[r] : p_dst[r] + 1) * kFrag;  // a DF lands where its second fragment will be
  auto issue = [&](long tile, int slot) {
    const uint32_t base = ring + slot * kSlotBytes;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (has_prim[r]) glds16_s(reinterpret_cast<const char*>(p_base[r] + tile * p_stride[r] * 64), p_voff, base + p_off[r]);
    if (has_sec) glds16_s(reinterpret_cast<const char*>(s_base + tile * s_stride * 64), s_voff, base + sec_off);
  };
  auto wait_outstanding = [&](int tiles_in_flight) {  // all but the newest `tiles_in_flight` tiles of this wave have landed
    switch (tiles_in_flight * n_ld) {
      case 6: wait_vm<6>(); break;
      case 5: wait_vm<5>(); break;
      case 4: wait_vm<4>(); break;
      case 3: wait_vm<3>(); break;
      case 2: wait_vm<2>(); break;
      case 1: wait_vm<1>(); break;
      default: wait_vm<0>(); break;
    }
  };
  auto decode_one = [&](int slot, int r) {
    if (!has_prim[r] || p_codec[r] == kRaw16F) return;
    char* sl = lds + slot * kSlotBytes;
    uint4* raw = reinterpret_cast<uint4*>(sl + (p_dst[r] + 1) * kFrag + lane * 16);
    const uint4 v = *raw;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[8];
    if (p_codec[r] == kPhase8F) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        o[q] = pack_bf16x2(__builtin_amdgcn_sinf(phase8_rev(w[q >> 1], 2 * (q & 1))), __builtin_amdgcn_sinf(phase8_rev(w[q >> 1], 2 * (q & 1) + 1)));
    } else {
      const uint32_t e = *reinterpret_cast<const uint8_t*>(sl + kOperandFragsF * kFrag + p_area[r] * 1024 + src_unit * 16 + p_byte[r]);
      const float s = mx8_scale(e), bias = -128.0f * s;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = pack_bf16x2(mx8_value(w[q >> 1], 2 * (q & 1), s, bias), mx8_value(w[q >> 1], 2 * (q & 1) + 1, s, bias));
    }
    *reinterpret_cast<uint4*>(sl + p_dst[r] * kFrag + lane * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    *raw = make_uint4(o[4], o[5], o[6], o[7]);
  };
  auto decode = [&](int slot) { decode_one(slot, 0), decode_one(slot, 1); };

  // ---- transposed operand reads (as wgrad8.hip) -------------------------------------------------------------------------------------
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  int rd_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int point = 16 * ks + 8 * hh + 4 * rd + m;
      rd_off[ks][rd] = rh * kFrag + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8;
    }
  auto rendezvous = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // prologue of both paths: tiles 0..2 in flight; tile 0 landed -> visible -> decoded; tile 1 landed -> visible
  const int pre = nt < kSlotsF - 1 ? nt : kSlotsF - 1;
  for (int i = 0; i < pre; ++i) issue(t_begin + i, i);
  if (nt > 0) {
    wait_outstanding(pre - 1);
    rendezvous();
    decode(0);
    wait_outstanding(pre > 2 ? pre - 2 : 0);
    rendezvous();
  }
  float* out = prm.partial + (long)blockIdx.x * kWgBlockFloats;
  const int n_rows = 16 * nr, n_cols = 16 * nc;
  const bool full = nr == 16 && nc == 16 && has_prim[0] && has_prim[1] && p_codec[0] != kRaw16F && p_codec[1] != kRaw16F;

  if (full) {
    // ================= full 256 x 256 block: the generated per-tile stream =================================================
    // fat wave (r2 = wave >> 2, wc = wave & 3): row pairs {2 r2, 2 r2 + 1, 2 r2 + 4, 2 r2 + 5} x column pairs {8 + 2 wc, + 1}; its aux
    // tile = the aux columns of row pair list[wc]
    const int r2 = wave >> 2, wc = wave & 3;
    const int row_pair[4] = {2 * r2, 2 * r2 + 1, 2 * r2 + 4, 2 * r2 + 5};
    const int xpair = wc == 0 ? row_pair[0] : wc == 1 ? row_pair[1] : wc == 2 ? row_pair[2] : row_pair[3];
    // the row double fragment is the MX8 one from dpre, the other one the column double fragment
    const int rr = (prim[0] & 3) == kSrcDpreF ? 0 : 1, cr = 1 - rr;
    const uint32_t aoff = __builtin_amdgcn_readfirstlane(2 * r2 * kPair), boff = __builtin_amdgcn_readfirstlane((8 + 2 * wc) * kPair);
    const uint32_t xoff = __builtin_amdgcn_readfirstlane(xpair * kPair);
    const uint32_t rraw = __builtin_amdgcn_readfirstlane(p_off[rr]), craw = __builtin_amdgcn_readfirstlane(p_off[cr]);
    const uint32_t rsc = __builtin_amdgcn_readfirstlane(kOperandFragsF * kFrag + p_area[rr] * 1024 + p_byte[rr]);
    const uint32_t csc = __builtin_amdgcn_readfirstlane(kOperandFragsF * kFrag + p_area[cr] * 1024 + p_byte[cr]);
    const bool col_mx = p_codec[cr] == kMx8F;
    const uint32_t rdo0 = (uint32_t)rd_off[0][0], rdo1 = (uint32_t)rd_off[1][0];
    const uint32_t lane16 = (uint32_t)lane * 16u, src16 = (uint32_t)src_unit * 16u;
    // (one loop per column codec: with both statements in one loop body hipcc copies the 144 accumulator registers around every tile)
    auto run_full = [&](auto mx_tag) {
      constexpr bool kColMx = decltype(mx_tag)::value;
      f32x32 c0 = {}, c1 = {}, c2 = {}, c3 = {};  // acc[a][c] at 16 (2 a + c)
      f32x16 cx = {};
      for (int i = 0; i < nt; ++i) {
#ifdef SR_W8_TIMING
        long long* stamp = (prm.dbg && blockIdx.x == 0 && i < 32 && lane == 0) ? prm.dbg + (wave * 32 + i) * 8 : nullptr;
        if (stamp) stamp[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
        const bool more = i + kSlotsF - 1 < nt;
        if (more) issue(t_begin + i + kSlotsF - 1, (i + kSlotsF - 1) & (kSlotsF - 1));
#ifdef SR_W8_TIMING
        if (stamp) stamp[1] = (long long)__builtin_amdgcn_s_memtime();
#endif
        const uint32_t cur = ring + (uint32_t)(i & (kSlotsF - 1)) * kSlotBytes;
        // the last tile decodes nothing useful: the statement still decodes the next slot in place (stale bytes, never read again)
        const uint32_t nxt = ring + (uint32_t)((i + 1) & (kSlotsF - 1)) * kSlotBytes;
#define SR_TILE_OPERANDS                                                                                                               \
  : "+{v[0:31]}"(c0), "+{v[32:63]}"(c1), "+{v[64:95]}"(c2), "+{v[96:127]}"(c3), "+{v[128:143]}"(cx)                                    \
  : "{v235}"(rdo0), "{v236}"(rdo1), "{v237}"(lane16), "{v238}"(src16), [cur] "s"(cur), [nxt] "s"(nxt), [aoff] "s"(aoff),                \
    [boff] "s"(boff), [xoff] "s"(xoff), [rraw] "s"(rraw), [craw] "s"(craw), [rsc] "s"(rsc), [csc] "s"(csc)                             \
  :
        if constexpr (kColMx) {
          asm volatile(
#include "wgrad8f_tile_m.inc"
              SR_TILE_OPERANDS
#include "wgrad8f_tile_clobbers.inc"
          );
        } else {
          asm volatile(
#include "wgrad8f_tile_p.inc"
              SR_TILE_OPERANDS
#include "wgrad8f_tile_clobbers.inc"
          );
        }
#undef SR_TILE_OPERANDS
#ifdef SR_W8_TIMING
        if (stamp) stamp[2] = (long long)__builtin_amdgcn_s_memtime();
#endif
        wait_outstanding(more ? 1 : 0);  // tile i+2 (if any) has landed; only tile i+3 may still be in flight
#ifdef SR_W8_TIMING
        if (stamp) stamp[3] = (long long)__builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SR_W8_TIMING
        if (stamp) stamp[4] = (long long)__builtin_amdgcn_s_memtime();
#endif
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU reads (the statement's last MFMAs are not padded by hipcc)
      const f32x32* cc[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int row = 32 * row_pair[a] + (g & 3) + 8 * (g >> 2) + 4 * hh;
            const int col = 64 * wc + 32 * c + (lane & 31);
            out[row * 256 + col] = (*cc[a])[16 * c + g];
          }
      float* oa = out + 256 * 256;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 32 * xpair + (g & 3) + 8 * (g >> 2) + 4 * hh;
        oa[row * 32 + (lane & 31)] = cx[g];
      }
    };
    if (col_mx) run_full(std::true_type{});
    else run_full(std::false_type{});
    return;
  }

  // ================= other blocks: the r02 contraction, two roles per wave =====================================================
  auto operand = [&](const char* buf, int frag_pair, int ks) {
    const char* p = buf + frag_pair * 2 * kFrag;
    const s16x4f lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4f*)(p + rd_off[ks][0]));
    const s16x4f hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4f*)(p + rd_off[ks][1]));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
  };
  auto mma = [](const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  auto clampi = [](int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; };
  const int wc = wave & 3;
  int wr[2], n_rt[2], n_ct;
  bool aux_on[2];
  n_ct = clampi((16 * nc - 64 * wc + 31) / 32, 2);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    wr[r] = (wave >> 2) + 2 * r;
    n_rt[r] = clampi((16 * nr - 64 * wr[r] + 31) / 32, 2);
    aux_on[r] = wc == 0 ? n_rt[r] >= 1 : wc == 1 ? n_rt[r] == 2 : false;
  }
  f32x16 acc[2][2][2] = {}, acc_aux[2] = {};
  auto kstep = [&](const char* b, int ks) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (n_rt[r] == 0) continue;
      const uint4 a0 = operand(b, 2 * wr[r], ks);
      uint4 a1 = a0;
      if (n_rt[r] > 1) a1 = operand(b, 2 * wr[r] + 1, ks);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        if (ct < n_ct) {
          const uint4 bc = operand(b, 8 + 2 * wc + ct, ks);
          acc[r][0][ct] = mma(a0, bc, acc[r][0][ct]);
          if (n_rt[r] > 1) acc[r][1][ct] = mma(a1, bc, acc[r][1][ct]);
        }
      }
      if (aux_on[r]) acc_aux[r] = mma(wc ? a1 : a0, operand(b, 16, ks), acc_aux[r]);
    }
  };
  for (int i = 0; i < nt; ++i) {
    const char* cur = lds + (i & (kSlotsF - 1)) * kSlotBytes;
    const bool more = i + kSlotsF - 1 < nt;
    if (more) issue(t_begin + i + kSlotsF - 1, (i + kSlotsF - 1) & (kSlotsF - 1));
    kstep(cur, 0);
    if (i + 1 < nt) decode((i + 1) & (kSlotsF - 1));
    kstep(cur, 1);
    wait_outstanding(more ? 1 : 0);
    rendezvous();
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (n_rt[r] > 0 && n_ct > 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int row = 64 * wr[r] + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh;
            const int col = 64 * wc + 32 * ct + (lane & 31);
            if (row < n_rows && col < n_cols) out[row * 256 + col] = acc[r][rt][ct][g];
          }
    }
    if (aux_on[r]) {
      float* oa = out + 256 * 256;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 64 * wr[r] + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
        if (row < n_rows) oa[row * 32 + (lane & 31)] = acc_aux[r][g];
      }
    }
  }
}

int launch_wgrad8f(const uint4* dpre, const uint4* acts, const int* blocks, const int* loads, float* partial, long n_tiles, int n_blocks,
                   int ak, int auxs, int dk, int n_slices, hipStream_t st) {
  Wgrad8Params p;
  p.dpre = dpre, p.acts = acts, p.blocks = blocks, p.loads = loads, p.partial = partial;
  p.n_tiles = n_tiles, p.n_blocks = n_blocks, p.ak = ak, p.auxs = auxs, p.dk = dk;
  p.dbg = nullptr;
#ifdef SR_W8_TIMING  // timing builds only (tools/ab_wgrad8.py passes the address of its stamp buffer): a product build never takes a pointer from the environment
  if (const char* dbg = getenv("SR_W8_DBG")) p.dbg = (long long*)strtoull(dbg, nullptr, 10);
#endif
  const size_t lds = (size_t)kSlotsF * kSlotBytes;
  if (!ensure_dynamic_lds((const void*)wgrad8f_kernel, lds)) return 1;
  hipLaunchKernelGGL(wgrad8f_kernel, dim3(n_slices), dim3(512), lds, st, p);
  return check_launch("wgrad8f_kernel");
}

}  // namespace sr
