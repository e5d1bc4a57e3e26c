// Weight gradients of the fused Sat-NeRF MLP from the 8-bit training workspaces (SR_FMT8, mlp_layout.h) for gfx950:
//   dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Same job as wgrad.hip (autograd's grad_weight = grad_output^T @ input / grad_bias of every nn.Linear in SatNeRF,
// models/satnerf.py:104-153) and the same contraction machinery -- a workgroup owns one job block of up to 256 x 256 (+ its aux
// columns) over a contiguous slice of 32-point tiles, operands are staged point-major in LDS and read back transposed with
// ds_read_b64_tr_b16, fp32 partial blocks are reduced by sr_unpack_grads / sr_grad_tail -- but the operands arrive as ONE BYTE
// per value: a 1-KiB double fragment (DF) carries what two bf16 fragments carried, so a block moves 17 KiB per tile instead
// of 34.  What a block loads is spelled out by a per-block load table built on the host (packing.wgrad8_loads).
//
// Workgroup = 8 waves (two per SIMD, <= 256 VGPRs) in a 2 x 4 grid: wave (wr, wc) owns rows 128 wr .. and columns 64 wc .. of
// the block (4 x 2 MFMA tiles, 128 accumulator registers) plus the aux columns of row tile 4 wr + wc; per point tile it fetches
// (LDS-DMA into a ring of four RAW tiles of 21 KiB: 16 primaries, 2 aux fragments, 3 scale units) and decodes (into one of two
// bf16 operand buffers) ONE row DF and ONE column DF and issues 18 MFMAs.  Every decode is linear -- MX8: (u - 128) * scale,
// SIN8: (u - 127.5) / 128 -- a convert and an fma per value: with the column operand stored as a PHASE the kernel was
// VALU-bound (a v_sin is 16 cycles; 1,150 cycles of MFMA + 1,470 of VALU per tile and SIMD, serialised whatever the
// interleaving: profiles/r02_ab_variants.txt), which is why the forward pass also saves the sin itself.
// The iteration is ONE branch-free scheduling region: 18 MFMAs of tile i, their transposed operand reads and the decode of
// tile i+1, so the decode's VALU / LDS stores issue in the shadow of the MFMAs.  Blocks with fewer than 8 + 8 loads are padded
// on the host with dummy loads that decode into a dump fragment (their MFMA results are never gathered).
//
// Pipeline per 32-point tile i: issue the DMA of tile i+4 into the raw slot tile i left -> [18 MFMAs of tile i from operand
// buffer i & 1 | decode tile i+1 into the other buffer] -> wait for this wave's DMA of tile i+2 -> rendezvous.  A tile's bytes
// are in flight for more than two iterations and are complete and visible to every wave one rendezvous before their decode, so
// a wave may read scale bytes another wave fetched.
#include <stdlib.h>

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
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// load table (host: packing.wgrad8_loads).  ints 0..7: row loads, ints 8..15: column loads (wave w fetches and decodes row load w
// and column load w; never 0: short blocks are padded with dummies aimed at the dump fragment):
//   bits 0-1 source (1 dpre, 2 acts) | 2-3 codec (0 RAW16, 1 SIN8, 2 MX8) | 4-11 unit within the tile | 12-17 operand fragment
//   (0..15 rows, 16..31 columns, 34 = dump; a DF expands into fragment and fragment + 1) | 18-19 scale area | 20-23 byte within
//   the lane's 16 B
// ints 16..18: the scale unit fetched into scale area 0..2 (bits 0-1 source, 4-11 unit; 0 = none), by waves 2..4.
// int 19: the codec of the column loads (one per block).  Row loads are MX8 or RAW16 (one bf16 fragment), chosen per wave by selects.
constexpr int kWg8LoadInts = 20;
enum { kSrcDpre = 1, kSrcActs = 2, kRaw16 = 0, kSin8 = 1, kMx8 = 2 };

constexpr int kFragStride8 = 1088;  // as wgrad.hip: 1-KiB lane-linear fragment image + 64 B so that transposed reads spread over the banks
constexpr int kOperandFrags = 35;   // 16 row + 16 column + 2 aux + 1 dump
constexpr int kDumpFrag = 34;
constexpr int kScaleAreas = 3;
constexpr int kOperandBytes = kOperandFrags * kFragStride8;  // one decoded tile
constexpr int kRawAux = 16 * 1024, kRawScale = 18 * 1024, kRawBytes = (18 + kScaleAreas) * 1024;  // raw tile: primaries | aux | scale units
constexpr int kRawSlots = 4;
constexpr int kLds8Bytes = 2 * kOperandBytes + kRawSlots * kRawBytes;
static_assert(kLds8Bytes <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void wait_vm8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int COLC>
__device__ __forceinline__ void wgrad8_body(const Wgrad8Params& prm, char* lds, const int* d, const int* ld_tab) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = (long)((int)blockIdx.x - d[kWgFirstSlice]) * tiles_per_split;
  long t_end = t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  const int nt = t_end > t_begin ? (int)(t_end - t_begin) : 0;
  const int nr = d[1] + d[3], nc = d[5] + d[7];

  // ---- this wave's loads: row load w, column load w, one secondary (aux fragment / scale unit) -------------------------------
  const int src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);  // rotated image: position `lane` holds this source lane's 16 B
  const int rdesc = __builtin_amdgcn_readfirstlane(ld_tab[wave]), cdesc = __builtin_amdgcn_readfirstlane(ld_tab[8 + wave]);
  const bool r_raw16 = ((rdesc >> 2) & 3) == kRaw16;
  const int r_dst = (rdesc >> 12) & 63, c_dst = (cdesc >> 12) & 63;
  const int r_dst2 = r_raw16 || r_dst == kDumpFrag ? kDumpFrag : r_dst + 1, c_dst2 = c_dst == kDumpFrag ? kDumpFrag : c_dst + 1;
  const int r_soff = kRawScale + ((rdesc >> 18) & 3) * 1024 + src_unit * 16 + ((rdesc >> 20) & 15);
  const int c_soff = kRawScale + ((cdesc >> 18) & 3) * 1024 + src_unit * 16 + ((cdesc >> 20) & 15);
  const long r_stride = (rdesc & 3) == kSrcDpre ? kD8Units : prm.ak, c_stride = (cdesc & 3) == kSrcDpre ? kD8Units : prm.ak;
  const uint4* r_base = ((rdesc & 3) == kSrcDpre ? prm.dpre : prm.acts) + ((rdesc >> 4) & 255) * 64 + src_unit;
  const uint4* c_base = ((cdesc & 3) == kSrcDpre ? prm.dpre : prm.acts) + ((cdesc >> 4) & 255) * 64 + src_unit;
  int sec_src = 0, sec_unit = 0, sec_off = 0;
  if (wave < prm.auxs) {
    sec_src = kSrcActs, sec_unit = wave, sec_off = kRawAux + wave * 1024;
  } else if (wave >= 2 && wave < 2 + kScaleAreas) {
    const int sd = __builtin_amdgcn_readfirstlane(ld_tab[16 + wave - 2]);
    if (sd != 0) sec_src = sd & 3, sec_unit = (sd >> 4) & 255, sec_off = kRawScale + (wave - 2) * 1024;
  }
  const bool has_sec = sec_src != 0;
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  const long s_stride = sec_src == kSrcDpre ? kD8Units : prm.ak;
  const uint4* s_base = (sec_src == kSrcDpre ? prm.dpre : prm.acts) + sec_unit * 64 + (wave < prm.auxs ? src_unit : lane);
  auto issue = [&](long tile, int slot) {
    const uint32_t base = ring + 2 * kOperandBytes + slot * kRawBytes;
    glds16(reinterpret_cast<const char*>(r_base + tile * r_stride * 64), base + wave * 1024);
    glds16(reinterpret_cast<const char*>(c_base + tile * c_stride * 64), base + (8 + wave) * 1024);
    if (has_sec) glds16(reinterpret_cast<const char*>(s_base + tile * s_stride * 64), base + sec_off);
  };
  auto wait_outstanding = [&](int tiles_in_flight) {  // all but the newest `tiles_in_flight` tiles of this wave have landed
    switch (tiles_in_flight * (has_sec ? 3 : 2)) {
      case 9: wait_vm8<9>(); break;
      case 6: wait_vm8<6>(); break;
      case 4: wait_vm8<4>(); break;
      case 3: wait_vm8<3>(); break;
      case 2: wait_vm8<2>(); break;
      default: wait_vm8<0>(); break;
    }
  };
  auto decode = [&](const char* raw, char* ob) {
    // aux fragment wave & 1: copied by every wave (identical bytes; with one aux fragment the second is never gathered) so that
    // the iteration has no branch
    *reinterpret_cast<uint4*>(ob + (32 + (wave & 1)) * kFragStride8 + lane * 16) = *reinterpret_cast<const uint4*>(raw + kRawAux + (wave & 1) * 1024 + lane * 16);
    const uint4 rv = *reinterpret_cast<const uint4*>(raw + wave * 1024 + lane * 16);
    const uint4 cv = *reinterpret_cast<const uint4*>(raw + (8 + wave) * 1024 + lane * 16);
    const uint32_t re = *reinterpret_cast<const uint8_t*>(raw + r_soff);
    const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w}, cw[4] = {cv.x, cv.y, cv.z, cv.w};
    uint32_t ro[8], co[8];
    {
      const float sc = mx8_scale(re), bias = -128.0f * sc;
#pragma unroll
      for (int q = 0; q < 8; ++q) ro[q] = pack_bf16x2(mx8_value(rw[q >> 1], 2 * (q & 1), sc, bias), mx8_value(rw[q >> 1], 2 * (q & 1) + 1, sc, bias));
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[q] = r_raw16 ? rw[q] : ro[q];  // a bf16 fragment goes through as it is (its second half is dumped)
    }
    if constexpr (COLC == kSin8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) co[q] = pack_bf16x2(sin8_value(cw[q >> 1], 2 * (q & 1)), sin8_value(cw[q >> 1], 2 * (q & 1) + 1));
    } else {
      const uint32_t ce = *reinterpret_cast<const uint8_t*>(raw + c_soff);
      const float sc = mx8_scale(ce), bias = -128.0f * sc;
#pragma unroll
      for (int q = 0; q < 8; ++q) co[q] = pack_bf16x2(mx8_value(cw[q >> 1], 2 * (q & 1), sc, bias), mx8_value(cw[q >> 1], 2 * (q & 1) + 1, sc, bias));
    }
#ifdef SR_W8_NO_LDSW
    if (ro[0] + ro[4] + co[0] + co[4] + ro[1] + ro[5] + co[1] + co[5] + ro[2] + ro[6] + co[2] + co[6] + ro[3] + ro[7] + co[3] + co[7] == 0x12345u) *reinterpret_cast<uint4*>(ob) = make_uint4(1, 2, 3, 4);
#else
    *reinterpret_cast<uint4*>(ob + r_dst * kFragStride8 + lane * 16) = make_uint4(ro[0], ro[1], ro[2], ro[3]);
    *reinterpret_cast<uint4*>(ob + r_dst2 * kFragStride8 + lane * 16) = make_uint4(ro[4], ro[5], ro[6], ro[7]);
    *reinterpret_cast<uint4*>(ob + c_dst * kFragStride8 + lane * 16) = make_uint4(co[0], co[1], co[2], co[3]);
    *reinterpret_cast<uint4*>(ob + c_dst2 * kFragStride8 + lane * 16) = make_uint4(co[4], co[5], co[6], co[7]);
#endif
  };

  // ---- MFMA side: operands are read back transposed (a 16-lane group reads a 4-point x 16-slot block, each lane receives one
  // slot's 4 points), two reads per 32x32x16 operand, exactly as wgrad.hip ------------------------------------------------------
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
  auto rendezvous = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  f32x16 acc[4][2] = {}, acc_aux = {};
  auto kstep = [&](const char* b, int ks) {
    uint4 a[4], bc[2];
#ifdef SR_W8_NO_OPREAD
    for (int rt = 0; rt < 4; ++rt) a[rt] = make_uint4(lane + rt, ks, lane, 7);
    for (int ct = 0; ct < 2; ++ct) bc[ct] = make_uint4(lane + ct, ks, 3, lane);
    const uint4 ax = make_uint4(lane, 1, 2, 3);
#else
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) a[rt] = operand(b, 4 * wr + rt, ks);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bc[ct] = operand(b, 8 + 2 * wc + ct, ks);
    const uint4 ax = operand(b, 16, ks);
#endif
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = mma(a[rt], bc[ct], acc[rt][ct]);
    uint4 ar = a[0];
#pragma unroll
    for (int rt = 1; rt < 4; ++rt)
      if (wc == rt) ar = a[rt];
    acc_aux = mma(ar, ax, acc_aux);
  };
  // prologue: tiles 0..3 in flight; tile 0 landed -> visible -> decoded; tile 1 landed -> visible
  const int pre = nt < kRawSlots ? nt : kRawSlots;
  for (int i = 0; i < pre; ++i) issue(t_begin + i, i);
  if (nt > 0) {
    wait_outstanding(pre - 1);
    rendezvous();
    decode(lds + 2 * kOperandBytes, lds);
    wait_outstanding(pre > 2 ? pre - 2 : 0);
    rendezvous();
  }
  for (int i = 0; i < nt; ++i) {
    const char* cur = lds + (i & 1) * kOperandBytes;
#ifndef SR_W8_NO_DMA
    if (i + kRawSlots < nt) issue(t_begin + i + kRawSlots, i & (kRawSlots - 1));  // raw slot of tile i: decoded one iteration ago
#endif
#ifndef SR_W8_NO_MMA
    kstep(cur, 0);
    kstep(cur, 1);
#endif
    // (after the last tile this decodes a stale raw slot into the idle operand buffer: harmless, and the loop stays branch-free)
#ifndef SR_W8_NO_DECODE
    decode(lds + 2 * kOperandBytes + ((i + 1) & (kRawSlots - 1)) * kRawBytes, lds + ((i + 1) & 1) * kOperandBytes);
#endif
    const int issued = i + kRawSlots < nt ? i + kRawSlots : nt - 1;  // newest tile requested so far
    const int fly = issued - (i + 2);
    wait_outstanding(fly > 0 ? fly : 0);  // tile i+2 (if any) has landed; tiles i+3, i+4 may still be in flight
    rendezvous();
  }
  float* out = prm.partial + (long)blockIdx.x * kWgBlockFloats;
  const int n_rows = 16 * nr, n_cols = 16 * nc;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 128 * wr + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh, col = 64 * wc + 32 * ct + (lane & 31);
        if (row < n_rows && col < n_cols) out[row * 256 + col] = acc[rt][ct][g];
      }
  float* oa = out + 256 * 256;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = 128 * wr + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
    if (row < n_rows) oa[row * 32 + (lane & 31)] = acc_aux[g];
  }
}

__global__ void __launch_bounds__(512) wgrad8_kernel(const Wgrad8Params prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int* d = prm.blocks;
  int blk = 0;
  for (; blk + 1 < prm.n_blocks && (int)blockIdx.x >= d[kWgFirstSlice] + d[kWgSlices]; ++blk) d += kWgTableInts;
  const int* ld_tab = prm.loads + blk * kWg8LoadInts;
  if (__builtin_amdgcn_readfirstlane(ld_tab[19]) == kMx8) wgrad8_body<kMx8>(prm, lds, d, ld_tab);
  else wgrad8_body<kSin8>(prm, lds, d, ld_tab);
}

}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_wgrad8(int feat, int tau, int64_t n_points, const uint16_t* dpre, const uint16_t* acts, const int32_t* blocks,
                                 const int32_t* loads, int n_blocks, int n_slices, float* partial, void* stream) {
  SR_REQUIRE(feat == kFeat, "sr_satnerf_wgrad8: feat=%d unsupported", feat);
  SR_REQUIRE(dpre && acts && blocks && loads && partial, "sr_satnerf_wgrad8: null pointer argument");
  SR_REQUIRE(n_blocks >= 1 && n_slices >= n_blocks, "sr_satnerf_wgrad8: bad plan (%d blocks, %d slices): run sr_wgrad_plan first", n_blocks, n_slices);
  Wgrad8Params p;
  p.dpre = (const uint4*)dpre, p.acts = (const uint4*)acts, p.blocks = blocks, p.loads = loads, p.partial = partial;
  p.n_tiles = (n_points + 31) / 32;
  p.n_blocks = n_blocks;
  p.auxs = aux_steps(tau);
  p.ak = act8_units(p.auxs);

  const size_t lds = (size_t)kLds8Bytes;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad8_kernel, dim3(n_slices), dim3(512), lds, (hipStream_t)stream, p);
  return check_launch("wgrad8_kernel");
}

extern "C" int sr_wgrad8_load_ints(void) { return kWg8LoadInts; }
