// Weight gradients of the fused Sat-NeRF MLP from the 8-bit training workspaces, third generation (gfx950):
//   dW[row][col] = sum over sample points of dpre[row] * act[col]
// = autograd's grad_weight = grad_output^T @ input / grad_bias of every nn.Linear in SatNeRF (models/satnerf.py:104-153).  Same job
// table, split-K plan and fp32 partial blocks as wgrad8.hip (reduced by sr_grad_tail / sr_unpack_grads); what changed is the machine:
//
//   * 4 waves per workgroup, ONE PER SIMD, each owning a 128 x 128 quadrant of the 256 x 256 job block: 4 x 4 MFMA tiles = 256 fp32
//     accumulators in AGPRs (+ 2 aux tiles in VGPRs): one transposed operand read per MFMA.
//   * The 8-bit operands go HBM -> VGPRs, are decoded in registers with packed fp16 arithmetic (MX8: v_perm + v_pk_fma_f16 per pair;
//     PHASE8: v_perm + two v_sin_f16 per pair) and reach the LDS once, as fp16 fragments; the MFMAs take fp16 operands.  With one wave
//     per SIMD a wave's time is the sum of its issue slots (tools/probe_lds.hip), so the instruction count per MFMA is what matters.
//   * fp16's range is fitted per workgroup AND per 32-row pair (= per accumulator row tile, one scale group of the dpre workspace): the
//     dX kernel leaves the largest MX8 exponent byte of every scale group per 4 tiles in a table behind the dpre workspace
//     (mlp_layout.h); the workgroup reads the entries of ITS slice of points (a few dozen 16-byte loads; r04 scanned the exponent bytes
//     themselves: 28 MB and 9 us per launch), the rows of pair p are decoded times G_p = 2^(138 - Emax_p) (|value| < 2^12; lanes 2^-19
//     below the largest of their own group flush to zero -- rows of another layer that share the block no longer matter), MX8 columns
//     (feats) likewise with their own Emax, and the fp32 accumulators are unscaled tile by tile when the partial block is written.
//   * The whole slice loop is one generated, hand-placed asm statement (csrc/gen/wgrad9_loop.py -> wgrad9_loop_{p,m}.inc): 36 MFMAs per
//     tile, the decode of tile i + 2, the global loads of tile i + 5 and the operand reads of the next k-step in their gaps; LDS ring
//     of four fp16 slots, waves synchronise through per-slot publish counters in the LDS (no s_barrier in the loop).
//
// What each wave fetches is the "duty" table built on the host (packing.wgrad9_duties): ints 20.. of a block's row of the load table.
#include <stdlib.h>

#include "codec8.h"
#include "common.h"
#include "mlp_device.h"
#include "mlp_layout.h"

namespace sr {

typedef float f32x32w __attribute__((ext_vector_type(32)));

namespace {
constexpr int kFrag9 = 1088, kPair9 = 2 * kFrag9, kSlotFrags9 = 36, kSlot9 = kSlotFrags9 * kFrag9, kSlots9 = 4;  // = gen/wgrad9_loop.py
constexpr int kEmaxFeatsByte = 14;  // = mlp_layout.h kEmaxFeats (this translation unit is width-agnostic and does not include the layout's namespace)
constexpr int kOldInts = 20, kDutyInts = 4, kDuties = 5, kDumpFrag = 34, kPairs = 8;
constexpr int kVariantAt = 4 * kDuties * kDutyInts + kPairs + 1;  // per-wave stream variant behind the duties, the pair groups and the quadrant mask (packing.WG9_VARIANT_INTS)

struct Wgrad9Params {
  const char* dpre;
  const char* acts;
  const uint4* emax;  // table of exponent maxima behind the dpre workspace: one 16-byte entry per 4 tiles (mlp_layout.h)
  const int* blocks;  // planned job table (mlp_layout.h kWgTableInts ints per block)
  const int* loads;   // load_ints ints per block; ints 20.. = the duty table
  float* partial;
  long n_tiles;
  int n_blocks;
  int ak, dk;  // 1-KiB units per tile of the activation / dpre workspace
  int load_ints;
  int keep_dump_raw;   // A/B (SATNERF_WGRAD_NORAW=0): every full-stream wave runs its raw duty, dump or not (the r05 behaviour)
  int span;            // stream-K plans: tile units per workgroup (= int kWgSpan of the table's first row); 0 = one slice per workgroup
  long long* dbg;  // SR_W9_TIMING builds: per workgroup (shader cycles, 100-MHz ticks, tiles) of wave 0's slice loop
};
}  // namespace

// SK: the stream-K build -- the body below runs once per segment of the workgroup's span.  The one-slice build (SK = false) is the same
// source with the loop known to run once: kept separate because the loop-carried state around the slice-loop statement (which clobbers
// nearly every register) cost the one-slice kernel 6 us in its partial-block write (spills), r05.
template <bool SK>
__global__ void __launch_bounds__(256) wgrad9_kernel(const Wgrad9Params prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef SR_W9_TIMING
  const uint64_t tk0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // which job block owns this slice.  sr_wgrad_plan hands this kernel equal slices, the remainder to the first blocks: the block index
  // follows from blockIdx by arithmetic, so the block's table row, this wave's duties, the scan list and the quadrant mask are all
  // fetched at once (one memory latency instead of three dependent ones); the row confirms the guess -- any other plan falls back to a
  // search (one lane per table row, one ballot) and fetches again
  int blk;
  int d[kWgTableInts], du[kDuties * kDutyInts], pairs[kPairs + 1];  // pairs[]: exponent group of each 32-row pair; [kPairs]: the quadrant mask
  int variant = 0;  // this wave's instruction stream: 0 = full, k > 0 = thin stream k (gen/wgrad9_loop.py THIN, packing.WG9_THIN)
  auto fetch_tables = [&](int b) {
    const int* row = prm.blocks + kWgTableInts * b;
    const int* lt = prm.loads + (long)b * prm.load_ints + kOldInts;
#pragma unroll
    for (int i = 0; i < kWgTableInts; ++i) d[i] = row[i];
#pragma unroll
    for (int i = 0; i < kDuties * kDutyInts; ++i) du[i] = lt[wave * kDuties * kDutyInts + i];
#pragma unroll
    for (int i = 0; i < kPairs + 1; ++i) pairs[i] = lt[4 * kDuties * kDutyInts + i];
    variant = lt[kVariantAt + wave];
  };
  // the entries of the exponent-maxima table that cover tiles [first, last]: lane i takes entries first / 4 + i, + 64, ...; folded byte-wise
  // (a u16 maximum orders by the HIGH byte, so pk_max tracks bytes 1 and 3 of a dword exactly in the high bytes of its halves; the same
  // on the dword shifted left by 8 tracks bytes 0 and 2): mx[2 k] / mx[2 k + 1] = odd / even bytes of dword k
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 mx[8];
  auto fetch_emax = [&](long first, long last) {
#pragma unroll
    for (int k = 0; k < 8; ++k) mx[k] = u16x2{0, 0};
    for (long e = first / 4 + lane; e <= last / 4; e += 64) {
      const uint4 w = prm.emax[e];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mx[2 * k] = __builtin_elementwise_max(mx[2 * k], __builtin_bit_cast(u16x2, ww[k]));
        mx[2 * k + 1] = __builtin_elementwise_max(mx[2 * k + 1], __builtin_bit_cast(u16x2, ww[k] << 8));
      }
    }
  };
  auto slice_of = [&](int n_slices, int sl, long& first, long& last) {  // tiles [first, last] of slice sl of n_slices; false: empty
    const long per = (prm.n_tiles + n_slices - 1) / n_slices;
    first = (long)sl * per, last = first + per - 1;
    if (last > prm.n_tiles - 1) last = prm.n_tiles - 1;
    return first <= last;
  };
  // stream-K plans (sr_wgrad_plan writes the span into int 11 of the table's first row; 0 = one slice per workgroup): the job list is ONE
  // line of n_blocks x n_tiles tile units and workgroup i takes units [i span, (i + 1) span) -- a workgroup whose span crosses a block
  // boundary finishes its slice of block b, writes the partial block, and starts over on block b + 1 (`seg`).  At width 512 the 47 blocks
  // do not divide the 256 workgroups (5 or 6 slices each: 342 / 410 tiles, a sixth of the chip idle at the end); this gives every
  // workgroup 376.
  const long span = SK ? (long)prm.span : 0l;
  const long u_all = (long)prm.n_blocks * prm.n_tiles;
  long u0 = (long)blockIdx.x * span;
  const long u_end = u0 + span < u_all ? u0 + span : u_all;
  if (span > 0 && u0 >= u_all) return;  // (the launch has one workgroup per partial slice: a few more than spans)
  for (int seg = 0;; ++seg) {
  if (seg > 0) __syncthreads();  // every wave has left the previous segment's slice loop: the LDS slots and counters are free again
  int slice;  // this workgroup's slice of the block: partial block d[kWgFirstSlice] + slice
  long sk_begin = 0, sk_end = 0;  // (stream-K) this segment's tiles of block blk
  if (span > 0) {
    blk = (int)(u0 / prm.n_tiles);
    sk_begin = u0 - (long)blk * prm.n_tiles;
    sk_end = sk_begin + (u_end - u0) < prm.n_tiles ? sk_begin + (u_end - u0) : prm.n_tiles;
    slice = (int)((long)blockIdx.x - ((long)blk * prm.n_tiles) / span);  // the block's first workgroup writes its slice 0
    fetch_tables(blk);
    fetch_emax(sk_begin, sk_end - 1);
  } else {
    // workgroup i = slice i / n_blocks of block i % n_blocks (the r blocks with one slice more take the last r workgroups): workgroup i
    // runs on XCD i % 8, so blocks 8 table positions apart -- packing.backward_maps puts blocks that share an operand there -- stream the
    // same tiles through the same L2 at the same time.  Valid for sr_wgrad_plan's equal split (q or q + 1 slices, the larger first)
    const int q = (int)gridDim.x / prm.n_blocks, r = (int)gridDim.x % prm.n_blocks, idx = (int)blockIdx.x;
    if (q == 0) blk = idx < prm.n_blocks ? idx : prm.n_blocks - 1, slice = 0;
    else if (idx < q * prm.n_blocks) blk = idx % prm.n_blocks, slice = idx / prm.n_blocks;
    else blk = idx - q * prm.n_blocks, slice = q;
    fetch_tables(blk);
    {  // the slice's entries of the exponent maxima, on the same guess (in flight together with the tables)
      long first, last;
      if (q > 0 && slice_of(q + (blk < r ? 1 : 0), slice, first, last)) fetch_emax(first, last);
      else fetch_emax(1, 0);
    }
    // the arithmetic numbering is valid only if EVERY block has q (+ 1 for the first r) slices: all workgroups take the same decision from
    // the whole table (one lane per row, fetched alongside the tables above)
    bool equal_split = q > 0;
    for (int b0 = 0; b0 < prm.n_blocks; b0 += 64) {
      const int b = b0 + lane;
      const bool bad = b < prm.n_blocks && prm.blocks[kWgTableInts * b + kWgSlices] != q + (b < r ? 1 : 0);
      if (__ballot(bad)) equal_split = false;
    }
    if (!equal_split && prm.n_blocks <= 64) {
      // a cost-weighted plan (r06: thin blocks take fewer workgroups): still SLICE-MAJOR -- workgroups are handed out level by level,
      // level s to every block that has more than s slices, in table order -- so that for the levels all blocks share the numbering is
      // the equal split's (blocks 8 table positions apart on one XCD at the same time), and a block that runs out of slices only
      // closes the ranks behind it.  One lane per block holds its slice count; a ballot per level finds this workgroup's.
      const int mine = lane < prm.n_blocks ? prm.blocks[kWgTableInts * lane + kWgSlices] : 0;
      int rem = idx;
      blk = prm.n_blocks - 1, slice = 0;
      for (int lvl = 0; lvl < 4096; ++lvl) {
        const unsigned long long has = __ballot(mine > lvl);
        const int cnt = __builtin_popcountll(has);
        if (cnt == 0) break;  // (more workgroups than slices: the launch never does that)
        if (rem < cnt) {
          unsigned long long m = has;
          for (int k = 0; k < rem; ++k) m &= m - 1;  // drop the `rem` lowest set bits
          blk = __builtin_ctzll(m), slice = lvl;
          break;
        }
        rem -= cnt;
      }
      blk = __builtin_amdgcn_readfirstlane(blk), slice = __builtin_amdgcn_readfirstlane(slice);
      fetch_tables(blk);
      long first, last;
      if (slice_of(d[kWgSlices], slice, first, last)) fetch_emax(first, last);
      else fetch_emax(1, 0);
    } else if (!equal_split) {  // some other plan: block-major numbering, found by search
      blk = 0;
      for (int b0 = 0; b0 < prm.n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const bool mine = b < prm.n_blocks && idx >= prm.blocks[kWgTableInts * b + kWgFirstSlice] &&
                          idx < prm.blocks[kWgTableInts * b + kWgFirstSlice] + prm.blocks[kWgTableInts * b + kWgSlices];
        const unsigned long long hit = __ballot(mine);
        if (hit) { blk = b0 + __builtin_ctzll(hit); break; }
      }
      blk = __builtin_amdgcn_readfirstlane(blk);
      fetch_tables(blk);
      slice = idx - d[kWgFirstSlice];
      long first, last;
      if (slice_of(d[kWgSlices], slice, first, last)) fetch_emax(first, last);
      else fetch_emax(1, 0);
    }
  }
  const int nr = d[1] + d[3], nc = d[5] + d[7];
  const bool col_mx = d[8] == 0;  // packing.KIND_BF16: the identity stage (feats) -> MX8 columns
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = span > 0 ? sk_begin : (long)slice * tiles_per_split;
  long t_end = span > 0 ? sk_end : t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  // (every value the slice loop takes as a scalar operand is forced into an SGPR: the table reads above are scalar loads only as long as
  // the compiler can prove them uniform, and an "s" operand held in a VGPR is passed in that VGPR without a diagnostic)
  uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t_end > t_begin ? (uint32_t)(t_end - t_begin) : 0u));
  const uint32_t tleft = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t_begin < prm.n_tiles ? (uint32_t)(prm.n_tiles - 1 - t_begin) : 0u));
  const long t0 = t_begin < prm.n_tiles ? t_begin : 0;

  // ---- this wave's duties -> wave-uniform bases -------------------------------------------------------------------------------
  uint64_t base[kDuties], sbase[4];
  uint32_t wb[kDuties];
  bool on[kDuties];  // the duty feeds an operand fragment (not the dump area)
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  int raw_src = 0;
#pragma unroll
  for (int k = 0; k < kDuties; ++k) {
    const int src = __builtin_amdgcn_readfirstlane(du[kDutyInts * k]), unit = __builtin_amdgcn_readfirstlane(du[kDutyInts * k + 1]);
    const int dst = __builtin_amdgcn_readfirstlane(du[kDutyInts * k + 2]), sc = __builtin_amdgcn_readfirstlane(du[kDutyInts * k + 3]);
    const char* ws = src == 1 ? prm.dpre : prm.acts;
    auto uni64 = [](uint64_t v) {
      return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    base[k] = uni64((uint64_t)(uintptr_t)(ws + (long)unit * 1024));
    if (k < 4) sbase[k] = uni64((uint64_t)(uintptr_t)(ws + (long)(sc >> 4) * 1024 + (sc & 15)));
    wb[k] = ring + (uint32_t)dst * kFrag9;
    on[k] = dst != kDumpFrag;
    if (k == 4) raw_src = src;
  }
#ifdef SR_W9_L2WINDOW  // A/B build (wrong results): every tile of a slice re-reads the slice's FIRST tile -- the same loop, decode and MFMAs with its
  const uint32_t strd = 0u, stra = 0u;  // 410 MB of operand traffic served by the L2 instead of HBM (is the kernel's 4.25 TB/s free?  DESIGN section 4)
#else
  const uint32_t strd = (uint32_t)__builtin_amdgcn_readfirstlane(prm.dk * 1024), stra = (uint32_t)__builtin_amdgcn_readfirstlane(prm.ak * 1024);
#endif
  const uint32_t strx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(raw_src == 1 ? strd : stra));
  // rotated image (as wgrad8): LDS position `lane` of a fragment holds source lane src_unit's 16 bytes, so that the transposed reads
  // spread over the banks; every global load fetches that lane's bytes, every LDS write goes to lane * 16
  const uint32_t src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);
  const uint32_t vd = (uint32_t)t0 * strd + src_unit * 16u, va = (uint32_t)t0 * stra + src_unit * 16u, vx = (uint32_t)t0 * strx + src_unit * 16u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // transposed operand reads (as wgrad8): per-lane offset of k-step 0 / 1 inside a fragment pair
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  uint32_t rdo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int point = 16 * ks + 8 * hh + m;
    rdo[ks] = ring + (uint32_t)(rh * kFrag9 + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8);
  }
  // quadrant (wr, wc): row pairs 4 wr + ((a + 2 wc) & 3) for operand slot a = 0..3 (slots 0, 1 also take the aux columns), column pairs
  // 8 + 4 wc + c
  const int wr = wave >> 1, wc = wave & 1;
  variant = __builtin_amdgcn_readfirstlane(variant);
  const bool thin = variant > 0;  // thin streams contract row pair 0 (the block's only one) in operand slot 0, whatever the wave
  const uint32_t aofl = thin ? 0u : (uint32_t)(4 * wr + 2 * wc) * kPair9, aofh = (uint32_t)(4 * wr + ((2 * wc + 2) & 3)) * kPair9;
  const uint32_t bof = (uint32_t)(8 + 4 * wc) * kPair9;

  // ---- fp16 range per row pair (and of MX8 columns) over this workgroup's slice: the byte-wise maxima over the lanes' entries ----------
  // a value is < 2^(E - 126) for an MX8 lane with exponent byte E and for a bf16 with biased exponent E alike
#ifdef SR_W9_TIMING
  const uint64_t tk1 = __builtin_amdgcn_s_memrealtime() + (uint64_t)(d[0] & 0);  // (after the tables have arrived)
#endif
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, mx[k]), sh);
      mx[k] = __builtin_elementwise_max(mx[k], __builtin_bit_cast(u16x2, o));
    }
  }
  auto emax_of = [&](int g) -> uint32_t {  // byte g of the folded entry (g wave-uniform; < 0: no such pair); clamped: gradients below
    uint32_t w = 0;                        // 2^-94 are zero for every purpose, and the scales stay normal floats
#pragma unroll
    for (int k = 0; k < 8; ++k) w = (g >> 2) == (k >> 1) && ((g & 1) != (k & 1)) ? __builtin_bit_cast(uint32_t, mx[k]) : w;
    uint32_t e = g < 0 ? 0u : (g & 2) ? w >> 24 : (w >> 8) & 0xffu;
    e = e < 32u ? 32u : e > 254u ? 254u : e;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
  };
  uint32_t ep[kPairs];  // Emax of row pair p
#pragma unroll
  for (int p = 0; p < kPairs; ++p) ep[p] = emax_of(__builtin_amdgcn_readfirstlane(pairs[p]));
  const uint32_t ec = emax_of(kEmaxFeatsByte);
  auto pair_emax = [&](int p) -> uint32_t {  // (p wave-uniform)
    uint32_t e = ep[0];
#pragma unroll
    for (int k = 1; k < kPairs; ++k) e = p == k ? ep[k] : e;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
  };
  if (tid < kSlots9) reinterpret_cast<uint32_t*>(lds + kSlots9 * kSlot9)[tid] = 0u;  // the slots' publish counters (gen/wgrad9_loop.py)
  __syncthreads();
  // rows of pair p: value * G_p, G_p = 2^(138 - Emax_p); the stream forms the fp16 scale of a lane as exponent field E - erow.  Everything
  // handed to the stream as a scalar is made one explicitly: an "s" operand fed a value the compiler holds in a VGPR is silently passed in
  // that VGPR, which may be one of the statement's output registers (the stream zeroes them first)
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  // this wave's two row duties decode into LDS fragments (wb - ring) / kFrag9: pair = fragment / 2 (a dump duty: any scale will do)
  const int dpair0 = __builtin_amdgcn_readfirstlane(du[2]) >> 1, dpair1 = __builtin_amdgcn_readfirstlane(du[kDutyInts + 2]) >> 1;
  const int dpairx = __builtin_amdgcn_readfirstlane(du[4 * kDutyInts + 2]) >> 1;  // ... and the raw duty, when it is a dpre row
  const uint32_t erow0 = uni(pair_emax(dpair0 < kPairs ? dpair0 : 0) - 20u), erow1 = uni(pair_emax(dpair1 < kPairs ? dpair1 : 0) - 20u);
  const uint32_t ecol = uni(ec - 20u);
  const uint32_t g_col_bits = uni(col_mx ? (265u - ec) << 23 : 0x3f800000u);
  const float g_col = __builtin_bit_cast(float, g_col_bits);
  // the raw fragment: a dpre row (x G of its pair) or the aux columns (x 1)
  const uint32_t sraw = uni(raw_src == 1 ? (265u - pair_emax(dpairx < kPairs ? dpairx : 0)) << 23 : 0x3f800000u);
  const float un_col = 1.0f / g_col;  // (powers of two: exact)
  float un_row[4];                    // accumulator row tile a of this wave = pair 4 wr + ((a + 2 wc) & 3)
#pragma unroll
  for (int a = 0; a < 4; ++a) un_row[a] = __builtin_bit_cast(float, uni((pair_emax(thin ? 0 : 4 * wr + ((a + 2 * wc) & 3)) - 11u) << 23));  // 2^(Emax - 138)

  const uint32_t flags = ring + (uint32_t)(kSlots9 * kSlot9);
  f32x32w c0, c1, c2, c3, c4, c5, c6, c7, cx;
#ifdef SR_W9_TIMING
  const uint32_t nt_in = nt;
  const uint64_t tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef SR_W9_P_INC
#define SR_W9_P_INC "wgrad9_loop_p.inc"
#define SR_W9_M_INC "wgrad9_loop_m.inc"
#define SR_W9_PX_INC "wgrad9_loop_px.inc"
#define SR_W9_MX_INC "wgrad9_loop_mx.inc"
#endif
#define SR_W9_OUTS                                                                                                                    \
  "=&{a[0:31]}"(c0), "=&{a[32:63]}"(c1), "=&{a[64:95]}"(c2), "=&{a[96:127]}"(c3), "=&{a[128:159]}"(c4), "=&{a[160:191]}"(c5),        \
      "=&{a[192:223]}"(c6), "=&{a[224:255]}"(c7), "=&{v[0:31]}"(cx), [nt] "+s"(nt)
#define SR_W9_INS                                                                                                                     \
  "{v240}"(rdo[0]), "{v241}"(rdo[1]), "{v242}"(lane16), "{v243}"(vd), "{v244}"(va), "{v245}"(vx), [b0] "s"(base[0]), [b1] "s"(base[1]), \
      [b2] "s"(base[2]), [b3] "s"(base[3]), [bx] "s"(base[4]), [sb0] "s"(sbase[0]), [sb1] "s"(sbase[1]), [sb2] "s"(sbase[2]),        \
      [sb3] "s"(sbase[3]), [w0] "s"(wb[0]), [w1] "s"(wb[1]), [w2] "s"(wb[2]), [w3] "s"(wb[3]), [wx] "s"(wb[4]), [aofl] "s"(aofl),    \
      [aofh] "s"(aofh), [bof] "s"(bof), [strd] "s"(strd), [stra] "s"(stra), [strx] "s"(strx), [tleft] "s"(tleft), [erow0] "s"(erow0), \
      [erow1] "s"(erow1), [ecol] "s"(ecol), [sraw] "s"(sraw), [flags] "s"(flags)
  // a wave whose 128 x 128 quadrant nobody reads (narrow blocks: packing.wgrad9_duties' quadrant mask) runs the stream without the 32
  // main MFMAs and their operand reads: same loads, decode, rendezvous, aux tiles -- the time of a tile is unchanged, its energy is not
  const bool quad_on = (__builtin_amdgcn_readfirstlane(pairs[kPairs]) >> wave) & 1;
  // a full-stream wave whose raw duty is a dump (no aux fragment, no bf16 row fragment for it) runs the variant without that duty
  const bool has_raw = prm.keep_dump_raw || __builtin_amdgcn_readfirstlane(du[4 * kDutyInts + 2]) != kDumpFrag;
  if (variant == 1) {
    asm volatile(
#include "wgrad9_loop_t1.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (variant == 2) {
    asm volatile(
#include "wgrad9_loop_d3.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (variant == 3) {
    asm volatile(
#include "wgrad9_loop_t0.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (variant == 4) {
    asm volatile(
#include "wgrad9_loop_r2.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (variant == 5) {
    asm volatile(
#include "wgrad9_loop_d2.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (variant == 6) {
    asm volatile(
#include "wgrad9_loop_d0.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (!has_raw && col_mx && quad_on) {   // (r06) full streams without the raw duty: this wave's raw duty is a dump
    asm volatile(
#include "wgrad9_loop_mn.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (!has_raw && col_mx) {
    asm volatile(
#include "wgrad9_loop_mxn.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (!has_raw && quad_on) {
    asm volatile(
#include "wgrad9_loop_pn.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (!has_raw) {
    asm volatile(
#include "wgrad9_loop_pxn.inc"
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (col_mx && quad_on) {
    asm volatile(
#include SR_W9_M_INC
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (col_mx) {
    asm volatile(
#include SR_W9_MX_INC
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else if (quad_on) {
    asm volatile(
#include SR_W9_P_INC
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  } else {
    asm volatile(
#include SR_W9_PX_INC
        : SR_W9_OUTS
        : SR_W9_INS
        :
#include "wgrad9_loop_clobbers.inc"
    );
  }
#undef SR_W9_OUTS
#undef SR_W9_INS
#ifdef SR_W9_TIMING
  const uint64_t tr1 = __builtin_amdgcn_s_memrealtime();
  if (prm.dbg && tid == 0) {
    prm.dbg[3 * blockIdx.x] = (long long)(__builtin_amdgcn_s_memtime() - tc0);
    prm.dbg[3 * blockIdx.x + 1] = (long long)(tr1 - tr0);
    prm.dbg[3 * blockIdx.x + 2] = nt_in;
    prm.dbg[3 * 1024 + 4 * blockIdx.x] = (long long)tk0, prm.dbg[3 * 1024 + 4 * blockIdx.x + 1] = (long long)tr0;
    prm.dbg[7 * 1024 + blockIdx.x] = (long long)tk1;
    prm.dbg[3 * 1024 + 4 * blockIdx.x + 2] = (long long)tr1;
  }
#endif

  // ---- partial block of this slice ------------------------------------------------------------------------------------------------
  float* out = prm.partial + (long)(d[kWgFirstSlice] + slice) * kWgBlockFloats;
  const int n_rows = 16 * nr, n_cols = 16 * nc;
  const f32x32w* cc[8] = {&c0, &c1, &c2, &c3, &c4, &c5, &c6, &c7};
  // thin streams: variants 1 and 3 (t1, t0) contract row pair 0 into accumulator row tile 0; the others hold no results
  const bool thin_mfma = variant == 1 || variant == 3;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row0 = thin ? 0 : 32 * (4 * wr + ((a + 2 * wc) & 3));
    if (thin && (a != 0 || !thin_mfma)) continue;
    if (row0 >= n_rows || !quad_on) continue;  // (wave-uniform) tiles the block does not have / nobody reads: never written, never summed
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 128 * wc + 32 * c + (lane & 31);
      if (128 * wc + 32 * c >= n_cols) continue;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = row0 + (g & 3) + 8 * (g >> 2) + 4 * hh;
        out[row * 256 + col] = (*cc[2 * a + (c >> 1)])[16 * (c & 1) + g] * un_row[a] * un_col;
      }
    }
  }
  float* oa = out + 256 * 256;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row0 = thin ? 0 : 32 * (4 * wr + ((a + 2 * wc) & 3));
    if (thin && (a != 0 || !thin_mfma || wc != 0)) continue;  // (both thin MFMA waves hold pair 0's aux tile: the first one writes it)
    if (row0 >= n_rows) continue;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int row = row0 + (g & 3) + 8 * (g >> 2) + 4 * hh;
      oa[row * 32 + (lane & 31)] = cx[16 * a + g] * un_row[a];
    }
  }
#ifdef SR_W9_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  if (prm.dbg && tid == 0) prm.dbg[3 * 1024 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (!SK) break;
  u0 += t_end - t_begin;
  if (u0 >= u_end) break;
  }  // next segment of a stream-K span
}

// the 4-wave kernel reads both workspaces through 32-bit per-lane offsets
bool wgrad9_fits(long n_tiles, int ak, int dk) {
  const long big = ak > dk ? ak : dk;
  return n_tiles * big * 1024l < (1l << 32);
}

int launch_wgrad9(const uint4* dpre, const uint4* acts, const uint4* emax, const int* blocks, const int* loads, float* partial, long n_tiles,
                  int n_blocks, int ak, int dk, int load_ints, int n_slices, int span, hipStream_t st) {
  Wgrad9Params p;
  p.dpre = (const char*)dpre, p.acts = (const char*)acts, p.emax = emax, p.blocks = blocks, p.loads = loads, p.partial = partial;
  p.n_tiles = n_tiles, p.n_blocks = n_blocks, p.ak = ak, p.dk = dk, p.load_ints = load_ints, p.span = span;
  p.dbg = nullptr;
  static const bool noraw_off = [] { const char* e = getenv("SATNERF_WGRAD_NORAW"); return e && e[0] == '0'; }();
  p.keep_dump_raw = noraw_off ? 1 : 0;
#ifdef SR_W9_TIMING  // timing builds only (tools/ab_wgrad8.py passes the address of its stamp buffer): a product build never takes a pointer from the environment
  if (const char* dbg = getenv("SR_W9_DBG")) p.dbg = (long long*)strtoull(dbg, nullptr, 10);
#endif
  const size_t lds = (size_t)kSlots9 * kSlot9 + 16;  // four operand slots + their publish counters
  if (span > 0) {
    if (!ensure_dynamic_lds((const void*)wgrad9_kernel<true>, lds)) return 1;
    hipLaunchKernelGGL(wgrad9_kernel<true>, dim3(n_slices), dim3(256), lds, st, p);
    return check_launch("wgrad9_kernel");
  }
  if (!ensure_dynamic_lds((const void*)wgrad9_kernel<false>, lds)) return 1;
  hipLaunchKernelGGL(wgrad9_kernel<false>, dim3(n_slices), dim3(256), lds, st, p);
  return check_launch("wgrad9_kernel");
}

}  // namespace sr
