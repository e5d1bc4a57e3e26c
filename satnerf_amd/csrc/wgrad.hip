// Weight gradients of the fused Sat-NeRF MLP for gfx950:  dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Replaces autograd's `grad_weight = grad_output^T @ input` / `grad_bias = sum(grad_output)` of every nn.Linear in
// SatNeRF (models/satnerf.py:104-153).  Both operands live in HBM as B fragments (point-major: one 16-byte unit holds 8
// consecutive slots of ONE point, see mlp_layout.h) because that is how the forward / dX kernels hold them in registers;
// this contraction runs over POINTS, so both MFMA operands need 8 consecutive points of one slot instead.  The transpose
// is done by the LDS: fragments are staged point-major and read back with ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-point x 16-slot block and each lane receives one slot's 4 points), two reads per 32x32x16 MFMA operand.
//
// Grid = split-K slices of the job blocks (sr_wgrad_plan numbers them).  A workgroup (16 waves) owns one job block of up to
// 256 x 256 (16 row fragments of dpre x 16 column fragments of the saved activations, packing.backward_maps lists them) PLUS
// the block's aux columns (biases, skip / sun / embedding columns) over a contiguous slice of 32-point tiles, so every operand
// fragment is read from HBM once per job.  Phase-coded activation fragments are decoded to bf16 sin() in LDS.  fp32 partial
// blocks go to `partial[slice][256*256 + 256*32]`; sr_unpack_grads sums a block's slices and scatters into the flat gradient.
#include <stdlib.h>

#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"

namespace sr {

struct WgradParams {
  const uint4* dpre;
  const uint4* acts;
  const int* blocks;  // kWgTableInts ints per block (mlp_layout.h): two row ranges, two column ranges, kind, n_slices, first_slice
  float* partial;
  long n_tiles;
  int n_blocks;
  int ak;             // activation fragments per tile
  int auxs;           // aux fragments (1 or 2) at the head of each activation tile
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS image of one fragment: the 1-KiB fragment as the LDS-DMA writes it (lane-linear) with hslot 1 (lanes 32..63) rotated by
// 8 points -- the DMA cannot pad, but each lane chooses which 16-byte unit it fetches -- and a 64-byte gap after every
// fragment: the four 64-byte pieces a 32-lane group of a transposed read touches (2 fragments x 2 hslots) then sit 64 B
// apart in the 256-B bank row.
constexpr int kFragStride = 1088;
constexpr int kWgFrags = 34;  // 16 row + 16 column + 2 aux fragments
constexpr int kSlotBytes = kWgFrags * kFragStride;
constexpr int kSlots = 4;     // ring of point tiles: one being multiplied, three in flight / landed
constexpr int kBlockFloats = kWgBlockFloats;

__device__ __forceinline__ uint32_t phase_pair_to_bf16(uint32_t w) {
  const float a = __builtin_amdgcn_sinf((float)(w & 0xffffu) * (1.0f / 65535.0f));
  const float b = __builtin_amdgcn_sinf((float)(w >> 16) * (1.0f / 65535.0f));
  return pack_bf16x2(a, b);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup = 16 waves in a 4 x 4 grid; wave (wr, wc) owns rows 64*wr.. and columns 64*wc.. of the 256 x 256 block (2 x 2 MFMA
// tiles, 64 accumulator registers, <= 128 VGPRs, four waves per SIMD) plus the aux columns of one row tile; it moves ONE row
// fragment, ONE column fragment (waves 0,1 also an aux fragment) per point tile and decodes the column fragment it fetched.
//
// Pipeline (MI355X measurements in profiles/r01_ab_variants.txt).  Fragments go HBM -> LDS by LDS-DMA into a 4-slot ring of
// 32-point tiles (three tiles, ~100 KiB per CU, in flight, no staging registers); phase-coded fragments are decoded to bf16
// sin in place one tile ahead; one rendezvous per tile.  Iteration i: issue the pieces of tile i+3 into the slot the previous
// rendezvous released -> k-step 0 of tile i -> wait for this wave's pieces of tile i+1, decode -> k-step 1 -> rendezvous.
// A workgroup's time per tile is set by that latency chain, not by what the block moves (LDS 23 %, MFMA 30 % busy, 0.8-1.2 us
// per tile for a one-fragment head block and a full block alike), so: (a) narrow layers are packed together in the job table,
// (b) 16 thin waves beat 8 fat ones (each has half the DMA-issue stalls, decode and operand reads per tile and twice as many
// waves cover each other: 196 -> 187 us, 0.505 -> 0.49 ms per training step) although they read 1.6x the LDS operands,
// (c) blocks narrower than 256 run an edge path that skips dead tiles and fragments.
__global__ void __launch_bounds__(1024) wgrad_kernel(const WgradParams prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* d = prm.blocks;
  for (int b = 0; b + 1 < prm.n_blocks && (int)blockIdx.x >= d[kWgFirstSlice] + d[kWgSlices]; ++b) d += kWgTableInts;
  const int rf0 = d[0], nr0 = d[1], rf1 = d[2], nr = d[1] + d[3], cf0 = d[4], nc0 = d[5], cf1 = d[6], nc = d[5] + d[7], kind = d[8];
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = (long)((int)blockIdx.x - d[kWgFirstSlice]) * tiles_per_split;
  long t_end = t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  const int nt = t_end > t_begin ? (int)(t_end - t_begin) : 0;

  const bool ld_r = wave < nr, ld_c = wave < nc, ld_ax = wave < 2;
  const int fr = ld_r ? (wave < nr0 ? rf0 + wave : rf1 + wave - nr0) : 0;
  const int fc = ld_c ? (wave < nc0 ? cf0 + wave : cf1 + wave - nc0) : 0;
  const int fa = wave < prm.auxs ? wave : prm.auxs - 1;
  const int n_ld = (int)ld_r + (int)ld_c + (int)ld_ax;  // 0..3
  const int src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto issue = [&](long tile, int slot) {
    const char* dp = reinterpret_cast<const char*>(prm.dpre + tile * kDpFrags * 64);  // wave-uniform bases + the lane's fixed offset
    const char* ac = reinterpret_cast<const char*>(prm.acts + tile * prm.ak * 64);
    const uint32_t base = ring + slot * kSlotBytes + wave * kFragStride, voff = (uint32_t)src_unit * 16u;
    if (ld_r) glds16_s(dp + fr * 1024, voff, base);
    if (ld_c) glds16_s(ac + fc * 1024, voff, base + 16 * kFragStride);
    if (ld_ax) glds16_s(ac + fa * 1024, voff, base + 32 * kFragStride);
  };
  auto wait_tiles = [&](int c) {
    switch (c * n_ld) {
      case 9: wait_vm<9>(); break;
      case 6: wait_vm<6>(); break;
      case 4: wait_vm<4>(); break;
      case 3: wait_vm<3>(); break;
      case 2: wait_vm<2>(); break;
      case 1: wait_vm<1>(); break;
      default: wait_vm<0>(); break;
    }
  };
  auto decode = [&](int slot) {
    if (kind != 1 || !ld_c) return;
    uint4* p = reinterpret_cast<uint4*>(lds + slot * kSlotBytes + (16 + wave) * kFragStride + lane * 16);
    const uint4 v = *p;
    *p = make_uint4(phase_pair_to_bf16(v.x), phase_pair_to_bf16(v.y), phase_pair_to_bf16(v.z), phase_pair_to_bf16(v.w));
  };
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  int rd_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int point = 16 * ks + 8 * hh + 4 * rd + m;
      rd_off[ks][rd] = rh * kFragStride + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8;
    }
  auto operand = [&](const char* buf, int frag_pair, int ks) {
    const char* p = buf + frag_pair * 2 * kFragStride;
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
  // aux columns: row tile 2*wr is handled by the wave with wc == 0, row tile 2*wr + 1 by wc == 1
  const bool aux_on = wc == 0 ? n_rt >= 1 : wc == 1 ? n_rt == 2 : false;
  const bool full = n_rt == 2 && n_ct == 2;
  auto rendezvous = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  auto run = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
    f32x16 acc[2][2] = {}, acc_aux = {};
    auto kstep = [&](const char* b, int ks) {
      if constexpr (kFull) {
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
    for (int i = 0; i < kSlots - 1 && i < nt; ++i) issue(t_begin + i, i);
    if (nt > 0) {
      wait_tiles((nt < kSlots - 1 ? nt : kSlots - 1) - 1);
      decode(0);
      rendezvous();
    }
    for (int i = 0; i < nt; ++i) {
      const char* cur = lds + (i & (kSlots - 1)) * kSlotBytes;
      if (i + kSlots - 1 < nt) issue(t_begin + i + kSlots - 1, (i + kSlots - 1) & (kSlots - 1));
      kstep(cur, 0);
      if (i + 1 < nt) {
        const int last = nt - 1 < i + kSlots - 1 ? nt - 1 : i + kSlots - 1;
        wait_tiles(last - (i + 1));
        decode((i + 1) & (kSlots - 1));
      }
      kstep(cur, 1);
      rendezvous();
    }
    float* out = prm.partial + (long)blockIdx.x * kBlockFloats;
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

// grad[e] (+)= gscale[e] * sum over the slices of partial element gidx[e]
__global__ void __launch_bounds__(256) unpack_grads_kernel(const float* __restrict__ partial, const int* __restrict__ gidx,
                                                          const float* __restrict__ gscale, long n, const int* __restrict__ blocks,
                                                          float* __restrict__ grad, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = gidx[i];
  if (k < 0) return;  // not produced by the fused MLP (sky head): left to its own kernel
  const float s = wg_sum_slices(partial, blocks, k) * gscale[i];
  grad[i] = accumulate ? grad[i] + s : s;
}

}  // namespace sr

using namespace sr;

// Split-K plan (host): slices per job block, written into the table together with each block's first slice number.
// Measured on MI355X (profiles/r01_ab_variants.txt): a workgroup's time per 32-point tile is set by the per-tile latency chain
// (LDS-DMA landed -> decode -> rendezvous -> transposed reads -> MFMA), 1.0-1.2 us for narrow head blocks and full
// 256 x 256 blocks alike, so equal slices for every block beat slices in proportion to the fragments moved (195 vs 282 us at
// 65,536 points), and n_blocks * floor(n_wg / n_blocks) <= n_wg workgroups keeps the launch to one round of workgroups.
extern "C" int sr_wgrad_plan(int32_t* blocks, int n_blocks, int64_t n_points, int n_wg, int fmt, int* n_slices) {
  SR_REQUIRE(blocks && n_slices, "sr_wgrad_plan: null pointer");
  SR_REQUIRE(n_blocks >= 1 && n_blocks <= 4096 && n_points >= 1, "sr_wgrad_plan: bad sizes (%d blocks, %lld points)", n_blocks, (long long)n_points);
  if (n_wg <= 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
    n_wg = cus;
  }
  const long n_tiles = (n_points + 31) / 32;
  long per_block = n_wg / n_blocks;
  per_block = per_block < 1 ? 1 : per_block > n_tiles ? n_tiles : per_block;
#ifdef SR_PLAN_ENV
  if (const char* e = getenv("SR_WGRAD_UNIFORM")) per_block = atoi(e);
#endif
  // cost-weighted split (fmt 8): a block's time per tile grows with the fragments it decodes, so slices are handed out greedily
  // to the block whose workgroups would otherwise finish last: minimises max_b cost_b * ceil(tiles / slices_b)
  // cost of a tile of a block = ca + cb * row fragments + cc * column fragments (relative; MI355X A/B in profiles/r02_ab_variants.txt)
  double ca = 0.4, cb = 0.02, cc = 0.0175;
  // the 4-wave kernel (wgrad9.hip, the default) runs ONE instruction stream for every block -- narrow blocks contract stale operands --
  // so a tile costs the same whatever the block: equal slices (the remainder goes to the first blocks)
  static const bool old_kernel = [] {
    const char *a = getenv("SATNERF_WGRAD_V1"), *b = getenv("SATNERF_WGRAD_V2");
    return (a && a[0] == '1') || (b && b[0] == '1');
  }();
  if (!old_kernel) ca = 1.0, cb = 0.0, cc = 0.0;
#ifdef SR_PLAN_ENV
  if (const char* e = getenv("SR_WGRAD_COST")) sscanf(e, "%lf,%lf,%lf", &ca, &cb, &cc);
#endif
  int first = 0;
  static thread_local int sl[4096];
  const bool weighted = fmt == SR_FMT8 && n_blocks <= n_wg;
  blocks[kWgSpan] = 0;
  // r06: blocks whose row operand is one raw bf16 fragment (head rows: 1 row fragment, PHASE8 columns) run THIN streams in the 4-wave
  // kernel (gen/wgrad9_loop.py THIN, packing.wgrad9_thin_blocks): a tile of theirs costs a fraction of a full block's.  Relative costs
  // measured with tools/ab_wgrad8.py on sub-tables (profiles/r06_ab_variants.txt); SATNERF_WGRAD_THIN_COST="c16,c8" overrides them,
  // SATNERF_WGRAD_THIN=0 (full streams everywhere, packing reads the same switch) makes every block cost 1 again.
  double thin16 = 0.5, thin8 = 0.4;
  bool thin_on = true;
  if (const char* e = getenv("SATNERF_WGRAD_THIN")) thin_on = e[0] != '0';
  if (const char* e = getenv("SATNERF_WGRAD_THIN_COST")) sscanf(e, "%lf,%lf", &thin16, &thin8);
  bool any_thin = false;
  static thread_local double w9cost[4096];
  for (int b = 0; b < n_blocks; ++b) {
    const int32_t* t = blocks + kWgTableInts * b;
    const int nr = t[1] + t[3], nc = t[5] + t[7];
    const bool thin = thin_on && nr == 1 && t[8] == 1 && (nc == 8 || nc == 16);
    w9cost[b] = thin ? (nc == 16 ? thin16 : thin8) : 1.0;
    any_thin |= thin && w9cost[b] != 1.0;
  }
  bool use_weights = false;
  if (weighted && !old_kernel && any_thin && n_blocks <= 64 && n_wg >= n_blocks) {
    // cost-weighted slices for the 4-wave kernel: greedily to the block whose workgroups would finish last (minimises
    // max_b cost_b * ceil(tiles / slices_b)); the kernel numbers workgroups slice-major over the blocks that still have slices
    for (int b = 0; b < n_blocks; ++b) sl[b] = 1;
    for (int left = n_wg - n_blocks; left > 0; --left) {
      int worst = -1;
      double wt = -1;
      for (int b = 0; b < n_blocks; ++b) {
        const double tb = w9cost[b] * (double)((n_tiles + sl[b] - 1) / sl[b]);
        if (tb > wt && sl[b] < n_tiles) wt = tb, worst = b;
      }
      if (worst < 0) break;
      ++sl[worst];
    }
    double makespan = 0;
    for (int b = 0; b < n_blocks; ++b) {
      const double tb = w9cost[b] * (double)((n_tiles + sl[b] - 1) / sl[b]);
      makespan = tb > makespan ? tb : makespan;
    }
    // ... unless a stream-K plan (below: equal tile spans, thin blocks simply finish early) is shorter still -- width 512's 47 blocks
    const long span = ((long)n_blocks * n_tiles + n_wg - 1) / n_wg;
    use_weights = makespan <= (double)span * 1.03 || n_tiles * 208l * 1024l >= (1l << 32);
    if (const char* e = getenv("SATNERF_WGRAD_STREAMK")) use_weights = e[0] != '1';
  }
  if (use_weights) {
    // (sl[] holds the weighted split)
  } else if (weighted && !old_kernel) {
    // equal split, the remainder to the first blocks: what wgrad9.hip's workgroup numbering (slice-major, blocks 8 positions apart on one
    // XCD) assumes; a block never gets more slices than it has tiles
    const long q = n_wg / n_blocks, r = n_wg % n_blocks;
    for (int b = 0; b < n_blocks; ++b) {
      long v = q + (b < r ? 1 : 0);
      sl[b] = (int)(v > n_tiles ? n_tiles : v);
    }
    // ... unless that leaves the workgroups unevenly loaded (width 512: 47 blocks over 256 workgroups = 5 or 6 slices, 410 against 342
    // tiles): then stream-K -- the job list as one line of n_blocks x n_tiles tile units, `span` consecutive units per workgroup, a
    // workgroup that crosses a block boundary writes two partial blocks (wgrad9.hip).  Block b's slices = the workgroups whose span
    // touches it, numbered in tile order.  SATNERF_WGRAD_STREAMK = 0 / 1 forces the choice (A/B).
    const long total = (long)n_blocks * n_tiles, span = (total + n_wg - 1) / n_wg;
    const long worst = q > 0 ? (n_tiles + q - 1) / q : n_tiles;
    // ... and only where the 4-wave kernel will run it: workspaces it cannot address with 32-bit per-lane offsets (wgrad9_fits; <= 208
    // 1-KiB units per tile at either width) fall back to the r02 kernel, which takes equal slices only
    bool streamk = q > 0 && span >= 8 && worst * 100 > span * 103 && n_tiles * 208l * 1024l < (1l << 32);
    if (const char* e = getenv("SATNERF_WGRAD_STREAMK")) streamk = e[0] == '1' && span >= 1 && n_tiles * 208l * 1024l < (1l << 32);
    if (streamk) {
      for (int b = 0; b < n_blocks; ++b) sl[b] = (int)((((long)(b + 1) * n_tiles - 1) / span) - (((long)b * n_tiles) / span) + 1);
      blocks[kWgSpan] = (int32_t)span;
    }
  } else if (weighted) {
    double cost[4096];
    for (int b = 0; b < n_blocks; ++b) {
      const int32_t* t = blocks + kWgTableInts * b;
      cost[b] = ca + cb * (t[1] + t[3]) + cc * (t[5] + t[7]);
      sl[b] = 1;
    }
    for (int left = n_wg - n_blocks; left > 0; --left) {
      int worst = 0;
      double wt = -1;
      for (int b = 0; b < n_blocks; ++b) {
        const double tb = cost[b] * (double)((n_tiles + sl[b] - 1) / sl[b]);
        if (tb > wt && sl[b] < n_tiles) wt = tb, worst = b;
      }
      if (wt < 0) break;
      ++sl[worst];
    }
  }
  for (int b = 0; b < n_blocks; ++b) {
    int32_t* t = blocks + kWgTableInts * b;
    const int nr = t[1] + t[3], nc = t[5] + t[7];
    SR_REQUIRE(t[1] >= 1 && t[3] >= 0 && nr <= 16 && t[5] >= 0 && t[7] >= 0 && nc <= 16,
               "sr_wgrad_plan: block %d has %d row / %d column fragments (1..16 / 0..16)", b, nr, nc);
    const long mine = weighted ? sl[b] : per_block;
    t[kWgSlices] = (int)mine, t[kWgFirstSlice] = first, first += (int)mine;
  }
  *n_slices = first;
  return 0;
}

extern "C" int sr_satnerf_wgrad(int feat, int tau, int64_t n_points, const uint16_t* dpre, const uint16_t* acts, const int32_t* blocks,
                                int n_blocks, int n_slices, float* partial, void* stream) {
  SR_REQUIRE(feat == kFeat, "sr_satnerf_wgrad: feat=%d unsupported", feat);
  SR_REQUIRE(dpre && acts && blocks && partial, "sr_satnerf_wgrad: null pointer argument");
  SR_REQUIRE(n_blocks >= 1 && n_slices >= n_blocks, "sr_satnerf_wgrad: bad plan (%d blocks, %d slices): run sr_wgrad_plan first", n_blocks, n_slices);
  WgradParams p;
  p.dpre = (const uint4*)dpre, p.acts = (const uint4*)acts, p.blocks = blocks, p.partial = partial;
  p.n_tiles = (n_points + 31) / 32;
  p.n_blocks = n_blocks;
  p.auxs = aux_steps(tau);
  p.ak = act_ksteps(p.auxs);
  const size_t lds = (size_t)kSlots * kSlotBytes;
  if (!ensure_dynamic_lds((const void*)wgrad_kernel, lds)) return 1;
  hipLaunchKernelGGL(wgrad_kernel, dim3(n_slices), dim3(1024), lds, (hipStream_t)stream, p);
  return check_launch("wgrad_kernel");
}

extern "C" int sr_unpack_grads(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks,
                               float* grad, int accumulate, void* stream) {
  SR_REQUIRE(partial && gidx && gscale && grad && blocks, "sr_unpack_grads: null pointer");
  if (n_params <= 0) return 0;
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, gidx, gscale,
                     (long)n_params, blocks, grad, accumulate);
  return check_launch("unpack_grads_kernel");
}
