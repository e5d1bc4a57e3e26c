// Weight gradients of the fused Sat-NeRF MLP for gfx950:  dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Replaces autograd's `grad_weight = grad_output^T @ input` / `grad_bias = sum(grad_output)` of every nn.Linear in
// SatNeRF (models/satnerf.py:104-153).  Both operands live in HBM as B fragments (point-major: one 16-byte unit holds 8
// consecutive slots of ONE point, see mlp_layout.h) because that is how the forward / dX kernels hold them in registers;
// this contraction runs over POINTS, so both MFMA operands need 8 consecutive points of one slot instead.  The transpose
// is done by the LDS: fragments are staged point-major and read back with ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-point x 16-slot block and each lane receives one slot's 4 points), two reads per 32x32x16 MFMA operand.
//
// Grid = (job blocks, split-K slices).  A workgroup (8 waves) owns one job block of up to 256 x 256 (16 row fragments of
// dpre x 16 column fragments of the saved activations, packing.backward_maps lists them) PLUS the block's aux columns
// (biases, skip / sun / embedding columns) over a contiguous slice of 32-point tiles, so every operand fragment is read
// from HBM once per job.  Phase-coded activation fragments are decoded to bf16 sin() on the way into LDS.  fp32 partial
// blocks go to `partial[slice][block][256*256 + 256*32]`; sr_unpack_grads sums the slices and scatters into the flat
// gradient.
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

#ifdef SR_WG_TRACE
__device__ long long g_wg_trace[4096 * 8];
#define SR_T(k) do { const long long t_now = clock64(); tr[k] += t_now - t_last; t_last = t_now; } while (0)
#else
#define SR_T(k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup = 8 waves in a 4 x 2 grid; wave (wr, wc) owns rows 64*wr.. and columns 128*wc.. of the 256 x 256 block
// (2 x 4 MFMA tiles, 128 accumulator registers) plus the aux columns of row tile 2*wr + wc.
//
// Pipeline (MI355X measurements in profiles/r01_ab_variants.txt): per 32-point tile a workgroup moves 34 KiB and spends
// ~1150 MFMA cycles per SIMD, i.e. every CU must sustain ~10 B/clk from HBM; with operands staged through registers only
// two tiles (<= 68 KiB) per CU were in flight and the kernel sat at 4.1 TB/s on load latency.  Here the fragments go
// HBM -> LDS by LDS-DMA into a 4-slot ring (three tiles, ~100 KiB per CU, in flight, no staging registers); the wave that
// fetched a phase-coded fragment decodes it in place one tile ahead; one barrier per tile.
__global__ void __launch_bounds__(512) wgrad_kernel(const WgradParams prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* d = prm.blocks;  // find the job block this slice belongs to
  for (int b = 0; b + 1 < prm.n_blocks && (int)blockIdx.x >= d[kWgFirstSlice] + d[kWgSlices]; ++b) d += kWgTableInts;
  const int rf0 = d[0], nr0 = d[1], rf1 = d[2], nr = d[1] + d[3], cf0 = d[4], nc0 = d[5], cf1 = d[6], nc = d[5] + d[7], kind = d[8];
  const long tiles_per_split = (prm.n_tiles + d[kWgSlices] - 1) / d[kWgSlices];
  const long t_begin = (long)((int)blockIdx.x - d[kWgFirstSlice]) * tiles_per_split;
  long t_end = t_begin + tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;
  const int nt = t_end > t_begin ? (int)(t_end - t_begin) : 0;

  // staging: wave w moves row fragments w, w+8 and column fragments w, w+8; waves 0,1 also move the aux fragments
  auto row_frag = [&](int q) { return q < nr0 ? rf0 + q : rf1 + q - nr0; };  // fragment at row / column position q of the block
  auto col_frag = [&](int q) { return q < nc0 ? cf0 + q : cf1 + q - nc0; };
  const int fr0 = row_frag(wave < nr ? wave : nr - 1), fr1 = row_frag(wave + 8 < nr ? wave + 8 : nr - 1);
  const int fc0 = nc > 0 ? col_frag(wave < nc ? wave : nc - 1) : 0, fc1 = nc > 0 ? col_frag(wave + 8 < nc ? wave + 8 : nc - 1) : 0;
  const int fa = wave < prm.auxs ? wave : prm.auxs - 1;  // aux fragments are the first fragments of the activation tile
  const int src_unit = lane < 32 ? lane : 32 + ((lane - 8) & 31);  // LDS position `lane` <- this 16-byte unit of the fragment
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  // fragments past the block's edge are neither fetched nor decoded; their LDS image is stale data that only feeds output rows /
  // columns the epilogue masks
  const bool ld_r0 = wave < nr, ld_r1 = wave + 8 < nr, ld_c0 = wave < nc, ld_c1 = wave + 8 < nc, ld_ax = wave < 2;
  const int n_ld = (int)ld_r0 + (int)ld_r1 + (int)ld_c0 + (int)ld_c1 + (int)ld_ax;  // wave-uniform, 0..5
  auto tile_src = [&](long tile, const char*& dp, const char*& ac) {
#ifdef SR_ABL_L2HIT
    tile &= 15;  // every workgroup re-reads the same 16 tiles: loads hit in L2, HBM drops out
#endif
    dp = reinterpret_cast<const char*>(prm.dpre + tile * kDpFrags * 64 + src_unit);
    ac = reinterpret_cast<const char*>(prm.acts + tile * prm.ak * 64 + src_unit);
  };
  auto issue_piece = [&](int j, const char* dp, const char* ac, int slot) {  // j is a compile-time constant at every call
    const uint32_t base = ring + slot * kSlotBytes + wave * kFragStride;
    if (j == 0 && ld_r0) glds16(dp + fr0 * 1024, base);
    if (j == 1 && ld_r1) glds16(dp + fr1 * 1024, base + 8 * kFragStride);
    if (j == 2 && ld_c0) glds16(ac + fc0 * 1024, base + 16 * kFragStride);
    if (j == 3 && ld_c1) glds16(ac + fc1 * 1024, base + 24 * kFragStride);
    if (j == 4 && ld_ax) glds16(ac + fa * 1024, base + 32 * kFragStride);
  };
  auto issue = [&](long tile, int slot) {
    const char *dp, *ac;
    tile_src(tile, dp, ac);
#pragma unroll
    for (int j = 0; j < 5; ++j) issue_piece(j, dp, ac, slot);
  };
  // own DMA of the oldest tile has landed when at most `c` younger tiles (n_ld loads each) are outstanding
  auto wait_tiles = [&](int c) {
    switch (c * n_ld) {
      case 15: wait_vm<15>(); break;
      case 12: wait_vm<12>(); break;
      case 10: wait_vm<10>(); break;
      case 9: wait_vm<9>(); break;
      case 8: wait_vm<8>(); break;
      case 6: wait_vm<6>(); break;
      case 5: wait_vm<5>(); break;
      case 4: wait_vm<4>(); break;
      case 3: wait_vm<3>(); break;
      case 2: wait_vm<2>(); break;
      case 1: wait_vm<1>(); break;
      default: wait_vm<0>(); break;
    }
  };
  auto decode = [&](int slot) {  // phase-coded sin stage -> bf16 activation values, in place, own column fragments only
#ifndef SR_ABL_NODECODE
    if (kind != 1) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!(k ? ld_c1 : ld_c0)) continue;
      uint4* p = reinterpret_cast<uint4*>(lds + slot * kSlotBytes + (16 + 8 * k + wave) * kFragStride + lane * 16);
      const uint4 v = *p;
      *p = make_uint4(phase_pair_to_bf16(v.x), phase_pair_to_bf16(v.y), phase_pair_to_bf16(v.z), phase_pair_to_bf16(v.w));
    }
#endif
  };

  // transposed operand reads: lane = (hh, rh, m, q): MFMA row/col = 16*rh + 4*q + e, k = 8*hh + 4*rd + m <-> point 16*ks + k
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  int rd_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int point = 16 * ks + 8 * hh + 4 * rd + m;
      rd_off[ks][rd] = rh * kFragStride + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8;
    }
  const int wr = wave >> 1, wc = wave & 1;
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
  // MFMA tiles of this wave that intersect the block: row tiles 64*wr + 32*{0,1}, column tiles 128*wc + 32*{0..3}
  auto clampi = [](int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; };
  const int n_rt = clampi((16 * nr - 64 * wr + 31) / 32, 2), n_ct = clampi((16 * nc - 128 * wc + 31) / 32, 4);
  const bool aux_on = wc ? n_rt == 2 : n_rt >= 1;  // aux columns of row tile 2*wr + wc
  const bool full = n_rt == 2 && n_ct == 4;

  auto rendezvous = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // The whole tile loop + epilogue is instantiated twice (a wave picks one; both execute the same barriers) so that the hot
  // full-tile path keeps straight-line code and its own register allocation.
  //
  // Iteration i (tile i in slot i & 3; tiles i+1, i+2 landed or in flight; slot (i+3) & 3 was released by the previous
  // rendezvous):  k-step 0 of tile i with the five LDS-DMA pieces of tile i+3 issued between its MFMAs (a piece stalls the
  // issuing wave for 60-180 cycles, which the queued MFMAs cover) -> wait for this wave's pieces of tile i+1 -> k-step 1 with
  // the in-place phase decode of tile i+1 spread between its MFMAs -> rendezvous.
  auto run = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
    f32x16 acc[2][4] = {}, acc_aux = {};
    struct Ops {
      uint4 a0, a1, bc[4], bx;
    };
    // Full path: the 14 transposed reads of a k-step are issued together, and both k-steps of a tile before the first MFMA --
    // with two waves per SIMD the LDS latency is only hidden by the reads a wave keeps in flight itself
    // (MI355X_MICROARCH.md, LDS: >= 16 DS operations per wait).
    auto load_ops = [&](const char* b, int ks) {
      Ops o;
      o.a0 = operand(b, 2 * wr, ks), o.a1 = operand(b, 2 * wr + 1, ks);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) o.bc[ct] = operand(b, 8 + 4 * wc + ct, ks);
      o.bx = operand(b, 16, ks);  // aux fragments 32, 33 = fragment pair 16
      return o;
    };
    auto mma_ops = [&](const Ops& o, auto&& filler) {  // filler(j), j = 0..4, runs after MFMA pair j
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#ifndef SR_ABL_NOMFMA
        acc[0][ct] = mma(o.a0, o.bc[ct], acc[0][ct]);
        acc[1][ct] = mma(o.a1, o.bc[ct], acc[1][ct]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        filler(ct);
        __builtin_amdgcn_sched_barrier(0);
      }
      acc_aux = mma(wc ? o.a1 : o.a0, o.bx, acc_aux);  // wave-uniform select
      __builtin_amdgcn_sched_barrier(0);
      filler(4);
    };
    // Edge blocks (heads narrower than 256): only the live tiles, under wave-uniform branches
    auto kstep_edge = [&](const char* b, int ks) {
      if (n_rt == 0) return;
      const uint4 a0 = operand(b, 2 * wr, ks);
      uint4 a1 = a0;
      if (n_rt > 1) a1 = operand(b, 2 * wr + 1, ks);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (ct < n_ct) {
          const uint4 bc = operand(b, 8 + 4 * wc + ct, ks);
          acc[0][ct] = mma(a0, bc, acc[0][ct]);
          if (n_rt > 1) acc[1][ct] = mma(a1, bc, acc[1][ct]);
        }
      }
      if (aux_on) acc_aux = mma(wc ? a1 : a0, operand(b, 16, ks), acc_aux);
    };
    auto kstep = [&](const char* b, auto ks_tag, auto&& filler) {
      constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
      for (int j = 0; j < 5; ++j) filler(j);
      kstep_edge(b, ks);
    };
    for (int i = 0; i < kSlots - 1 && i < nt; ++i) issue(t_begin + i, i);
    if (nt > 0) {
      wait_tiles((nt < kSlots - 1 ? nt : kSlots - 1) - 1);
      decode(0);
      rendezvous();
    }
#ifdef SR_WG_TRACE
    long long tr[6] = {0, 0, 0, 0, 0, 0}, t_last = clock64();
#endif
    for (int i = 0; i < nt; ++i) {
      const char* cur = lds + (i & (kSlots - 1)) * kSlotBytes;
      const int slot3 = (i + kSlots - 1) & (kSlots - 1);
      bool fetch = i + kSlots - 1 < nt;
#ifdef SR_ABL_NOLOAD
      fetch = false;
#endif
      const char *dp, *ac;
      tile_src(t_begin + (fetch ? i + kSlots - 1 : i), dp, ac);
      Ops o0, o1;
      if constexpr (kFull) {
        o0 = load_ops(cur, 0), o1 = load_ops(cur, 1);
        mma_ops(o0, [&](int j) {
          if (fetch) issue_piece(j, dp, ac, slot3);
        });
      } else {
        kstep(cur, std::integral_constant<int, 0>{}, [&](int j) {
          if (fetch) issue_piece(j, dp, ac, slot3);
        });
      }
      SR_T(0);
      bool dec = false;
      if (i + 1 < nt) {  // tiles i+1 .. min(nt-1, i+3) have been issued; this wave's pieces of tile i+1 must have landed
        const int last = nt - 1 < i + kSlots - 1 ? nt - 1 : i + kSlots - 1;
        wait_tiles(last - (i + 1));
        SR_T(1);
#ifndef SR_ABL_NODECODE
        dec = kind == 1;
#endif
      }
      // k-step 1 with the decode of this wave's two column fragments of tile i+1 spread over the fillers
      char* next = lds + ((i + 1) & (kSlots - 1)) * kSlotBytes;
      uint4* p0 = reinterpret_cast<uint4*>(next + (16 + wave) * kFragStride + lane * 16);
      uint4* p1 = reinterpret_cast<uint4*>(next + (24 + wave) * kFragStride + lane * 16);
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
      const bool dec0 = dec && ld_c0, dec1 = dec && ld_c1;
      if (dec0) v0 = *p0;
      if (dec1) v1 = *p1;
      auto decode_part = [&](int j) {
        if (j == 0 && dec0) v0.x = phase_pair_to_bf16(v0.x), v0.y = phase_pair_to_bf16(v0.y);
        if (j == 1 && dec0) v0.z = phase_pair_to_bf16(v0.z), v0.w = phase_pair_to_bf16(v0.w);
        if (j == 2 && dec1) v1.x = phase_pair_to_bf16(v1.x), v1.y = phase_pair_to_bf16(v1.y);
        if (j == 3 && dec1) v1.z = phase_pair_to_bf16(v1.z), v1.w = phase_pair_to_bf16(v1.w);
      };
      if constexpr (kFull) mma_ops(o1, decode_part);
      else kstep(cur, std::integral_constant<int, 1>{}, decode_part);
      if (dec0) *p0 = v0;
      if (dec1) *p1 = v1;
      SR_T(2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SR_T(3);
      rendezvous();  // every wave is done with tile i's slot; tile i+1 is decoded and visible
      SR_T(4);
    }
#ifdef SR_WG_TRACE
    if (lane == 0 && blockIdx.x < 512) {
      long long* o = g_wg_trace + (blockIdx.x * 8 + wave) * 8;
      for (int k = 0; k < 5; ++k) o[k] = tr[k];
      o[5] = nt;
    }
#endif

    float* out = prm.partial + (long)blockIdx.x * kBlockFloats;
    const int n_rows = 16 * nr, n_cols = 16 * nc;
    if (n_rt > 0 && n_ct > 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int row = 64 * wr + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh;
            const int col = 128 * wc + 32 * ct + (lane & 31);
            if (row < n_rows && col < n_cols) out[row * 256 + col] = acc[rt][ct][g];
          }
    }
    float* oa = out + 256 * 256;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int row = 64 * wr + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
      if (row < n_rows) oa[row * 32 + (lane & 31)] = acc_aux[g];
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
extern "C" int sr_wgrad_plan(int32_t* blocks, int n_blocks, int64_t n_points, int n_wg, int* n_slices) {
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
  int first = 0;
  for (int b = 0; b < n_blocks; ++b) {
    int32_t* t = blocks + kWgTableInts * b;
    const int nr = t[1] + t[3], nc = t[5] + t[7];
    SR_REQUIRE(t[1] >= 1 && t[3] >= 0 && nr <= 16 && t[5] >= 0 && t[7] >= 0 && nc <= 16,
               "sr_wgrad_plan: block %d has %d row / %d column fragments (1..16 / 0..16)", b, nr, nc);
    t[kWgSlices] = (int)per_block, t[kWgFirstSlice] = first, first += (int)per_block;
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
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3(n_slices), dim3(512), lds, (hipStream_t)stream, p);
#ifdef SR_WG_TRACE
  static int trace_calls = 0;
  if (++trace_calls == 30) {
    static long long host[4096 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_trace), sizeof(host));
    for (int wg = 0; wg < n_slices && wg < 512; wg += 7) {
      for (int w = 0; w < 8; w += 7) {
        const long long* o = host + (wg * 8 + w) * 8;
        const double n = o[5] > 0 ? (double)o[5] : 1.0;
        printf("wg %3d wave %d tiles %lld: ks0+dma %.0f  wait_vm %.0f  ks1+decode %.0f  lgkm %.0f  barrier %.0f  (clock64 ticks per tile)\n", wg, w, o[5],
               o[0] / n, o[1] / n, o[2] / n, o[3] / n, o[4] / n);
      }
    }
  }
#endif
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
