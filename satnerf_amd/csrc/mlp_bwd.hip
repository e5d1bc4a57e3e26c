// Fused Sat-NeRF MLP backward, data-gradient chain ("dX") for gfx950.
//
// Replaces autograd's backward through SatNeRF.forward (models/satnerf.py:156-208) for every sample point: starting
// from the gradients of the four per-point outputs it walks the network in reverse, register-resident exactly like the
// forward kernel (swapped orientation, 32 points per wave, transposed weights streamed L2 -> LDS ring, see mlp_layout.h),
// and writes the gradient of EVERY pre-activation, as bf16 B fragments, to the `dpre` workspace; the weight-gradient GEMMs
// (wgrad.hip) then contract dpre with the activations the forward pass saved.  No gradient flows to xyz / z / rays
// (inputs are data, rendering.py:122-124 detaches the resampled depths), so fc_net.0 needs no dX.
//
// sin stages: the forward saved the pre-activation PHASE (unorm16 revolutions); d pre = d out * cos(2 pi phase) (for
// fc_net.0 the factor w0 = 30 is applied to its weight gradient by the gather scale, packing.backward_maps).
// Arithmetic: single-pass bf16 MFMA with fp32 accumulation (mixed-precision backward) in both numeric modes.
// Workspace format (template FMT, mlp_layout.h): SR_FMT16 reads unorm16 phases and writes bf16 gradients; SR_FMT8 (the
// throughput mode) reads PHASE8 double fragments and writes the gradients as MX8 (codec8.h) -- the gradient that continues down
// the chain stays bf16 in registers either way, only the copy the weight-gradient kernel reads is 8-bit.
#include "mlp_device.h"
#include "mlp_params.h"
#include "codec8.h"

namespace sr {

using BS = BwdStream;
constexpr int kBSlot = BS::SLOTP * 1024;  // ring slot bytes

template <int G>
__device__ __forceinline__ void bchunk_enter(char* ring, const char* stream, int wave, int lane) {
  constexpr int later = [] {
    int n = 0;
    for (int c = 1; c < kD; ++c) n += min_loads<1>(BS::np(G + c));
    return n;
  }();
  wait_then_barrier<later>();
  if constexpr (G + kD < BS::NCH) {
    constexpr long off = BS::offset_pieces(G + kD) * 1024L;
    constexpr int slot = (G + kD) % kNSLOT;
    issue_chunk<1, BS::np(G + kD)>(stream, nullptr, off, ring + slot * kBSlot, wave, lane);
  }
}

__device__ __forceinline__ float phase_cos(uint32_t w, int half) {
  const uint32_t u = half ? (w >> 16) : (w & 0xffffu);
  return __builtin_amdgcn_cosf((float)u * (1.0f / 65535.0f));  // v_cos_f32 takes revolutions
}

// phases of one output tile's 16 values per lane: SR_FMT16 two unorm16 fragments, SR_FMT8 one PHASE8 double fragment
template <int FMT>
struct Phase {
  uint4 p0, p1;
  __device__ __forceinline__ void load(const uint4* acts_tile, int auxs, int frag) {  // frag = logical fragment of value 0 (even)
    if constexpr (FMT == SR_FMT8) p0 = ws_load(acts_tile + (auxs + (frag >> 1)) * 64);
    else p0 = ws_load(acts_tile + (auxs + frag) * 64), p1 = ws_load(acts_tile + (auxs + frag + 1) * 64);
  }
  __device__ __forceinline__ void from_lds(const char* slot, int lane) {  // staged by stage_phase()
    p0 = *reinterpret_cast<const uint4*>(slot + lane * 16);
    if constexpr (FMT != SR_FMT8) p1 = *reinterpret_cast<const uint4*>(slot + 1024 + lane * 16);
  }
  __device__ __forceinline__ float cos(int g) const {
    if constexpr (FMT == SR_FMT8) {
      const uint32_t pw[4] = {p0.x, p0.y, p0.z, p0.w};
      return __builtin_amdgcn_cosf(phase8_rev(pw[g >> 2], g & 3));
    } else {
      const uint32_t pw[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
      return phase_cos(pw[g >> 1], g & 1);
    }
  }
};

// 16 accumulator values (fragments 2T, 2T+1 of the stage's input vector) -> [x cos(phase)] -> two bf16 fragments (the next
// stage's B operand), stored to the dpre workspace at logical fragment `frag` in the format FMT; returns the MX8 scale byte.
constexpr int dp8_unit(int frag) { return frag < kDpSigma ? frag >> 1 : (frag - 1) >> 1; }  // mlp_layout.h: d_sigma_pre sits between
template <bool COS, int FMT>
__device__ __forceinline__ uint32_t bpack(const f32x16& acc, const Phase<FMT>& ph, uint4& o0, uint4& o1, uint4* dpre_tile, int frag) {
  float v[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) v[g] = COS ? acc[g] * ph.cos(g) : acc[g];
  o0 = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  o1 = make_uint4(pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15]));
  if constexpr (FMT == SR_FMT8) {
    const uint32_t e = mx8_exponent(v);
    ws_store(dpre_tile + dp8_unit(frag) * 64, mx8_encode(v, e));
    return e;
  } else {
    ws_store(dpre_tile + frag * 64, o0);
    ws_store(dpre_tile + (frag + 1) * 64, o1);
    return 0u;
  }
}
// scale bytes of a group's tiles (SR_FMT8): byte (group & 1) * 8 + t of the lane's 16 B in unit kD8Scale + (group >> 1)
template <int FMT, int NT>
__device__ __forceinline__ void store_scales(uint4* dpre_tile, int group, const uint32_t (&eb)[2]) {
  if constexpr (FMT == SR_FMT8) {
    char* p = reinterpret_cast<char*>(dpre_tile + (kD8Scale + (group >> 1)) * 64) + (group & 1) * 8;
    if constexpr (NT > 4) *reinterpret_cast<uint2*>(p) = make_uint2(eb[0], eb[1]);
    else *reinterpret_cast<uint32_t*>(p) = eb[0];
  }
}
constexpr int dp8_group(int frag) { return frag < kDpFeats ? frag / 16 : frag < kDpSigma ? 8 : 9 + (frag - kDpRgbh) / kHS; }

// one output tile: acc = sum_i A(piece P0+i of the slot) x in[i]
template <int KIN>
__device__ __forceinline__ f32x16 btile(const char* slot, int p0, const uint4 (&in)[KIN], int lane) {
  f32x16 acc = {0};
#pragma unroll
  for (int i = 0; i < KIN; ++i) {
    const uint4 a = *reinterpret_cast<const uint4*>(slot + (p0 + i) * 1024 + lane * 16);
    acc = mfma(a, in[i], acc);
  }
  return acc;
}

// ---- trunk schedule ---------------------------------------------------------------------------------------------------------
// In the trunk (7 layers x 8 output tiles, two thirds of the kernel) every vector-memory LOAD of a wave is an LDS-DMA issued
// through inline asm -- the weight chunks into the shared ring, the tile's saved phases into a wave-private staging ring -- so
// the compiler never sees a load it would have to wait for (left to hipcc, each tile's phase load became an s_waitcnt vmcnt(0)
// that also drained the weight prefetch and the workspace stores: 62 % of the wave cycles were spent parked, PMC r02a).  The
// waits are ours and counted: LOADS retire in order, the instruction stream of a wave is fixed, so the number of vector-memory
// loads issued after the one being waited for is a compile-time constant of the position in the layer (stores share the counter
// but complete out of order with respect to loads: they are left out of the count, which can only make a wait longer).
// Per layer (NP = phase DMAs per tile, NS = workspace stores per tile, ND = weight DMAs per group of 2 chunks):
//   tile t:   [t even: wait A(t), s_barrier, ND weight DMAs for chunks t+2, t+3]  NP phase DMAs for tile t+2  MFMAs(t)
//             [t > 0: wait B(t-1), epilogue(t-1) = cos, pack, encode, NS stores]
//   end:      wait B(7), epilogue(7), scale store (NSS)
template <int NP, int NS, int NSS>
struct TrunkSched {
  static constexpr int ND = 2 * (kKS / 8);  // 2 chunks x 16 pieces / 8 waves
  static constexpr int tile_len(int t) { return (t % 2 == 0 ? ND : 0) + NP + (t > 0 ? NS : 0); }
  static constexpr int tile_start(int t) {
    int p = 0;
    for (int k = 0; k < t; ++k) p += tile_len(k);
    return p;
  }
  static constexpr int period() { return tile_start(kMT) + NS + NSS; }
  static constexpr int after_d(int t) { return tile_start(t) + ND; }                          // just after the weight DMAs of tile t (even)
  static constexpr int after_p(int t) { return tile_start(t) + (t % 2 == 0 ? ND : 0) + NP; }  // just after the phase DMAs issued in tile t
  static constexpr int at_store(int t) { return t < kMT - 1 ? after_p(t + 1) : tile_start(kMT); }  // where epilogue(t)'s wait sits
  // instructions issued after the phase DMAs of tile t (issued in tile t-2; in the previous layer -- or the prologue -- for t < 2)
  static constexpr int wait_b(int t) {
    if (t >= 2) return at_store(t) - after_p(t - 2);
    const int steady = at_store(t) - (after_p(t + kMT - 2) - period());
    const int first = t == 0 ? NP + at_store(0) : at_store(1);  // first layer: phases of tiles 0, 1 are staged by the prologue
    return steady < first ? steady : first;
  }
  // instructions issued after the weight DMAs of tile t's group (issued at the entry of tile t-2)
  static constexpr int wait_a(int t) { return t >= 2 ? tile_start(t) - after_d(t - 2) : tile_start(0) - (after_d(kMT - 2) - period()); }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- phases of the stages before the trunk ---------------------------------------------------------------------------------
// The 28 output tiles before the trunk that multiply by cos(phase) -- bH (12: rgb hidden, sun hidden 3, beta hidden), bS3 (4), bS2
// (4), bG1 (8) -- take their phases the same way: LDS-DMA into the wave's 4-slot staging ring TWO phase tiles ahead (tile k + 2
// is requested right after tile k is used; tiles 0, 1 by the prologue), counted wait before the use.  The instruction stream is
// static, so the loads issued between a request and its use are known: the phase DMAs of tile k + 1 and the weight DMAs of every
// chunk entered in between (at least min_loads per wave).  PreSched replays that stream at compile time.
struct PreSched {
  static constexpr int kTiles = 3 * kMTH + 2 * 2 + 2 * 2 + kMT;  // 28
  // phase tile k -> logical activation fragment of its first value
  static constexpr int frag(int k) {
    if (k < 3 * kMTH) {
      const int part = k / kMTH, t = k % kMTH;
      return (part == 0 ? kActRgbh : part == 1 ? kActS3 : kActE1) + 2 * t;
    }
    k -= 3 * kMTH;
    if (k < 4) return kActS2 + 2 * k;  // bS3 multiplies by cos(phase s2)
    k -= 4;
    if (k < 4) return kActS1 + 2 * k;  // bS2: cos(phase s1)
    return kActA0 + 7 * kKS + 2 * (k - 4);  // bG1: cos(phase a7)
  }
  static constexpr int chunk_of(int k) {  // the chunk in which phase tile k is used
    if (k < 3 * kMTH) return BS::G_H;
    k -= 3 * kMTH;
    if (k < 4) return BS::G_S3 + k / 2;
    k -= 4;
    if (k < 4) return BS::G_S2 + k / 2;
    return BS::G_G1 + (k - 4);
  }
  // loads (lower bound per wave) issued after the request of phase tile k and before its use; nph = phase DMAs per tile.
  // Request points: tiles 0, 1 in the prologue (after the DMAs of chunks 0..2, before chunk 0 is entered); tile k >= 2 right after
  // the use of tile k - 2, i.e. inside chunk_of(k - 2).  Entering chunk g requests chunk g + kD.
  static constexpr int wait(int k, int nph) {
    int n = 0;
    const int req_chunk = k >= 2 ? chunk_of(k - 2) : -1;
    for (int g = req_chunk + 1; g <= chunk_of(k); ++g) n += min_loads<1>(BS::np(g + kD));  // chunks entered after the request
    if (k + 1 < kTiles) n += nph;  // tile k + 1 is requested in between (after the use of tile k - 1; tile 1 right after tile 0)
    return n;
  }
};

// generic transposed stage: NCHUNK chunks of TPC tiles, each KIN pieces; tile t -> out[2t], out[2t+1], stored to the dpre
// workspace from logical fragment DF0; AF0 = logical activation fragment of the stage's phases (COS stages)
// PK0 = phase-tile index (PreSched) of the stage's first tile; `stage` = the wave's phase staging ring, request(slot, frag) = LDS-DMA
// of a phase tile into it
template <int FMT, int KIN, int TPC, int NCHUNK, int G0, bool COS, int NOUT, int PK0, int DF0, class Req>
__device__ __forceinline__ void bstage(const uint4 (&in)[KIN], uint4 (&out)[NOUT], char* ring, const char* stream, int wave, int lane,
                                       const char* stage, Req&& request, uint4* dpre_tile) {
  constexpr int NPH = FMT == SR_FMT8 ? 1 : 2;
  uint32_t eb[2] = {0u, 0u};
  static_for<NCHUNK>([&](auto cc) {
    constexpr int c = decltype(cc)::value, g = G0 + c;
    static_assert(BS::np(g) == TPC * KIN, "backward stream geometry mismatch");
    bchunk_enter<g>(ring, stream, wave, lane);
    const char* slot = ring + (g % kNSLOT) * kBSlot;
    static_for<TPC>([&](auto tc) {
      constexpr int tt = decltype(tc)::value, t = c * TPC + tt, o = 2 * t;
      Phase<FMT> ph = {};
      const f32x16 acc = btile<KIN>(slot, tt * KIN, in, lane);
      if constexpr (COS) {
        constexpr int k = PK0 + t;
        wait_vmcnt<PreSched::wait(k, NPH)>();
        ph.from_lds(stage + (k & 3) * NPH * 1024, lane);
      }
      eb[t >> 2] |= bpack<COS, FMT>(acc, ph, out[o], out[o + 1], dpre_tile, DF0 + 2 * t) << (8 * (t & 3));
      if constexpr (COS && PK0 + t + 2 < PreSched::kTiles) request((PK0 + t + 2) & 3, PreSched::frag(PK0 + t + 2));
    });
  });
  store_scales<FMT, TPC * NCHUNK>(dpre_tile, dp8_group(DF0), eb);
}

template <int FMT>
__global__ void __launch_bounds__(512) satnerf_bwd_kernel(const BwdParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, pl = lane & 31;
  const char* stream = prm.stream;
  const int A = prm.auxs, AK = FMT == SR_FMT8 ? act8_units(prm.auxs) : act_ksteps(prm.auxs);
  constexpr int DK = FMT == SR_FMT8 ? kD8Units : kDpFrags;  // workspace units per tile

#pragma unroll
  for (int g = 0; g < kD; ++g) {
    if (g == 0) issue_chunk<1, BS::np(0)>(stream, nullptr, 0, ring, wave, lane);
    if (g == 1) issue_chunk<1, BS::np(1)>(stream, nullptr, BS::offset_pieces(1) * 1024L, ring + kBSlot, wave, lane);
    if (g == 2) issue_chunk<1, BS::np(2)>(stream, nullptr, BS::offset_pieces(2) * 1024L, ring + 2 * kBSlot, wave, lane);
  }

#ifdef SR_BWD_REVERSE
  const long tile = (long)(gridDim.x - 1 - blockIdx.x) * 8 + wave;  // newest activations first: whatever the Infinity Cache kept
#else
  const long tile = (long)blockIdx.x * 8 + wave;
#endif
  const long pt = tile * 32 + pl;
  const bool valid = pt < prm.n_points;
  const uint4* acts = prm.acts + tile * AK * 64 + lane;   // + unit * 64
  uint4* dpre = prm.dpre + tile * DK * 64 + lane;         // + unit * 64

  // ---- gradients of the head pre-activations (rows 0..2 albedo logits, 3 sun logit, 4 beta; sigma separately) ----------
  uint4 dhead[1], dsig;
  {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sg = 0.f;
    if (valid) {
      if (h == 0) {
        if (prm.g_albedo) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float s = (prm.albedo[pt * 3 + c] + 0.001f) * (1.0f / 1.002f);  // sigmoid output before the rgb_padding affine
            v[c] = prm.g_albedo[pt * 3 + c] * 1.002f * s * (1.0f - s);
          }
        }
        if (prm.g_sun) {
          const float s = prm.sun_v[pt];
          v[3] = prm.g_sun[pt] * s * (1.0f - s);
        }
        if (prm.g_sigma) sg = prm.g_sigma[pt] * (1.0f - expf(-prm.sigma[pt]));  // softplus' = 1 - exp(-softplus)
      } else if (prm.g_beta) {
        v[0] = prm.g_beta[pt] * (1.0f - expf(-prm.beta[pt]));
      }
    }
    dhead[0] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    dsig = make_uint4(pack_bf16x2(sg, 0.f), 0u, 0u, 0u);
    dpre[(FMT == SR_FMT8 ? kD8Head : kDpHead) * 64] = dhead[0];
    dpre[(FMT == SR_FMT8 ? kD8Sigma : kDpSigma) * 64] = dsig;
  }

  // this wave's phase staging ring (4 tiles) behind the weight ring; the first two phase tiles are requested now
  constexpr int NPH = FMT == SR_FMT8 ? 1 : 2;  // phase units per tile
  char* stage = smem + kNSLOT * kBSlot + wave * (4 * NPH * 1024);
  const uint32_t stage_addr = __builtin_amdgcn_readfirstlane(lds_addr_of(stage));
  auto stage_phase = [&](int slot, int frag) {  // frag = logical activation fragment of the tile's first value
    const char* src = reinterpret_cast<const char*>(acts + (A + (FMT == SR_FMT8 ? frag >> 1 : frag)) * 64);
#pragma unroll
    for (int k = 0; k < NPH; ++k) glds16(src + k * 1024, stage_addr + (slot * NPH + k) * 1024);
  };
  stage_phase(0, PreSched::frag(0)), stage_phase(1, PreSched::frag(1));

  // ---- bH: d_head -> d rgb-hidden | d sun-hidden-3 | d beta-hidden (12 tiles x 1 piece, one chunk) --------------------
  uint4 d_rgbh[kHS], d_s3[kHS], d_e1[kHS];
  {
    bchunk_enter<BS::G_H>(ring, stream, wave, lane);
    const char* slot = ring + (BS::G_H % kNSLOT) * kBSlot;
    uint32_t eb[3][2] = {};
    static_for<3 * kMTH>([&](auto tc) {
      constexpr int T = decltype(tc)::value, part = T / kMTH, t = T % kMTH;
      constexpr int dp0 = part == 0 ? kDpRgbh : (part == 1 ? kDpS3 : kDpE1);
      const f32x16 acc = btile<1>(slot, T, dhead, lane);
      wait_vmcnt<PreSched::wait(T, NPH)>();
      Phase<FMT> ph;
      ph.from_lds(stage + (T & 3) * NPH * 1024, lane);
      uint4 o0, o1;
      eb[part][0] |= bpack<true, FMT>(acc, ph, o0, o1, dpre, dp0 + 2 * t) << (8 * t);
      stage_phase((T + 2) & 3, PreSched::frag(T + 2));
      if constexpr (part == 0) d_rgbh[2 * t] = o0, d_rgbh[2 * t + 1] = o1;
      else if constexpr (part == 1) d_s3[2 * t] = o0, d_s3[2 * t + 1] = o1;
      else d_e1[2 * t] = o0, d_e1[2 * t + 1] = o1;
    });
    store_scales<FMT, kMTH>(dpre, dp8_group(kDpRgbh), eb[0]);
    store_scales<FMT, kMTH>(dpre, dp8_group(kDpS3), eb[1]);
    store_scales<FMT, kMTH>(dpre, dp8_group(kDpE1), eb[2]);
  }
  // ---- sun chain: bS3 (d s3 -> d s2), bS2 (d s2 -> d s1) -------------------------------------------------------------
  uint4 d_g2[3 * kHS];  // [d rgbh | d s1 | d e1] : the input of bG2
  {
    uint4 d_s2[kHS], d_s1[kHS];
    bstage<FMT, kHS, 2, 2, BS::G_S3, true, kHS, 3 * kMTH, kDpS2>(d_s3, d_s2, ring, stream, wave, lane, stage, stage_phase, dpre);
    bstage<FMT, kHS, 2, 2, BS::G_S2, true, kHS, 3 * kMTH + 4, kDpS1>(d_s2, d_s1, ring, stream, wave, lane, stage, stage_phase, dpre);
#pragma unroll
    for (int i = 0; i < kHS; ++i) d_g2[i] = d_rgbh[i], d_g2[kHS + i] = d_s1[i], d_g2[2 * kHS + i] = d_e1[i];
  }
  // ---- bG2: -> d feats (identity stage), bDT: d beta-hidden -> d t (embedding columns) ------------------------------
  uint4 d_g1[kKS + 1];  // [d feats (16) | d sigma_pre (1)] : the input of bG1
  {
    uint4 d_feats[kKS];
    bstage<FMT, 3 * kHS, 1, kMT, BS::G_G2, false, kKS, -1, kDpFeats>(d_g2, d_feats, ring, stream, wave, lane, stage, stage_phase, dpre);
    bchunk_enter<BS::G_DT>(ring, stream, wave, lane);
    const f32x16 acc = btile<kHS>(ring + (BS::G_DT % kNSLOT) * kBSlot, 0, d_e1, lane);
    if (valid && prm.d_t) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = (g & 3) + 8 * (g >> 2) + 4 * h;  // = embedding component
        if (row < prm.tau) prm.d_t[pt * prm.tau + row] = acc[g];
      }
    }
#pragma unroll
    for (int i = 0; i < kKS; ++i) d_g1[i] = d_feats[i];
    d_g1[kKS] = dsig;
  }
  // ---- bG1: -> d a7, x cos(phase a7) = d pre_7 ----------------------------------------------------------------------
  uint4 cur[kKS], nxt[kKS];
  bstage<FMT, kKS + 1, 1, kMT, BS::G_G1, true, kKS, 3 * kMTH + 8, kDpL + 7 * kKS>(d_g1, cur, ring, stream, wave, lane, stage, stage_phase, dpre);
  // ---- bL7 .. bL1: d pre_l -> d a_{l-1}, x cos(phase a_{l-1}) = d pre_{l-1}; scheduled by TrunkSched (above) -------------------
  constexpr long offL = BS::offset_pieces(BS::G_L);
  // stores are NOT counted (NS = NSS = 0): loads retire in order among themselves, but a store may complete before an older load,
  // so only "younger LOADS still outstanding" proves that an older load has landed; the price is an occasional wait for a store
  using TS = TrunkSched<NPH, 0, 0>;  // (counting the stores as well measured the same: 138.3 vs 139.6 us)
  // G1's chunk protocol already requested the first trunk chunks; drain everything once and start from a known queue
  wait_then_barrier<0>();
  stage_phase(0, kActA0 + 6 * kKS), stage_phase(1, kActA0 + 6 * kKS + 2);
#pragma unroll 1
  for (int l = 7; l >= 1; --l) {
    const long cbase = (long)(7 - l) * kMT;
    const int lfrag = (l - 1) * kKS;                    // logical fragment of a_{l-1} / d_pre_{l-1}
    const int nfrag = l >= 2 ? lfrag - kKS : lfrag;     // ... of the next layer (the last layer re-stages its own: uniform counts)
    f32x16 acc[2];
    uint32_t eb[2] = {0u, 0u};
    auto epilogue = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      wait_vmcnt<TS::wait_b(t)>();
      Phase<FMT> ph;
      ph.from_lds(stage + (t & 3) * NPH * 1024, lane);
      eb[t >> 2] |= bpack<true, FMT>(acc[t & 1], ph, nxt[2 * t], nxt[2 * t + 1], dpre, kDpL + lfrag + 2 * t) << (8 * (t & 3));
    };
    static_for<kMT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t % 2 == 0) {
        wait_then_barrier<TS::wait_a(t)>();
#pragma unroll
        for (int k = 2; k < 4; ++k) {  // chunks t+2, t+3 (clamped at the end of the stream: same instruction count every layer)
          long c = cbase + t + k;
          c = c < kTrunkLayers * kMT ? c : kTrunkLayers * kMT - 1;
          issue_chunk<1, kKS>(stream, nullptr, (offL + c * kKS) * 1024L, ring + ((BS::G_L + t + k) % kNSLOT) * kBSlot, wave, lane);
        }
      }
      stage_phase((t + 2) & 3, kActA0 + (t + 2 < kMT ? lfrag : nfrag) + 2 * ((t + 2) % kMT));
      acc[t & 1] = btile<kKS>(ring + ((BS::G_L + t) % kNSLOT) * kBSlot, 0, cur, lane);
      if constexpr (t > 0) epilogue(std::integral_constant<int, t - 1>{});
    });
    epilogue(std::integral_constant<int, kMT - 1>{});
    store_scales<FMT, kMT>(dpre, l - 1, eb);
#pragma unroll
    for (int i = 0; i < kKS; ++i) cur[i] = nxt[i];
  }
}

}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_mlp_bwd(int feat, int tau, int64_t n_points, const uint16_t* bwd_stream, const uint16_t* acts, const float* albedo,
                                  const float* sigma, const float* sun_v, const float* beta, const float* g_albedo, const float* g_sigma,
                                  const float* g_sun_v, const float* g_beta, uint16_t* dpre, float* d_t, int fmt, void* stream) {
  SR_REQUIRE(feat == kFeat, "sr_satnerf_mlp_bwd: feat=%d unsupported (this build handles %d)", feat, kFeat);
  SR_REQUIRE(tau >= 1 && tau <= 24, "sr_satnerf_mlp_bwd: tau=%d unsupported (1..24)", tau);
  SR_REQUIRE(fmt == SR_FMT16 || fmt == SR_FMT8, "sr_satnerf_mlp_bwd: workspace format must be 16 or 8 (got %d)", fmt);
  SR_REQUIRE(bwd_stream && acts && dpre && albedo && sigma && sun_v && beta, "sr_satnerf_mlp_bwd: null pointer argument");
  if (n_points <= 0) return 0;
  BwdParams p;
  p.g_albedo = g_albedo, p.g_sigma = g_sigma, p.g_sun = g_sun_v, p.g_beta = g_beta;
  p.albedo = albedo, p.sigma = sigma, p.sun_v = sun_v, p.beta = beta;
  p.acts = (const uint4*)acts, p.dpre = (uint4*)dpre, p.d_t = d_t;
  p.stream = (const char*)bwd_stream;
  p.n_points = n_points, p.tau = tau, p.auxs = aux_steps(tau);
  const size_t lds = (size_t)kNSLOT * kBSlot + 8 * 4 * (fmt == SR_FMT8 ? 1 : 2) * 1024;  // weight ring + per-wave phase staging
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)satnerf_bwd_kernel<SR_FMT16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)satnerf_bwd_kernel<SR_FMT8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return 1;
    }
    attr_set = true;
  }
  const long tiles = (n_points + 31) / 32;
  if (fmt == SR_FMT8) hipLaunchKernelGGL(satnerf_bwd_kernel<SR_FMT8>, dim3((unsigned)((tiles + 7) / 8)), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(satnerf_bwd_kernel<SR_FMT16>, dim3((unsigned)((tiles + 7) / 8)), dim3(512), lds, (hipStream_t)stream, p);
  return check_launch("satnerf_bwd_kernel");
}

extern "C" int64_t sr_bwd_stream_elems(int feat, int tau) {
  if (feat != kFeat || tau < 1 || tau > 24) return -1;
  return BwdStream::total_pieces() * 512;
}

extern "C" int64_t sr_dpre_elems_per_tile(int feat, int fmt) {
  if (feat != kFeat || (fmt != SR_FMT16 && fmt != SR_FMT8)) return -1;
  return (int64_t)(fmt == SR_FMT8 ? kD8Units : kDpFrags) * 64 * 8;
}
