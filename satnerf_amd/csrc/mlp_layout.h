// Geometry of the fused Sat-NeRF MLP kernels: the weight "stream", the slot permutation and the
// saved-activation layout.  Shared by the forward kernel, the backward kernels and (through the
// sr_*_stream_elems entry points) the host packer satnerf_amd/packing.py, which mirrors it in numpy.
//
// Orientation.  Every dense layer is computed "swapped":  C[feature][point] = sum_k W[feature][k] *
// act[k][point], i.e. the WEIGHTS are the MFMA A operand (rows = output features, 32 per tile) and the
// ACTIVATIONS are the B operand (columns = the wave's 32 points).  With v_mfma_f32_32x32x16_bf16 the
// lane (p = lane&31, h = lane>>5) then receives, for its point p, output rows (g&3)+8*(g>>2)+4*h in
// accumulator register g -- which after the activation and a bf16 pack is already a valid B fragment
// of the next layer, provided the next layer's weight columns are permuted to match.  That
// permutation ("slot" order) is baked into the stream by the packer; no cross-lane traffic is needed
// and activations never leave registers between layers.
//
//   slot sigma = 16*s + 8*h + j   (s = k-step, h = lane half, j = element of the lane's 8-bf16 fragment)
//   holds feature  phi(sigma) = 32*(s>>1) + (g&3) + 8*(g>>2) + 4*h   with g = j + 8*(s&1).
//
// Stream.  A "piece" is the A fragment of one k-step of one 32-row output tile: 64 lanes x 16 B = 1 KiB
// (x2 in BF16X3 mode: hi plane then lo plane), unit L = h*32 + r holding row r, slots 16*s+8*h+[0,8).
// A "chunk" is up to SLOTP = 16+AUXS consecutive pieces of one tile; the kernel walks the chunks
// strictly in order through an LDS ring.  Every tile ends with AUXS "aux" k-steps whose B operand is the
// per-point constant vector [sun(3), 1, xyz(3), 0, t(tau)...] -- this is how biases, the skip
// connection's xyz columns, the sun-direction columns and the embedding columns enter the MFMA.
#pragma once

// The trunk width is a compile-time constant of a translation unit: SR_FEAT = 256 (BASELINE's width; every kernel) or 512
// (opt.py:50's default, what run_all.sh trains sat-nerf with: the forward kernel only, mlp_fwd512_*.hip).  Everything that depends
// on it lives in an inline namespace named after the width, so the two builds of the same templates never share a symbol.
#ifndef SR_FEAT
#define SR_FEAT 256
#endif
#define SR_CAT_(a, b) a##b
#define SR_CAT(a, b) SR_CAT_(a, b)
#ifdef SR_F16  // the fp16-operand build of the forward kernel (SR_MODE_F16) gets its own namespace too
#define SR_FEAT_NS SR_CAT(SR_CAT(f, SR_FEAT), h)
#else
#define SR_FEAT_NS SR_CAT(f, SR_FEAT)
#endif

namespace sr {
inline namespace SR_FEAT_NS {

constexpr int kFeat = SR_FEAT;    // trunk width of this translation unit
constexpr int kHalf = kFeat / 2;  // head width
constexpr int kKS = kFeat / 16;   // k-steps of a feat-wide input (16)
constexpr int kHS = kHalf / 16;   // k-steps of a head-wide input (8)
constexpr int kTrunkLayers = 7;   // fc_net.2 .. fc_net.14 on MFMA (fc_net.0, K=3, runs on VALU)
constexpr int kMT = kFeat / 32;   // output tiles of a feat-wide layer (8)
constexpr int kMTH = kHalf / 32;  // output tiles of a head-wide layer (4)

constexpr int aux_steps(int tau) { return (8 + ((tau + 7) / 8) * 8 + 15) / 16; }  // tau<=8 -> 1, tau<=24 -> 2

// forward stream: chunk list in consumption order.  After the trunk each 128-wide hidden vector is folded into the
// 5-row output head right after it is produced, so at most one hidden vector is live beside `feats`:
//   G1 (feats, sigma) | G2r (rgb hidden) | Hr | G2s (sun hidden 1) | S2 | S3 | Hs | G2b (beta hidden) | Hb (+aux: biases)
template <int AUXS>
struct FwdStream {
  static constexpr int SLOTP = kKS + AUXS;  // pieces per ring slot
  static constexpr int NSTAGE = 10;
  static constexpr int N_TRUNK = kTrunkLayers * kMT;
  static constexpr int cnt(int st) {
    constexpr int c[NSTAGE] = {kTrunkLayers * kMT, kMT + 1, kMTH, 1, kMTH, kMTH, kMTH, 1, kMTH, 1};
    return c[st];
  }
  static constexpr int size(int st) {
    constexpr int z[NSTAGE] = {kKS + AUXS, kKS + AUXS, kKS + AUXS, kHS, kKS + AUXS, kHS + AUXS, kHS + AUXS, kHS, kKS + AUXS, kHS + AUXS};
    return z[st];
  }
  static constexpr int first(int st) {  // global index of the stage's first chunk
    int g = 0;
    for (int k = 0; k < st; ++k) g += cnt(k);
    return g;
  }
  static constexpr int G_G1 = N_TRUNK, G_G2R = G_G1 + kMT + 1, G_HR = G_G2R + kMTH, G_G2S = G_HR + 1, G_S2 = G_G2S + kMTH,
                       G_S3 = G_S2 + kMTH, G_HS = G_S3 + kMTH, G_G2B = G_HS + 1, G_HB = G_G2B + kMTH;
  static constexpr int NCH = G_HB + 1;
  static constexpr int np(int g) {
    if (g < 0 || g >= NCH) return 0;
    int st = 0;
    while (g >= first(st) + cnt(st)) ++st;
    return size(st);
  }
  static constexpr long offset_pieces(int g) {  // pieces preceding chunk g in the stream
    long n = 0;
    for (int c = 0; c < g; ++c) n += np(c);
    return n;
  }
  static constexpr long total_pieces() { return offset_pieces(NCH); }
};

// activations saved per 32-point tile in training mode, as whole B-fragment registers (uint4 per (fragment, lane)):
//   aux | a0..a7 (trunk) | feats | rgb_hid sun1 beta_hid | sun2 | sun3
// sin stages are saved as the unorm16 PHASE of their pre-activation, feats and aux as bf16 values.
constexpr int act_ksteps(int auxs) { return auxs + 8 * kKS + kKS + 3 * kHS + 2 * kHS; }
constexpr int kActA0 = 0;                       // + auxs: fragment index of a0 (a_l at kActA0 + 16 l)
constexpr int kActFeats = 8 * kKS;              // + auxs
constexpr int kActRgbh = kActFeats + kKS, kActS1 = kActRgbh + kHS, kActE1 = kActS1 + kHS, kActS2 = kActE1 + kHS, kActS3 = kActS2 + kHS;

// pre-activation gradients written by the dX kernel per 32-point tile (bf16 B fragments, forward row-slot order):
//   d_pre_0..d_pre_7 | d_feats | d_sigma_pre | d_rgbh d_s1 d_e1 | d_s2 | d_s3 | d_head
constexpr int kDpL = 0, kDpFeats = 8 * kKS, kDpSigma = kDpFeats + kKS, kDpRgbh = kDpSigma + 1, kDpS1 = kDpRgbh + kHS,
              kDpE1 = kDpS1 + kHS, kDpS2 = kDpE1 + kHS, kDpS3 = kDpS2 + kHS, kDpHead = kDpS3 + kHS, kDpFrags = kDpHead + 1;

// ---- 8-bit training workspaces (SR_FMT8: the throughput mode's saved state; SR_FMT16 = the layouts above) -----------------
// The fused training step is bound by the bytes the forward / dX / weight-gradient kernels exchange through HBM
// (2.08 GB per 1024x64 step at 16 bits), so the throughput mode stores one byte per value.  Unit = 1 KiB =
// 64 lanes x 16 B; a "double fragment" (DF) holds, for lane (p, h), the 16 values of one 32-row output tile = B fragments
// 2t and 2t+1 of the 16-bit layout (logical fragment f lives in half f & 1 of DF f >> 1).  Two codecs:
//   PHASE8: sin stages, u = round(frac(pre-activation in revolutions) * 256) mod 256 (sin and cos recovered to +-pi/256 rad);
//   MX8   : identity stages and pre-activation gradients, offset-binary int8 with ONE shared power-of-two scale per lane per DF
//           (16 values): v = (u - 128) * 2^(E - 133), E = biased exponent of 1.0079 * max|v| -- the micro-scaled int8 of the
//           OCP MX formats with the block laid along a point's features (what a lane holds) instead of along K.
// activations (per 32-point tile):  aux (auxs bf16 fragments) | a0..a7 (kMT DF each, PHASE8) | feats (kMT DF, MX8) |
//   rgbh s1 e1 s2 s3 (kMTH DF each, PHASE8) | feats scale unit (lane's 16 B: byte t = E of feats DF t).  The DF of logical
//   fragment f is unit f >> 1 (+ auxs): feat 256: 64 + 8 + 20 = 92 DF, feat 512: 184.
constexpr int kA8Scale = (9 * kKS + 5 * kHS) / 2;  // + auxs
constexpr int act8_units(int auxs) { return auxs + kA8Scale + 1; }
// pre-activation gradients:  d_pre_0..7 (kMT DF each) | d_feats (kMT) | d_rgbh d_s1 d_e1 d_s2 d_s3 (kMTH each) -- all MX8 -- |
//   d_sigma_pre, d_head (one bf16 fragment each) | scale units.  Scale group g = 0..7 trunk layer, 8 feats, 9 rgbh, 10 s1, 11 e1,
//   12 s2, 13 s3 owns a slot of kMT bytes (one per tile of the group); kD8GroupsPerUnit = 16 / kMT slots share a lane's 16 B:
//   byte (g % kD8GroupsPerUnit) * kMT + t of unit kD8Scale + g / kD8GroupsPerUnit.  feat 256: 7 units (101 in all), feat 512: 14 (200).
constexpr int kD8Sigma = (9 * kKS + 5 * kHS) / 2, kD8Head = kD8Sigma + 1, kD8Scale = kD8Head + 1;
constexpr int kD8GroupsPerUnit = 16 / kMT, kD8ScaleUnits = (14 + kD8GroupsPerUnit - 1) / kD8GroupsPerUnit, kD8Units = kD8Scale + kD8ScaleUnits;
static_assert(kMT <= 16, "a scale slot must fit the lane's 16 bytes");
constexpr int dp8_unit(int frag) { return frag < kDpSigma ? frag >> 1 : (frag - 1) >> 1; }  // d_sigma_pre sits between the logical fragments
constexpr int dp8_group(int frag) { return frag < kDpFeats ? frag / kKS : frag < kDpSigma ? 8 : 9 + (frag - kDpRgbh) / kHS; }
// SR_FMT16 / SR_FMT8 are defined in include/satrender.h
//
// Exponent maxima (SR_FMT8).  The 4-wave weight-gradient kernel (wgrad9.hip) contracts fp16 operands and has to know, before it decodes
// its first value, the largest exponent its slice of points will show.  The dX kernel holds every MX8 exponent byte in a register when
// it stores it, so it leaves the maxima behind: a table of 16-byte entries behind the last tile of the dpre workspace, one entry per
// kEmaxTiles consecutive tiles, byte g = the largest exponent byte of scale group g (0..13, above) over those tiles' 64 lanes, byte
// kEmaxFeats = the same for the feats exponents the FORWARD saved (the MX8 column operand), byte kEmaxRaw = the largest biased exponent
// of the two bf16 row fragments d_sigma_pre / d_head.  (r04 scanned the exponent bytes in the weight-gradient kernel itself: 28 MB of
// reads and a 9-us serial prologue per launch.)
constexpr int kEmaxTiles = 4, kEmaxFeats = 14, kEmaxRaw = 15;

// backward (dX) stream: transposed weights, scale 1, chunk list in consumption order.  A chunk is `tiles(st)` output
// tiles of `ppt(st)` pieces each (pieces of a tile are contiguous); every chunk fits one ring slot of SLOTP pieces (24 at feat
// 256, 33 at feat 512, where a bG2 tile -- 48 pieces -- is consumed as kG2Split = 2 chunks of 24 accumulating into one tile).
//   bH (d_head -> d rgbh | d s3 | d e1) | bS3 | bS2 | bG2 ([d rgbh, d s1, d e1] -> d feats) | bDT (d e1 -> d t) |
//   bG1 ([d feats, d sigma] -> d a7) | bL7 .. bL1 (d pre_l -> d a_{l-1})
constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
struct BwdStream {
  static constexpr int kG2Split = (3 * kHS + 23) / 24;  // chunks per bG2 tile
  static constexpr int kG2Part = 3 * kHS / kG2Split;    // pieces per bG2 chunk
  static constexpr int SLOTP = max3(kG2Part, kKS + 1, 3 * kMTH);  // pieces per ring slot
  static constexpr int NSTAGE = 7;
  static constexpr int cnt(int st) {
    constexpr int c[NSTAGE] = {1, kMTH / 2, kMTH / 2, kMT * kG2Split, 1, kMT, kTrunkLayers * kMT};
    return c[st];
  }
  static constexpr int tiles(int st) {
    constexpr int t[NSTAGE] = {3 * kMTH, 2, 2, 1, 1, 1, 1};
    return t[st];
  }
  static constexpr int ppt(int st) {
    constexpr int z[NSTAGE] = {1, kHS, kHS, kG2Part, kHS, kKS + 1, kKS};
    return z[st];
  }
  static constexpr int first(int st) {
    int g = 0;
    for (int k = 0; k < st; ++k) g += cnt(k);
    return g;
  }
  static constexpr int G_H = 0, G_S3 = 1, G_S2 = G_S3 + kMTH / 2, G_G2 = G_S2 + kMTH / 2, G_DT = G_G2 + kMT * kG2Split, G_G1 = G_DT + 1,
                       G_L = G_G1 + kMT;
  static constexpr int NCH = G_L + kTrunkLayers * kMT;
  static constexpr int np(int g) {
    if (g < 0 || g >= NCH) return 0;
    int st = 0;
    while (g >= first(st) + cnt(st)) ++st;
    return tiles(st) * ppt(st);
  }
  static constexpr long offset_pieces(int g) {
    long n = 0;
    for (int c = 0; c < g; ++c) n += np(c);
    return n;
  }
  static constexpr long total_pieces() { return offset_pieces(NCH); }
};

// ---- weight-gradient partial blocks (wgrad.hip) -------------------------------------------------------------------
// One split-K slice of one job block: 256 x 256 main block + 256 x 32 aux columns, fp32.  Job table row (kWgTableInts int32):
//   rf0 nr0 rf1 nr1 | cf0 nc0 cf1 nc1 | col_kind n_slices first_slice span
// = up to two ranges of dpre row fragments (nr0 + nr1 <= 16), up to two ranges of activation column fragments (nc0 + nc1 <= 16);
// slices of all blocks are numbered consecutively (sr_wgrad_plan) and slice s lives at partial + s * kWgBlockFloats.
constexpr int kWgBlockFloats = 256 * 256 + 256 * 32;
constexpr int kWgTableInts = 12, kWgSlices = 9, kWgFirstSlice = 10;
constexpr int kWgSpan = 11;  // (first row only) stream-K plans: tile units per workgroup of the 4-wave kernel; 0 = one slice per workgroup

#ifdef __HIPCC__
// sum over the slices of element k = block * kWgBlockFloats + offset of a block with `ns` slices, the first at slice `first`.
// All of a chunk's loads are issued before the first add (slices beyond ns are clamped to the last one and masked): with a plain loop
// every 295-KiB-strided load waited for the previous add and the reduction ran at 2.6 TB/s (23.6 us, PMC r02a); ten in flight: 20.0 us;
// r06: a whole block's slices (18-19 at width 256) in flight at once, the callers fetch everything that does not depend on the sum
// (optimizer state, the block table) BEFORE it -- the tail launch was a chain of five dependent memory round trips per thread.
#ifndef SR_TAIL_INFLIGHT
#define SR_TAIL_INFLIGHT 20
#endif
__device__ __forceinline__ float wg_sum_slices(const float* __restrict__ partial, int ns, int first, int w) {
  const float* p = partial + (long)first * kWgBlockFloats + w;
  constexpr int NF = SR_TAIL_INFLIGHT;
  float sum = 0.f;
  for (int sp = 0; sp < ns; sp += NF) {
    float v[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int sl = sp + i;
      v[i] = p[(long)(sl < ns ? sl : ns - 1) * kWgBlockFloats];
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) sum += sp + i < ns ? v[i] : 0.f;
  }
  return sum;
}
__device__ __forceinline__ float wg_sum_slices(const float* __restrict__ partial, const int* __restrict__ blocks, int k) {
  const int b = k / kWgBlockFloats;
  return wg_sum_slices(partial, blocks[kWgTableInts * b + kWgSlices], blocks[kWgTableInts * b + kWgFirstSlice], k - b * kWgBlockFloats);
}
#endif

}  // inline namespace SR_FEAT_NS
}  // namespace sr
