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

namespace sr {

constexpr int kFeat = 256;        // trunk width handled by this build (BASELINE fixes 256)
constexpr int kHalf = kFeat / 2;  // head width
constexpr int kKS = kFeat / 16;   // k-steps of a feat-wide input (16)
constexpr int kHS = kHalf / 16;   // k-steps of a head-wide input (8)
constexpr int kTrunkLayers = 7;   // fc_net.2 .. fc_net.14 on MFMA (fc_net.0, K=3, runs on VALU)
constexpr int kMT = kFeat / 32;   // output tiles of a feat-wide layer (8)
constexpr int kMTH = kHalf / 32;  // output tiles of a head-wide layer (4)

constexpr int aux_steps(int tau) { return (8 + ((tau + 7) / 8) * 8 + 15) / 16; }  // tau<=8 -> 1, tau<=24 -> 2

// forward stream: chunk list in consumption order
template <int AUXS>
struct FwdStream {
  static constexpr int SLOTP = 16 + AUXS;  // pieces per ring slot
  static constexpr int N_TRUNK = kTrunkLayers * kMT;
  static constexpr int N_G1 = kMT + 1;       // feats (8 tiles) + sigma tile
  static constexpr int N_G2 = 3 * kMTH;      // rgb1 | sun1 | beta1
  static constexpr int N_S = kMTH;           // sun2, sun3
  static constexpr int H_PIECES = 3 * kHS + AUXS;
  static constexpr int N_H = (H_PIECES + SLOTP - 1) / SLOTP;
  static constexpr int G_G1 = N_TRUNK, G_G2 = G_G1 + N_G1, G_S2 = G_G2 + N_G2, G_S3 = G_S2 + N_S, G_H = G_S3 + N_S;
  static constexpr int NCH = G_H + N_H;
  static constexpr int np(int g) {
    if (g < 0 || g >= NCH) return 0;
    if (g < G_S2) return SLOTP;
    if (g < G_H) return kHS + AUXS;
    int k = g - G_H;
    int left = H_PIECES - k * SLOTP;
    return left > SLOTP ? SLOTP : left;
  }
  static constexpr long offset_pieces(int g) {  // pieces preceding chunk g in the stream
    long n = 0;
    for (int c = 0; c < g; ++c) n += np(c);
    return n;
  }
  static constexpr long total_pieces() {
    long n = 0;
    for (int g = 0; g < NCH; ++g) n += np(g);
    return n;
  }
};

// activations saved per 32-point tile in training mode, as whole B-fragment registers
// (uint4 per (k-step, lane)): aux | a0..a7 (trunk) | feats | rgb_hid sun1 beta_hid | sun2 | sun3
constexpr int act_ksteps(int auxs) { return auxs + 8 * kKS + kKS + 3 * kHS + 2 * kHS; }

}  // namespace sr
