// C ABI of the fused Sat-NeRF MLP backward (dX chain): argument checks + dispatch to the per-width builds of mlp_bwd.inc
// (this translation unit holds the 256-wide one; mlp_bwd512.hip the 512-wide one, 8-bit workspaces only).
#ifndef SR_BWD_TILES
#define SR_BWD_TILES 1  // 32-point tiles per wave (mlp_bwd.inc): 8 waves of one tile.  2 = 4 waves of two tiles (one A-fragment read per two
#endif                  // MFMAs), correct and measured 30 % slower as compiler-scheduled code (profiles/r04_ab_variants.txt)
#if SR_BWD_TILES == 2
#define SR_MODE1_NW 4
#endif
#include "mlp_bwd.inc"

namespace sr {
int launch_bwd512(const BwdParams& p, int fmt, hipStream_t st);
long bwd512_stream_pieces();
int dpre8_units_512();
int act8_units_512(int auxs);
}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_mlp_bwd(int feat, int tau, int64_t n_points, const uint16_t* bwd_stream, const uint16_t* acts, const float* albedo,
                                  const float* sigma, const float* sun_v, const float* beta, const float* g_albedo, const float* g_sigma,
                                  const float* g_sun_v, const float* g_beta, uint16_t* dpre, float* d_t, int fmt, void* stream) {
  SR_REQUIRE(feat == kFeat || feat == 512, "sr_satnerf_mlp_bwd: feat=%d unsupported (this build handles %d and 512)", feat, kFeat);
  SR_REQUIRE(tau >= 1 && tau <= 24, "sr_satnerf_mlp_bwd: tau=%d unsupported (1..24)", tau);
  SR_REQUIRE(fmt == SR_FMT16 || fmt == SR_FMT8, "sr_satnerf_mlp_bwd: workspace format must be 16 or 8 (got %d)", fmt);
  SR_REQUIRE(feat == kFeat || fmt == SR_FMT8, "sr_satnerf_mlp_bwd: feat=512 trains on the 8-bit workspaces only");
  SR_REQUIRE(bwd_stream && acts && dpre && albedo && sigma && sun_v && beta, "sr_satnerf_mlp_bwd: null pointer argument");
  if (n_points <= 0) return 0;
  BwdParams p;
  p.g_albedo = g_albedo, p.g_sigma = g_sigma, p.g_sun = g_sun_v, p.g_beta = g_beta;
  p.albedo = albedo, p.sigma = sigma, p.sun_v = sun_v, p.beta = beta;
  p.acts = (const uint4*)acts, p.dpre = (uint4*)dpre, p.d_t = d_t;
  p.stream = (const char*)bwd_stream;
  p.n_points = n_points, p.tau = tau, p.auxs = aux_steps(tau);
  // SR_FMT8: the table of exponent maxima sits behind the workspace's last tile (sr_dpre_workspace_elems)
  p.emax = fmt == SR_FMT8 ? (uint4*)(dpre + sr_workspace_tiles(n_points) * sr_dpre_elems_per_tile(feat, fmt)) : nullptr;
  if (feat == 512) return launch_bwd512(p, fmt, (hipStream_t)stream);
  return fmt == SR_FMT8 ? launch_bwd_fmt<SR_FMT8>(p, (hipStream_t)stream) : launch_bwd_fmt<SR_FMT16>(p, (hipStream_t)stream);
}

extern "C" int64_t sr_bwd_stream_elems(int feat, int tau) {
  if ((feat != kFeat && feat != 512) || tau < 1 || tau > 24) return -1;
  return (feat == 512 ? bwd512_stream_pieces() : BwdStream::total_pieces()) * 512;
}

extern "C" int64_t sr_workspace_tiles(int64_t n_points) { return n_points < 0 ? -1 : sr::ws_tiles(n_points); }

extern "C" int64_t sr_dpre_elems_per_tile(int feat, int fmt) {
  if (fmt != SR_FMT16 && fmt != SR_FMT8) return -1;
  if (feat == 512) return fmt == SR_FMT8 ? (int64_t)dpre8_units_512() * 64 * 8 : -1;
  if (feat != kFeat) return -1;
  return (int64_t)(fmt == SR_FMT8 ? kD8Units : kDpFrags) * 64 * 8;
}

extern "C" int64_t sr_dpre_workspace_elems(int64_t n_points, int feat, int fmt) {
  const int64_t per_tile = sr_dpre_elems_per_tile(feat, fmt);
  if (per_tile < 0 || n_points < 0) return -1;
  return sr::ws_tiles(n_points) * per_tile + (fmt == SR_FMT8 ? sr::ws_emax_bytes(n_points) / 2 : 0);
}

extern "C" int64_t sr_act_elems_per_tile(int feat, int fmt) {
  if (fmt != SR_FMT16 && fmt != SR_FMT8) return -1;
  if (feat == 512) return fmt == SR_FMT8 ? (int64_t)act8_units_512(2) * 64 * 8 : -1;
  if (feat != kFeat) return -1;
  return (int64_t)(fmt == SR_FMT8 ? act8_units(2) : act_ksteps(2)) * 64 * 8;  // sized for the larger aux layout
}
