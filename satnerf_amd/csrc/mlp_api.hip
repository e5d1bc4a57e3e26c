// C ABI of the fused forward MLP: argument checks + dispatch to the per-mode translation units.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_params.h"

namespace sr {
int launch_fwd_p1a1(const FwdParams&, int, hipStream_t);
int launch_fwd_p1a2(const FwdParams&, int, hipStream_t);
int launch_fwd_p3a1(const FwdParams&, int, hipStream_t);
int launch_fwd_p3a2(const FwdParams&, int, hipStream_t);
int launch_fwd_h1a1(const FwdParams&, int, hipStream_t);  // fp16 operands (mlp_fwd_h1a*.hip)
int launch_fwd_h1a2(const FwdParams&, int, hipStream_t);
int launch_fwd512_p1a1(const FwdParams&, int, hipStream_t);  // the 512-wide build (mlp_fwd512_*.hip): inference, bf16
int launch_fwd512_p1a2(const FwdParams&, int, hipStream_t);
int launch_fwd512_h1a1(const FwdParams&, int, hipStream_t);  // ... with fp16 operands
int launch_fwd512_h1a2(const FwdParams&, int, hipStream_t);
long fwd512_stream_pieces_a1();
long fwd512_stream_pieces_a2();
}  // namespace sr
using namespace sr;
extern "C" int sr_satnerf_mlp_fwd(const sr_mlp_inputs* in, int feat, int tau, int mode, const uint16_t* stream_hi,
                                  const uint16_t* stream_lo, const float* l0, float* albedo, float* sigma, float* sun_v,
                                  float* beta, uint16_t* acts, int act_fmt, void* stream) {
  SR_REQUIRE(in != nullptr, "sr_satnerf_mlp_fwd: null inputs");
  SR_REQUIRE(feat == kFeat || feat == 512, "sr_satnerf_mlp_fwd: feat=%d unsupported (this build handles %d and 512)", feat, kFeat);
  SR_REQUIRE(mode == SR_MODE_BF16 || mode == SR_MODE_BF16X3 || mode == SR_MODE_F16, "sr_satnerf_mlp_fwd: bad mode %d", mode);
  SR_REQUIRE(feat == kFeat || ((mode == SR_MODE_BF16 || mode == SR_MODE_F16) && (acts == nullptr || act_fmt == SR_FMT8)),
             "sr_satnerf_mlp_fwd: feat=512 runs the fused kernel in SR_MODE_BF16 / SR_MODE_F16, saving activations in SR_FMT8 only (parity mode: layer by layer)");
  SR_REQUIRE(tau >= 1 && tau <= 24, "sr_satnerf_mlp_fwd: tau=%d unsupported (1..24)", tau);
  SR_REQUIRE(stream_hi && l0 && in->org && in->sun && in->temb, "sr_satnerf_mlp_fwd: null pointer argument");
  SR_REQUIRE(mode != SR_MODE_BF16X3 || stream_lo, "sr_satnerf_mlp_fwd: BF16X3 needs the lo plane");
  SR_REQUIRE(in->n_samples >= 1, "sr_satnerf_mlp_fwd: n_samples must be >= 1");
  SR_REQUIRE(acts == nullptr || act_fmt == SR_FMT16 || act_fmt == SR_FMT8, "sr_satnerf_mlp_fwd: act_fmt must be 16 or 8 (got %d)", act_fmt);
  if (in->n_points <= 0) return 0;
  FwdParams p;
  p.in = *in;
  p.rend = RenderParams{};
  p.train = TrainParams{};
  p.stream_hi = (const char*)stream_hi;
  p.stream_lo = (const char*)stream_lo;
  p.l0 = (const float4*)l0;
  p.albedo = albedo, p.sigma = sigma, p.sun_v = sun_v, p.beta = beta;
  p.acts = (uint4*)acts;
  p.tau = tau;
  hipStream_t st = (hipStream_t)stream;
  const int save = acts != nullptr ? act_fmt : 0;
  const int auxs = aux_steps(tau);
  if (feat == 512 && mode == SR_MODE_F16) return auxs == 1 ? launch_fwd512_h1a1(p, save, st) : launch_fwd512_h1a2(p, save, st);
  if (feat == 512) return auxs == 1 ? launch_fwd512_p1a1(p, save, st) : launch_fwd512_p1a2(p, save, st);
  if (mode == SR_MODE_BF16) return auxs == 1 ? launch_fwd_p1a1(p, save, st) : launch_fwd_p1a2(p, save, st);
  if (mode == SR_MODE_F16) return auxs == 1 ? launch_fwd_h1a1(p, save, st) : launch_fwd_h1a2(p, save, st);
  return auxs == 1 ? launch_fwd_p3a1(p, save, st) : launch_fwd_p3a2(p, save, st);
}

extern "C" int64_t sr_fwd_stream_elems(int feat, int tau) {
  if ((feat != kFeat && feat != 512) || tau < 1 || tau > 24) return -1;
  if (feat == 512) return (aux_steps(tau) == 1 ? fwd512_stream_pieces_a1() : fwd512_stream_pieces_a2()) * 512;
  return (aux_steps(tau) == 1 ? FwdStream<1>::total_pieces() : FwdStream<2>::total_pieces()) * 512;
}

extern "C" int sr_render_points_per_block(int feat, int mode) {
  if (mode != SR_MODE_BF16 && mode != SR_MODE_BF16X3 && mode != SR_MODE_F16) return -1;
  if (feat == kFeat) return mode == SR_MODE_BF16X3 ? 128 : 256;  // Mode<NPASS>::NW * 32
  if (feat == 512 && (mode == SR_MODE_BF16 || mode == SR_MODE_F16)) return 128;
  return -1;
}

// shared by sr_satnerf_render_fwd and sr_satnerf_render_train: argument checks of the render pass, parameter block, dispatch
static int render_launch(const char* who, const sr_render_args* in, int feat, int tau, int mode, const uint16_t* stream_hi, const uint16_t* stream_lo,
                         const float* l0, const sr_render_outputs* out, const sr_train_args* tr, uint16_t* acts, int act_fmt, void* stream) {
  SR_REQUIRE(in != nullptr && out != nullptr, "%s: null argument block", who);
  const int per_block = sr_render_points_per_block(feat, mode);
  SR_REQUIRE(per_block > 0, "%s: no fused kernel for feat=%d mode=%d (256: every mode; 512: SR_MODE_BF16 / SR_MODE_F16)", who, feat, mode);
  SR_REQUIRE(tau >= 1 && tau <= 24, "%s: tau=%d unsupported (1..24)", who, tau);
  SR_REQUIRE(in->n_samples >= 2 && per_block % in->n_samples == 0, "%s: n_samples=%d must be >= 2 and divide %d (sr_render_points_per_block)", who,
             in->n_samples, per_block);
  SR_REQUIRE(in->rays && in->ts && in->temb && in->ray_stride >= 11, "%s: rays (stride >= 11), ts and temb are required", who);
  SR_REQUIRE(stream_hi && l0 && (mode != SR_MODE_BF16X3 || stream_lo), "%s: null weight stream", who);
  SR_REQUIRE(in->sky_w1 && in->sky_b1 && in->sky_w2 && in->sky_b2 && in->sky_hidden >= 1, "%s: the sky head's weights are required", who);
  SR_REQUIRE(tr != nullptr || (out->weights && out->transparency), "%s: weights and transparency outputs are required", who);
  SR_REQUIRE(!in->tick || in->step_counter, "%s: tick needs the 4-float step counter block", who);
  SR_REQUIRE(in->bank_chunks >= 0 && (in->bank_chunks == 0 || in->step_counter), "%s: bank_chunks needs the step counter", who);
  if (tr != nullptr) {
    SR_REQUIRE(mode == SR_MODE_BF16 || mode == SR_MODE_F16, "%s: the fused training forward exists for SR_MODE_BF16 / SR_MODE_F16 (8-bit saved state)", who);
    SR_REQUIRE(acts != nullptr && act_fmt == SR_FMT8, "%s: needs the SR_FMT8 activation workspace", who);
    SR_REQUIRE(in->n_samples <= 64, "%s: n_samples=%d unsupported (one wave per ray: <= 64); use the separate launches", who, in->n_samples);
    SR_REQUIRE(tr->target && tr->loss_parts && tr->d_sigma && tr->d_albedo && tr->d_sun_v && tr->g_beta && tr->d_sky && out->sky, "%s: null training output", who);
    SR_REQUIRE(out->albedo && out->sigma && out->sun_v && out->beta, "%s: the four per-point outputs are required (the dX pass reads them)", who);
    if (tr->gather_idx != nullptr) {
      SR_REQUIRE(tr->cursor && tr->batches >= 1 && tr->batches < (1 << 24), "%s: the in-launch sampler needs a cursor block and 1 <= batches < 2^24", who);
      SR_REQUIRE(tr->out_rays && tr->out_rgbs && tr->out_ts, "%s: the in-launch sampler needs out_rays / out_rgbs / out_ts", who);
      SR_REQUIRE(in->ray_stride == 11 && in->bank_chunks == 0 && in->z_in == nullptr && in->u == nullptr && in->noise == nullptr,
                 "%s: the in-launch sampler takes a bank of 11-float rows and draws its own jitter (no z_in / u / noise / bank_chunks)", who);
    }
  }
  SR_REQUIRE(in->tick != 2 || (tr != nullptr && in->bank_chunks == 0), "%s: tick == 2 (tick first) is a training-launch option", who);
  SR_REQUIRE(in->tick >= 0 && in->tick <= 2, "%s: tick must be 0, 1 or 2", who);
  if (in->n_rays <= 0) return 0;
  FwdParams p;
  p.in.org = in->rays, p.in.org_stride = in->ray_stride;
  p.in.dir = in->rays + 3, p.in.dir_stride = in->ray_stride;
  p.in.sun = in->rays + 8, p.in.sun_stride = in->ray_stride;
  p.in.z = nullptr, p.in.temb = in->temb, p.in.ts = in->ts;
  p.in.n_points = in->n_rays * in->n_samples, p.in.n_samples = in->n_samples;
  RenderParams& r = p.rend;
  r.rays = in->rays, r.ray_stride = in->ray_stride, r.z_in = in->z_in, r.u = in->u, r.seed = in->seed, r.step_counter = in->step_counter;
  r.tick = in->tick, r.noise = in->noise, r.noise_std = in->noise_std, r.sky_hidden = in->sky_hidden;
  r.w1 = in->sky_w1, r.b1 = in->sky_b1, r.w2 = in->sky_w2, r.b2 = in->sky_b2;
  r.z_out = out->z_vals, r.sky = out->sky, r.weights = out->weights, r.transp = out->transparency, r.depth = out->depth, r.rgb = out->rgb;
  r.n_rays = in->n_rays, r.bank_chunks = in->bank_chunks;
  p.train = TrainParams{};
  if (tr != nullptr) {
    TrainParams& t = p.train;
    t.target = tr->target, t.sched = tr->sched, t.beta_min = tr->beta_min, t.loss_parts = tr->loss_parts, t.rgb = tr->rgb;
    t.d_sigma = tr->d_sigma, t.d_albedo = tr->d_albedo, t.d_sun = tr->d_sun_v, t.g_beta = tr->g_beta, t.d_sky = tr->d_sky;
    t.gather_idx = (const long long*)tr->gather_idx, t.cursor = tr->cursor, t.batches = (unsigned)tr->batches;
    t.out_rays = tr->out_rays, t.out_rgbs = tr->out_rgbs, t.out_ts = (long long*)tr->out_ts;
  }
  p.stream_hi = (const char*)stream_hi, p.stream_lo = (const char*)stream_lo, p.l0 = (const float4*)l0;
  p.albedo = out->albedo, p.sigma = out->sigma, p.sun_v = out->sun_v, p.beta = out->beta;
  p.acts = (uint4*)acts, p.tau = tau;
  hipStream_t st = (hipStream_t)stream;
  const int save = acts != nullptr ? act_fmt : 0;
  const int auxs = aux_steps(tau);
  if (feat == 512 && mode == SR_MODE_F16) return auxs == 1 ? launch_fwd512_h1a1(p, save, st) : launch_fwd512_h1a2(p, save, st);
  if (feat == 512) return auxs == 1 ? launch_fwd512_p1a1(p, save, st) : launch_fwd512_p1a2(p, save, st);
  if (mode == SR_MODE_BF16) return auxs == 1 ? launch_fwd_p1a1(p, save, st) : launch_fwd_p1a2(p, save, st);
  if (mode == SR_MODE_F16) return auxs == 1 ? launch_fwd_h1a1(p, save, st) : launch_fwd_h1a2(p, save, st);
  return auxs == 1 ? launch_fwd_p3a1(p, save, st) : launch_fwd_p3a2(p, save, st);
}

extern "C" int sr_satnerf_render_fwd(const sr_render_args* in, int feat, int tau, int mode, const uint16_t* stream_hi, const uint16_t* stream_lo,
                                     const float* l0, const sr_render_outputs* out, void* stream) {
  return render_launch("sr_satnerf_render_fwd", in, feat, tau, mode, stream_hi, stream_lo, l0, out, nullptr, nullptr, 0, stream);
}

extern "C" int sr_satnerf_render_train(const sr_render_args* in, int feat, int tau, int mode, const uint16_t* stream_hi, const uint16_t* stream_lo,
                                       const float* l0, const sr_render_outputs* out, const sr_train_args* train, uint16_t* acts, int act_fmt,
                                       void* stream) {
  SR_REQUIRE(train != nullptr, "sr_satnerf_render_train: null training argument block");
  return render_launch("sr_satnerf_render_train", in, feat, tau, mode, stream_hi, stream_lo, l0, out, train, acts, act_fmt, stream);
}
