// Kernel parameter blocks shared by the API translation unit and the kernel instantiations (independent of the trunk width:
// they stay outside the per-width inline namespace of mlp_layout.h).
#pragma once
#include "common.h"

namespace sr {

// the fused render pass around the MLP (csrc/mlp_fwd.inc REND): stratified sampling in the prologue, the sky head and the
// sigma->alpha compositing of the workgroup's rays in the epilogue.  rays == NULL: plain MLP launch.
struct RenderParams {
  const float* rays;       // (N, ray_stride >= 11): o(3) d(3) near far sun(3)
  int ray_stride;
  const float* z_in;       // (N,S) given depths (fine pass), or NULL: stratified from u / the in-kernel RNG
  const float* u;          // (N,S) jitter, or NULL: Philox(seed, step_counter[0])
  unsigned long long seed;
  float* step_counter;
  int tick;
  const float* noise;      // (N,S) or NULL
  float noise_std;
  int sky_hidden;
  const float *w1, *b1, *w2, *b2;
  float* z_out;            // (N,S) or NULL
  float* sky;              // (N,3)
  float* weights;          // (N,S)
  float* transp;           // (N,S)
  float* depth;            // (N)
  float* rgb;              // (N,3)
  long n_rays;
  long bank_chunks;        // > 0: rays / ts hold bank_chunks x n_rays rows, this launch renders chunk step_counter[0] % bank_chunks
};

// the training epilogue of the fused render pass (sr_satnerf_render_train): with target != NULL the compositing of a ray is followed, in
// the same wave, by the colour loss and the closed-form compositing backward (ray_device.h render_loss_ray) -- the launch then writes what
// the dX kernel consumes instead of weights / transparency
struct TrainParams {
  const float* target;     // (N,3) ground-truth colours, or NULL: plain render
  const float* sched;      // device-side schedule block ([2] != 0: SNerfLoss warm-up epochs) or NULL
  float beta_min;
  float* loss_parts;       // one partial sum per workgroup
  float* rgb;              // (N,3) rendered colour (logging) or NULL
  float* d_sigma;          // (N,S)
  float* d_albedo;         // (N,S,3)
  float* d_sun;            // (N,S)
  float* g_beta;           // (N,S)
  float* d_sky;            // (N,3)
  // the batch sampler inside the launch (include/satrender.h sr_train_args): rays / ts / target are the resident bank, ray r reads row
  // gather_idx[cursor[0] * n_rays + r]; the compositing wave copies the row out for the later launches of the step
  const long long* gather_idx;
  float* cursor;
  unsigned batches;
  float* out_rays;         // (N,11)
  float* out_rgbs;         // (N,3)
  long long* out_ts;       // (N)
};

struct FwdParams {
  sr_mlp_inputs in;
  RenderParams rend;
  TrainParams train;
  const char* stream_hi;
  const char* stream_lo;
  const float4* l0;
  float* albedo;
  float* sigma;
  float* sun_v;
  float* beta;
  uint4* acts;
  int tau;
};

struct BwdParams {
  const float* g_albedo;  // upstream gradients of the per-point outputs (any may be null = 0)
  const float* g_sigma;
  const float* g_sun;
  const float* g_beta;
  const float* albedo;    // forward outputs (activation derivatives of the heads)
  const float* sigma;
  const float* sun_v;
  const float* beta;
  const uint4* acts;      // saved activations, act_ksteps(auxs) fragments per 32-point tile
  uint4* dpre;            // out: pre-activation gradients, kDpFrags fragments per tile
  uint4* emax;            // out (SR_FMT8): the table of exponent maxima behind the workspace's last tile (mlp_layout.h), one entry per 4 tiles
  float* d_t;             // out: (P, tau) gradient of the embedding vector per point (may be null)
  const char* stream;     // transposed weight stream (bf16)
  long n_points;
  int tau;
  int auxs;
};

}  // namespace sr
