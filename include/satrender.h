/*
 * libsatrender -- C ABI of the MI355X-native Sat-NeRF volumetric-rendering hot path.
 *
 * The reference (centreborelli/satnerf) is pure Python/PyTorch and has no FFI; each entry point below
 * replaces the ATen op sequence of the reference lines it cites (paths relative to the reference
 * repository root).  The host side (the satnerf_amd Python package) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch); the library
 *     borrows it for the duration of the enqueue and allocates nothing;
 *   - all tensors fp32, contiguous, row-major unless stated; `ts` is int64;
 *   - `stream` is a hipStream_t (0 = default stream); calls only enqueue, they never synchronise;
 *   - return value 0 = ok, non-zero = error, text via sr_last_error() (thread-local);
 *   - gfx950 (MI355X) only.
 */
#ifndef SATRENDER_H
#define SATRENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_VERSION 100 /* major*10000 + minor*100 + patch */

/* numeric modes of the fused MLP (SURVEY.md section 7 "Hard parts") */
#define SR_MODE_BF16 1   /* single-pass bf16 MFMA, fp32 accumulate -- throughput mode            */
#define SR_MODE_BF16X3 3 /* hi*hi + lo*hi + hi*lo split on the same MFMA pipe -- parity mode (~2e-6) */
#define SR_MODE_F16 2    /* single-pass fp16 MFMA, fp32 accumulate: the throughput of SR_MODE_BF16 with 11-bit operands (~1.5e-4);
                            forward only -- weights and activations are O(1) here; the backward kernels stay bf16 -- widths 256 and 512,
                            stream_hi then holds fp16 (sr_pack_stream / sr_pack_all `n_f16`); the training workspaces keep their (bf16 / 8-bit) formats */

/* formats of the training workspaces the forward / dX / weight-gradient kernels exchange through HBM:
 * 16 = unorm16 phase / bf16 (the parity mode's backward), 8 = one byte per value (PHASE8 / micro-scaled int8): half the
 * bytes of a training step, used by the throughput mode */
#define SR_FMT16 16
#define SR_FMT8 8

int sr_version(void);
const char* sr_last_error(void);

/* ---- stream geometry (host-side, no GPU needed) -------------------------------------------------
 * The fused MLP consumes its weights as one linear "stream" of 1-KiB MFMA-A-fragment pieces in
 * consumption order.  These return the element counts the host packer
 * (satnerf_amd/packing.py) must produce for a given network shape; -1 on an unsupported shape. */
int64_t sr_fwd_stream_elems(int feat, int tau);  /* bf16 elements of the forward stream (per hi/lo plane) */
int64_t sr_bwd_stream_elems(int feat, int tau);  /* bf16 elements of the transposed (dX) stream          */
/* training workspaces, per 32-point tile, in 16-bit elements, for a workspace format `fmt` (below) */
int64_t sr_act_elems_per_tile(int feat, int fmt);   /* activations saved by the forward pass          */
int64_t sr_dpre_elems_per_tile(int feat, int fmt);  /* pre-activation gradients written by the dX pass */
/* tiles both workspaces must hold for n_points points: ceil(n_points / 32) rounded up to the 8 tiles of a workgroup -- the forward and
 * dX kernels run whole workgroups, and the waves past the last point still store their (unused) tile */
int64_t sr_workspace_tiles(int64_t n_points);
/* 16-bit elements of the WHOLE dpre workspace for n_points points = sr_workspace_tiles * sr_dpre_elems_per_tile, plus (SR_FMT8) the table
 * of exponent maxima the dX pass leaves behind the last tile for the weight-gradient pass: 16 bytes per 4 tiles, byte g = the largest MX8
 * exponent byte of scale group g over those tiles (csrc/mlp_layout.h), rounded up to whole KiB.  -1 on an unsupported shape. */
int64_t sr_dpre_workspace_elems(int64_t n_points, int feat, int fmt);

/* ---- weight packing:  replaces nothing in the reference (its weights feed addmm directly) ---------
 * out_hi[i] = bf16_rne(src[idx[i]] * scale[i]);  out_lo[i] = bf16_rne(src[idx[i]]*scale[i] - out_hi[i])
 * (out_lo may be NULL).  idx < 0 selects the constant 0. */
int sr_pack_stream(const float* src, const int32_t* idx, const float* scale, int64_t n,
                   uint16_t* out_hi, uint16_t* out_lo, int64_t n_f16, void* stream);
/* n_f16: the first n_f16 elements of out_hi are written as fp16 instead of bf16 (the forward stream of SR_MODE_F16; even) */
/* the same plus an fp32 gather (sr_gather_scale_f32) in ONE launch: the forward stream, the backward stream (concatenated
 * maps) and the fc_net.0 table are refreshed together after every optimizer step; `tick` (may be NULL) is a 1-float device
 * counter the launch increments: the step count sr_adam_step_graph reads later in the same captured graph */
int sr_pack_all(const float* src, const int32_t* idx, const float* scale, int64_t n, uint16_t* out_hi, uint16_t* out_lo,
                const int32_t* f32_idx, const float* f32_scale, int64_t n_f32, float* out_f32, float* tick, int64_t n_f16,
                void* stream);

/* out[i] = src[idx[i]] * scale[i] in fp32 (idx < 0 -> 0): builds the fc_net.0 table `l0` of sr_satnerf_mlp_fwd */
int sr_gather_scale_f32(const float* src, const int32_t* idx, const float* scale, int64_t n, float* out, void* stream);

/* grad[e] (+)= gscale[e] * (sum over the split-K slices of partial element gidx[e])  (gidx < 0: left untouched) -- reduces
 * the slices of sr_satnerf_wgrad (`blocks` = its planned job table, device) and scatters into the flat parameter-gradient
 * buffer (inverse of sr_pack_stream's gather) */
int sr_unpack_grads(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks,
                    float* grad, int accumulate, void* stream);

/* ---- stratified sampling: rendering.py:62-78 -----------------------------------------------------
 * rays (N, ray_stride>=8) with near at column 6, far at column 7; u (N,S) in [0,1) -> z_vals (N,S). */
int sr_ray_sample_fwd(const float* rays, int ray_stride, const float* u, int64_t n_rays, int n_samples,
                      float* z_vals, void* stream);

/* ---- sky-colour head, once per ray: models/satnerf.py:138-143,201 ---------------------------------
 * sun (N, sun_stride) -> sky (N,3) = sigmoid(W2 relu(W1 sun + b1) + b2);  W1 (H,3) b1 (H) W2 (3,H) b2 (3) */
int sr_sky_fwd(const float* sun, int sun_stride, int64_t n, int hidden, const float* w1, const float* b1,
               const float* w2, const float* b2, float* sky, void* stream);

/* ---- fused SatNeRF MLP over points: models/satnerf.py:20-40 (repeat_interleave + chunk loop) and
 *      SatNeRF.forward models/satnerf.py:156-208 (+ Siren models/nerf.py:23-33) ------------------------
 * point p (0 <= p < n_points) belongs to row r = p / n_samples of the per-ray arrays:
 *   xyz   = org[r] + dir[r] * z[p]           (rendering.py:81; dir==NULL or z==NULL -> xyz = org[r])
 *   sun_d = sun[r],  t = temb[ts ? ts[r] : r]  (rendering.py:99-100, nn.Embedding lookup)
 * outputs (any may be NULL): albedo (P,3), sigma (P), sun_v (P), beta (P)   [models/satnerf.py:45-49]
 * stream_hi/lo: packed forward stream from sr_pack_stream (lo required iff mode == SR_MODE_BF16X3)
 * l0: (feat,4) fp32 rows [w_x, w_y, w_z, b] of fc_net.0, each multiplied by 30/(2*pi), in slot order
 * acts: NULL, or training workspace of sr_act_elems_per_tile(feat, act_fmt) * sr_workspace_tiles(P) 16-bit elements, written in the
 * format act_fmt (SR_FMT8 needs mode == SR_MODE_BF16; act_fmt is ignored when acts == NULL). */
typedef struct sr_mlp_inputs {
  const float* org;
  int org_stride;
  const float* dir;
  int dir_stride;
  const float* sun;
  int sun_stride;
  const float* z;
  const float* temb;
  const int64_t* ts;
  int64_t n_points;
  int n_samples;
} sr_mlp_inputs;

int sr_satnerf_mlp_fwd(const sr_mlp_inputs* in, int feat, int tau, int mode, const uint16_t* stream_hi,
                       const uint16_t* stream_lo, const float* l0, float* albedo, float* sigma, float* sun_v,
                       float* beta, uint16_t* acts, int act_fmt, void* stream);

/* ---- one-launch render pass (no grad): rendering.py:62-81 (stratified sampling) -> models/satnerf.py:20-49 (the MLP over
 *      every sample) -> models/satnerf.py:52-70 (compositing) + the per-ray sky head (:138-143) -- the same arithmetic as
 *      sr_ray_setup -> sr_satnerf_mlp_fwd -> sr_composite_fwd, bit for bit, with the per-point values handed from the MLP to
 *      the compositing through LDS: a workgroup owns sr_render_points_per_block(feat, mode) consecutive points = whole rays,
 *      so n_samples must divide that number (64 and 128 do; other S: use the three separate launches).
 * rays (N, ray_stride >= 11) = o d near far sun; ts (N) image indices into temb (vocab, tau).
 * depths: z_in (N,S) given (the fine pass), else stratified with jitter u (N,S), else (both NULL) with jitter drawn in the kernel:
 *   Philox keyed by `seed`, stepping with step_counter[0]; tick != 0: the launch advances that counter (see sr_ray_setup_rng).
 * noise (N,S) or NULL with noise_std; sky_*: the sky head's weights (hidden, 3), (hidden), (3, hidden), (3).
 * bank_chunks > 0: `rays` / `ts` are a resident bank of bank_chunks x N rows and the launch renders chunk
 *   (step_counter[0] mod bank_chunks) of it -- with tick, a captured launch walks the bank chunk by chunk, replay after
 *   replay, as eval_satnerf.py:46-66 walks an image, with no host work and no gather in between (outputs: the chunk's N rays).
 * outputs: z_out (N,S) [NULL ok], albedo (N,S,3), sun_v (N,S), beta (N,S), sigma (N,S) [NULL ok], sky (N,3), weights (N,S),
 *   transparency (N,S), depth (N), rgb (N,3) clamped to [0,1]. */
typedef struct sr_render_args {
  const float* rays;
  int ray_stride;
  const int64_t* ts;
  const float* temb;
  int64_t n_rays;
  int n_samples;
  const float* z_in;
  const float* u;
  uint64_t seed;
  float* step_counter;
  int tick;
  const float* noise;
  float noise_std;
  int sky_hidden;
  const float* sky_w1;
  const float* sky_b1;
  const float* sky_w2;
  const float* sky_b2;
  int64_t bank_chunks;
} sr_render_args;

typedef struct sr_render_outputs {
  float* z_vals;
  float* albedo;
  float* sigma;
  float* sun_v;
  float* beta;
  float* sky;
  float* weights;
  float* transparency;
  float* depth;
  float* rgb;
} sr_render_outputs;

int sr_satnerf_render_fwd(const sr_render_args* in, int feat, int tau, int mode, const uint16_t* stream_hi, const uint16_t* stream_lo,
                          const float* l0, const sr_render_outputs* out, void* stream);
/* points a workgroup of the fused kernel owns (256 / 128), or -1 when (feat, mode) has no fused build */
int sr_render_points_per_block(int feat, int mode);

/* ---- the training forward in ONE launch: replaces main.py:60-75,127 + metrics.py:21-25,36-44,56-73 + autograd through the compositing ---
 * sr_satnerf_render_train = sr_ray_setup + sr_satnerf_mlp_fwd (saving the SR_FMT8 activations) + sr_render_loss: the render pass above
 * with, in its epilogue, the colour loss (SatNerfLoss; SNerfLoss while sched[2] != 0) and the closed-form compositing backward of every
 * ray by the wave that composited it (same per-ray function as sr_render_loss: bit-identical to the three launches).  n_samples <= 64,
 * SR_MODE_BF16 / SR_MODE_F16, widths 256 and 512.  `out`: albedo / sigma / sun_v / beta (N,S[,3]) and sky (N,3) are required (the dX
 * pass and the sky head's backward read them), z_vals optional, weights / transparency / depth / rgb unused.  `train` outputs: loss_parts
 * (one per workgroup = ceil(N * S / sr_render_points_per_block) floats: the loss is their sum), rgb (N,3) or NULL, d_sigma (N,S),
 * d_albedo (N,S,3), d_sun_v (N,S), g_beta (N,S), d_sky (N,3).  acts: the SR_FMT8 workspace of sr_satnerf_mlp_fwd.
 * r05, the batch sampler inside the launch (replaces the DataLoader of main.py:96-110 as sr_gather_batch does, without its launch):
 * gather_idx != NULL: `in->rays` (stride 11), `in->ts` and `target` are the RESIDENT RAY BANK, gather_idx holds the shuffled row indices
 * of a whole epoch = `batches` x N entries, and the launch trains on batch cursor[0] of it -- ray r of the launch is bank row
 * gather_idx[cursor[0] * N + r] -- then moves the cursor on modulo `batches` (cursor = a 4-float block as sr_gather_batch's).  The wave
 * that composites a ray also copies its row to out_rays (N,11) / out_rgbs (N,3) / out_ts (N): the batch as the later launches of the
 * step (sky-head and embedding gradients, solar-correction pass) read it.
 * in->tick == 2 ("tick first", training launches only): the launch advances the step counter as tick == 1 does AND draws its jitter for
 * the advanced value -- it is the first launch of a step that has no sr_pack_all in front of it to tick. */
typedef struct sr_train_args {
  const float* target;   /* (N,3), or the bank's colours (rows, 3) with gather_idx */
  const float* sched;    /* 4-float schedule block or NULL */
  float beta_min;
  float* loss_parts;
  float* rgb;
  float* d_sigma;
  float* d_albedo;
  float* d_sun_v;
  float* g_beta;
  float* d_sky;
  const int64_t* gather_idx;
  float* cursor;
  int64_t batches;
  float* out_rays;
  float* out_rgbs;
  int64_t* out_ts;
} sr_train_args;
int sr_satnerf_render_train(const sr_render_args* in, int feat, int tau, int mode, const uint16_t* stream_hi, const uint16_t* stream_lo,
                            const float* l0, const sr_render_outputs* out, const sr_train_args* train, uint16_t* acts, int act_fmt, void* stream);

/* ---- backward of the fused MLP: replaces autograd through SatNeRF.forward (models/satnerf.py:156-208) ------------
 * sr_satnerf_mlp_bwd: data-gradient chain.  Inputs: the forward's saved `acts`, its four outputs and the gradients of
 * those outputs (g_* may be NULL = 0); bwd_stream = packed transposed weights (sr_pack_stream with
 * packing.backward_maps).  Outputs: dpre (sr_dpre_workspace_elems(P, feat, fmt) 16-bit elements) and d_t (P,tau)
 * fp32, the gradient w.r.t. each point's embedding vector (NULL to skip).  `fmt` = format of BOTH workspaces.
 * sr_satnerf_wgrad: weight-gradient GEMMs dpre x acts over all points.  `blocks` (n_blocks x 12 int32, device) lists the job
 * blocks (rf0 nr0 rf1 nr1 | cf0 nc0 cf1 nc1 | col_kind n_slices first_slice -: up to two ranges of dpre row fragments and of
 * activation column fragments, <= 16 each; packing.backward_maps); every block is cut into n_slices contiguous ranges of
 * 32-point tiles (split-K), one workgroup each, writing slice s to partial + s * (256*256 + 256*32) floats; reduce with
 * sr_unpack_grads.
 * sr_wgrad_plan (host, no GPU work): fills n_slices / first_slice of a HOST copy of the table for n_points points and at
 * most n_wg workgroups (n_wg <= 0: the current device's CU count); *n_slices = total slices.  `fmt` = the workspace format of
 * the kernel that will run the plan: SR_FMT16 cuts every block into the same number of slices (that kernel's time per tile
 * does not depend on the block), SR_FMT8 hands the workgroups out by a per-block cost x tiles (equal costs for the default 4-wave
 * kernel; SATNERF_WGRAD_V1=1: the r02 kernel's decode work grows with the fragments a block moves). */
int sr_satnerf_mlp_bwd(int feat, int tau, int64_t n_points, const uint16_t* bwd_stream, const uint16_t* acts,
                       const float* albedo, const float* sigma, const float* sun_v, const float* beta, const float* g_albedo,
                       const float* g_sigma, const float* g_sun_v, const float* g_beta, uint16_t* dpre, float* d_t, int fmt,
                       void* stream);
int sr_wgrad_plan(int32_t* blocks, int n_blocks, int64_t n_points, int n_wg, int fmt, int* n_slices);
int sr_satnerf_wgrad(int feat, int tau, int64_t n_points, const uint16_t* dpre, const uint16_t* acts, const int32_t* blocks,
                     int n_blocks, int n_slices, float* partial, void* stream);
/* the same contraction from SR_FMT8 workspaces (autograd's grad_weight / grad_bias of every nn.Linear, models/satnerf.py:104-153).
 * Default kernel (csrc/wgrad9.hip): 4 waves per job block, each wave decodes its 8-bit double fragments in registers (PHASE8 -> fp16
 * sine, MX8 -> fp16 value scaled per workgroup) and contracts 128 x 128 register tiles with fp16 MFMAs; SATNERF_WGRAD_V1=1 (and
 * workspaces of 4 GiB or more) run the r02 kernel, which expands the fragments in the LDS to bf16.  `loads` (n_blocks x
 * sr_wgrad8_load_ints() int32, device; packing.wgrad8_loads) says which unit of which workspace each wave of a block fetches and where
 * it goes -- ints 0..19 for the r02 kernel, 20..112 the duty table, the exponent groups of the row pairs, the quadrant mask and the per-wave stream variants (r06: thin streams for one-row blocks) of the default one
 * (which reads the exponent maxima behind the dpre workspace: dpre must be the buffer sr_satnerf_mlp_bwd wrote, sr_dpre_workspace_elems long); `blocks` is
 * the same planned table.  sr_wgrad_plan hands the default kernel equal slices (it runs one instruction stream for every block) and the
 * r02 kernel cost-weighted ones.
 * plan_span = int 11 of the planned table's first row (sr_wgrad_plan): 0 = one slice per workgroup; > 0 = a stream-K plan (the 4-wave kernel
 * walks `plan_span` tile units of the block-major job list per workgroup, writing one partial block per block it touches). */
/* dpre_elems = 16-bit elements the caller's dpre buffer holds: rejected below sr_dpre_workspace_elems(n_points, feat, SR_FMT8) */
int sr_satnerf_wgrad8(int feat, int tau, int64_t n_points, const uint16_t* dpre, int64_t dpre_elems, const uint16_t* acts,
                      const int32_t* blocks, const int32_t* loads, int n_blocks, int n_slices, int plan_span, float* partial, void* stream);
int sr_wgrad8_load_ints(void);

/* parameter gradients of the sky head (atomicAdd into g_*; zero them first) and of the embedding table
 * g_emb[ts[r]] += sum_j d_t[r*S + j] (nn.Embedding backward, rendering.py:100) */
int sr_sky_bwd(const float* sun, int sun_stride, int64_t n, int hidden, const float* w1, const float* b1, const float* w2,
               const float* sky, const float* d_sky, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* stream);
int sr_embedding_bwd(const float* d_t, const int64_t* ts, int64_t n_rays, int n_samples, int tau, float* g_emb, void* stream);

/* ---- fused fast-path launches of the training step (same arithmetic as the separate entry points) -------------------
 * sr_ray_setup = sr_ray_sample_fwd + sr_sky_fwd (rays need >= 11 columns);
 * sr_render_loss = sr_composite_fwd -> sr_satnerf_loss -> sr_composite_bwd for n_samples <= 64, one wave per ray:
 *   outputs the MLP-backward inputs d_sigma (N,S), d_albedo (N,S,3), d_sun_v (N,S), g_beta (N,S), the per-ray d_sky (N,3),
 *   the loss partial sums (ceil(N/4)) and optionally the rendered rgb (N,3). */
/* sr_depth_loss: metrics.DepthLoss for the coarse model (metrics.py:75-92; main.py:134-141): value = sum(loss_parts[0 ..
 * ceil(N/256))) = lambda_ds/3 * mean(w * (depth - target)^2), g_depth (N) its gradient; depths (N, stride) = [target, weight, ...],
 * use_weights = 0 is --ds_noweights. */
/* sr_sc_loss: the solar-correction terms of SNerfLoss / SatNerfLoss (metrics.py:27-34) for the pass rendered along the sun
 * direction (rendering.py:102-108): compositing of that pass (z_vals, sigma[, noise]) -> value = sum(loss_parts[0 .. ceil(N/4))) =
 * lambda_sc/3 * (mean_r sum_j (T_j - sun_j)^2 + mean_r (1 - sum_j w_j sun_j)), d_sun_v (N,S) its gradient (T, w detached). */
int sr_sc_loss(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* sun_v, int64_t n_rays,
               int n_samples, float lambda_sc, float* loss_parts, float* d_sun_v, void* stream);
int sr_depth_loss(const float* depth, const float* depths, int depths_stride, int use_weights, int64_t n_rays, float lambda_ds,
                  float* loss_parts, float* g_depth, void* stream);
int sr_ray_setup(const float* rays, int ray_stride, const float* u, int64_t n_rays, int n_samples, int hidden, const float* w1,
                 const float* b1, const float* w2, const float* b2, float* z_vals, float* sky, void* stream);
/* the same with the stratified jitter u ~ U[0,1) (rendering.py:77) drawn INSIDE the kernel: Philox-4x32-10 keyed by `seed`,
 * counter = (ray, sample, step) with the step read from the device counter `step_counter[0]` (NULL = 0; the counter
 * sr_pack_all ticks) -- a captured training step then needs no RNG launch.  tick != 0: this launch also advances the counter
 * (captured forward passes have no sr_pack_all): `step_counter` is then a 4-float block as `sched`, [0] += 1 once every
 * workgroup has read it, [3] is scratch (an arrival counter, zero-initialised by the caller) */
int sr_ray_setup_rng(const float* rays, int ray_stride, uint64_t seed, float* step_counter, int tick, int64_t n_rays, int n_samples,
                     int hidden, const float* w1, const float* b1, const float* w2, const float* b2, float* z_vals, float* sky,
                     void* stream);
int sr_render_loss(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                   const float* sun_v, const float* beta, const float* sky, const float* target, int64_t n_rays, int n_samples,
                   float beta_min, const float* sched, float* loss_parts, float* rgb, float* d_sigma, float* d_albedo, float* d_sun_v,
                   float* g_beta, float* d_sky, void* stream);

/* `sched` (NULL = none) is the DEVICE-side schedule block of a captured training step, 4 floats the host updates between graph
 * replays: [0] 1-based optimizer step (ticked by sr_pack_all), [1] learning rate (sr_adam_step_graph with lr < 0 reads it:
 * StepLR(gamma 0.9) per epoch, main.py:86-94 / train_utils.py:41-57), [2] != 0 while the SNerfLoss warm-up lasts (the colour loss
 * is then the plain MSE of metrics.SNerfLoss, main.py:128-131 -- no beta term, no beta gradient), [3] reserved.
 * sr_grad_tail = sr_unpack_grads + sr_sky_bwd + sr_embedding_bwd as three block ranges of one launch (training fast path;
 * n_blocks = rows of the planned job table `blocks`);
 * sr_adam_step_graph = sr_adam_step with the 1-based step count read from the device (state[0], advanced by sr_pack_all's
 * `tick` earlier in the same step) so the launch can be replayed from a hipGraph. */
int sr_grad_tail(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks, int n_blocks,
                 float* grad, int accumulate, const float* sun, int sun_stride, int64_t n_rays, int hidden,
                 const float* w1, const float* b1, const float* w2, const float* sky, const float* d_sky, float* g_w1, float* g_b1,
                 float* g_w2, float* g_b2, const float* d_t, const int64_t* ts, int n_samples, int tau, float* g_emb, void* stream);
/* sr_grad_tail + sr_adam_step_graph in one launch (the single-GPU captured step): the thread that reduces a parameter's slices applies
 * torch.optim.Adam to it (params / exp_avg / exp_avg_sq are aligned with grad: element i of all four is parameter i) and zeroes its
 * gradient slot; `late_idx` (n_late flat indices relative to grad) lists the parameters whose gradients arrive by atomics -- the sky head
 * and the embedding rows: the last atomics block to finish updates them.  `state` = the 4-float schedule block ([0] 1-based step, [1]
 * rate used when lr < 0, [3] arrival counter, zero between launches).
 * `pack` (NULL or pack->map == NULL: none), r05: the launch ALSO keeps the weight streams current -- the thread that updated parameter i
 * writes it to its (at most two) places in the packed streams, so the next step needs no sr_pack_all launch.  map = 2 int32 per
 * parameter (packing.pack_scatter_map): position | scale index << 26 | (1 << 28: the fp32 fc_net.0 table `l0`), -1 = none; positions
 * address `hi` (bf16; fp16 below n_f16 as sr_pack_all's n_f16) and, if not NULL, `lo` (the bf16x3 remainder plane); the value written
 * is param * scales[index], the element-wise arithmetic of sr_pack_all: the streams hold the same bits as a fresh sr_pack_all. */
typedef struct sr_pack_scatter {
  const int32_t* map;
  uint16_t* hi;
  uint16_t* lo;
  float* l0;
  int64_t n_f16;
  float scales[4];
} sr_pack_scatter;
int sr_grad_tail_adam(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks, int n_blocks,
                      float* grad,
                      int accumulate, const float* sun, int sun_stride, int64_t n_rays, int hidden, const float* w1, const float* b1,
                      const float* w2, const float* sky, const float* d_sky, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                      const float* d_t, const int64_t* ts, int n_samples, int tau, float* g_emb, float* params, float* exp_avg,
                      float* exp_avg_sq, const int32_t* late_idx, int n_late, float* state, float lr, float beta1, float beta2, float eps,
                      float grad_scale, const sr_pack_scatter* pack, void* stream);
int sr_adam_step_graph(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, float grad_scale, float* state, int zero_grad, void* stream);
/* sr_adam_step_graph that also re-packs (r06: the data-parallel step = the single-GPU step + one collective): Adam on all n elements of
 * the flat buffers, and each of the first n_packed parameters (the coarse model's; `pack->map` holds n_packed x 2 words) written into its
 * places of the weight streams as sr_grad_tail_adam does.  Sequence at N > 1: ... sr_grad_tail | all-reduce of `grads` | sr_adam_step_pack
 * -- no sr_pack_all, no separate update launch.  pack may be NULL (n_packed 0): plain sr_adam_step_graph arithmetic. */
int sr_adam_step_pack(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                      float eps, float grad_scale, float* state, int zero_grad, int64_t n_packed, const sr_pack_scatter* pack,
                      void* stream);

/* ---- layer-by-layer path for widths the fused kernel does not cover (fc_units != 256; opt.py:50 defaults to 512) ----
 * One nn.Linear of models/satnerf.py:104-153 per call, its input being the concatenation of one or two sources with the
 * PREVIOUS layer's activation applied on load (tensors between layers are pre-activations):
 *   acat[p][f] = act_s(src_s.x[(p / row_div_s) * ld_s + f - off_s]),  act = none | sin(w0 x) (Siren, models/nerf.py:23-33) | relu
 *   sr_linear_fwd:        y[p][n] = out_act(sum_f acat[p][f] * weight[n][f] + bias[n])          (weight (n_out, K) row-major)
 *   sr_linear_bwd_input:  d_src[p][k] = (sum_n G[p][n] * weight[n][col0 + k]) * act'(target.x[p][k])   (autograd grad_input)
 *   sr_linear_bwd_weight: d_weight[n][f] += sum_p G[p][n] * acat[p][f],  d_bias[n] += sum_p G[p][n]     (fp32 atomics)
 * with G[p][n] = gy[p][n] * out_act'(y[p][n]) (y = the activated forward output; may be NULL for SR_OUT_NONE).
 * out_act: softplus (sigma, beta), sigmoid (sun), sigmoid*1.002-0.001 (rgb, models/satnerf.py:195-196).
 * fp32 in / fp32 out, bf16 hi/lo split 3-pass MFMA inside (~1e-6 relative).  sr_points_along: xyz[p] = o + d * z. */
enum { SR_ACT_NONE = 0, SR_ACT_SIN = 1, SR_ACT_RELU = 2 };
enum { SR_OUT_NONE = 0, SR_OUT_SOFTPLUS = 1, SR_OUT_SIGMOID = 2, SR_OUT_SIGMOID_RGB = 3 };
typedef struct sr_linear_src {
  const float* x; /* (ceil(P / row_div), >= k) fp32 */
  int ld;         /* row stride in floats */
  int k;          /* columns taken from this source */
  int act;        /* SR_ACT_* applied on load */
  float w0;       /* sin(w0 * x) */
  int row_div;    /* point p reads row p / row_div: 1 = per point, n_samples = per ray */
} sr_linear_src;
int sr_linear_fwd(const sr_linear_src* src, int n_src, const float* weight, const float* bias, int64_t n_points, int n_out,
                  int out_act, float* y, int ldy, void* stream);
int sr_linear_bwd_input(const float* gy, int ldg, const float* y, int ldy, int out_act, const float* weight, int k_total, int col0,
                        const sr_linear_src* target, int64_t n_points, int n_out, float* d_src, int ldd, void* stream);
int sr_linear_bwd_weight(const float* gy, int ldg, const float* y, int ldy, int out_act, const sr_linear_src* src, int n_src,
                         int64_t n_points, int n_out, float* d_weight, float* d_bias, void* stream);
int sr_points_along(const float* rays, int ray_stride, int dir_col, const float* z_vals, int64_t n_rays, int n_samples, float* xyz,
                    void* stream);
/* Mapping.forward of the classic nerf (models/nerf.py:36-69): x (rows, dim) -> out (rows, 2*n_freqs*dim) =
 * [sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...], no identity term */
int sr_positional_map(const float* x, int ld, int dim, int64_t rows, int n_freqs, float* out, void* stream);

/* ---- whole-image evaluation -------------------------------------------------------------
 * sr_composite_image: compositing (models/satnerf.py:52-70) that keeps per ray only what
 * eval_satnerf.save_nerf_output_to_images writes per pixel (eval_satnerf.py:106-146): image (N,13) =
 * [rgb(3) clamped, depth, sum w, sum w*sun, sum w*albedo(3), sum w*beta, sum w*sky(3)] -- 52 B/ray instead of 2,576 B.
 * sr_latlonalt_from_depth: SatelliteDataset.get_latlonalt_from_nerf_prediction (datasets/satellite.py:246-275) with
 * sat_utils.ecef_to_latlon_custom (sat_utils.py:76-95), in fp64: point = (o + d*depth) * range + center (ECEF, `center` =
 * 3 HOST doubles) -> lat, lon (degrees), alt (m), each (N,) fp64 on the device. */
int sr_composite_image(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                       const float* sun_v, const float* beta, const float* sky, int64_t n_rays, int n_samples, float* image,
                       void* stream);
int sr_latlonalt_from_depth(const float* rays, int ray_stride, const float* depth, int64_t n_rays, const double* center,
                            double range, double* lat, double* lon, double* alt, void* stream);

/* ---- RPC ray generation: datasets/satellite.py:18-65 get_rays + :218-227 normalize_rays + :229-244 sun -------
 * `rpc` = 90 HOST doubles of the image's RPC00B camera in rpcm's dict order: row_num[20], row_den[20], col_num[20], col_den[20],
 * row_offset, col_offset, lat_offset, lon_offset, alt_offset, row_scale, col_scale, lat_scale, lon_scale, alt_scale (apply
 * sat_utils.rescale_rpc for a down-scaled image first).  Pixel i = (col i % width, row i / width) is localised at max_alt (ray
 * origin) and min_alt in fp64 (Newton on the projection to rpcm's 1e-18 tolerance), converted to ECEF, and written as
 * rays8 (H*W, 8) fp32 = the reference's <cache_dir>/<img_id>.data content [o(3) d(3) near=0 far] and / or rays11 (H*W, 11) = the
 * same normalised by `center` (3 HOST doubles) / `range` in fp32 with the image's sun direction appended; either may be NULL. */
int sr_rpc_rays(const double* rpc, int width, int height, double min_alt, double max_alt, const double* center, double range,
                double sun_elevation_deg, double sun_azimuth_deg, float* rays11, float* rays8, void* stream);

/* ---- training-step kernels --------------------------------------------------------------
 * sr_satnerf_loss: metrics.SatNerfLoss for the coarse model (metrics.py:21-25,56-73): value = sum of
 * loss_parts[0 .. ceil(N/4)), and grad_scale * dLoss/d{rgb (N,3), weights (N,S), beta (N,S)} in g_*.
 * sr_adam_step: torch.optim.Adam update (main.py:84) over a flat buffer; `step` is the 1-based step count; grads are
 * multiplied by grad_scale first and zeroed afterwards when zero_grad != 0. */
int sr_satnerf_loss(const float* rgb, const float* weights, const float* beta, const float* target, int64_t n_rays,
                    int n_samples, float beta_min, float grad_scale, const float* sched, float* loss_parts, float* g_rgb,
                    float* g_weights, float* g_beta, void* stream);
/* batch gather from a GPU-resident ray bank: rows idx[0..n) of rays (.,11), rgbs (.,3), ts (.) -> contiguous batch tensors
 * (replaces DataLoader collate + host-to-device copy, main.py:96-110).  cursor != NULL (a zero-initialised 4-float device
 * block, [3] scratch): idx holds the shuffled indices of a whole epoch = `batches` x n entries, the launch gathers batch
 * cursor[0] and advances the cursor modulo `batches` -- inside a captured training step the sampler needs no host work */
int sr_gather_batch(const float* rays, const float* rgbs, const int64_t* ts, const int64_t* idx, int64_t n, float* out_rays,
                    float* out_rgbs, int64_t* out_ts, float* cursor, int64_t batches, void* stream);
/* sr_gather_batch + sr_ray_setup_rng in ONE launch (one wave per ray) for captured steps that sample for themselves: the rows go to
 * the static batch tensors, z_vals (n, n_samples) and sky (n,3) are computed from the row just read; the jitter step is
 * step_counter[0] + step_offset (the launch runs before sr_pack_all ticks that counter: offset 1 reproduces sr_ray_setup_rng's draws) */
int sr_gather_setup(const float* rays, const float* rgbs, const int64_t* ts, const int64_t* idx, int64_t n, float* out_rays,
                    float* out_rgbs, int64_t* out_ts, float* cursor, int64_t batches, int n_samples, int hidden, const float* w1,
                    const float* b1, const float* w2, const float* b2, float* z_vals, float* sky, uint64_t seed,
                    const float* step_counter, int step_offset, void* stream);
int sr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, float grad_scale, int64_t step, int zero_grad, void* stream);

/* ---- sigma -> alpha compositing: models/satnerf.py:52-70 ------------------------------------------
 * noise may be NULL (== noise_std 0).  sky is per ray (N,3).  Outputs: weights, transparency (N,S),
 * depth (N), rgb (N,3) (clamped to [0,1] when clamp_rgb != 0; the classic nerf variant does not clamp,
 * models/nerf.py:128, and passes sun_v = sky = NULL). */
int sr_composite_fwd(const float* z_vals, const float* sigma, const float* noise, float noise_std,
                     const float* albedo, const float* sun_v, const float* sky, int64_t n_rays, int n_samples,
                     int clamp_rgb, float* weights, float* transparency, float* depth, float* rgb, void* stream);

/* closed-form backward of sr_composite_fwd.  Upstream grads g_* may be NULL (=0).
 * Produces d_sigma (N,S), d_albedo (N,S,3), d_sun_v (N,S), d_sky (N,3); any output may be NULL. */
int sr_composite_bwd(const float* z_vals, const float* sigma, const float* noise, float noise_std,
                     const float* albedo, const float* sun_v, const float* sky, const float* weights,
                     const float* transparency, const float* rgb_unclamped_or_null, int64_t n_rays, int n_samples,
                     int clamp_rgb, const float* g_rgb, const float* g_depth, const float* g_weights,
                     const float* g_transparency, float* d_sigma, float* d_albedo, float* d_sun_v, float* d_sky,
                     void* stream);

/* ---- importance resampling + merge: rendering.py:10-49 and :121-125 -------------------------------
 * z_coarse (N,S) sorted, weights_coarse (N,S), u (N,I) -> z_fine (N,S+I) = sort(cat(z_coarse, z_new)). */
int sr_sample_pdf_merge(const float* z_coarse, const float* weights_coarse, const float* u, int64_t n_rays,
                        int n_samples, int n_importance, float eps, float* z_fine, void* stream);
/* stand-alone rendering.sample_pdf (rendering.py:10-49): bins (N,nb), weights (N,nb-1), u (N,I) -> samples (N,I) */
int sr_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins, int n_importance,
                  float eps, float* samples, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SATRENDER_H */
