// Training-step kernels around the renderer: the Sat-NeRF loss with its gradient in one launch,
// and Adam over the flat parameter buffer in one launch.  Both are HBM-bound streaming kernels over <= 3 MB.
#include <math.h>

#include "common.h"
#include "adam_device.h"
#include "ray_device.h"

namespace sr {

// device-side schedule block of the captured training step (4 floats, satnerf_amd.train.Trainer.sched): [0] 1-based optimizer
// step (ticked by sr_pack_all), [1] learning rate, [2] != 0 while the SNerfLoss warm-up epochs last, [3] reserved
enum { kSchedStep = 0, kSchedLr = 1, kSchedWarm = 2 };

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// metrics.SatNerfLoss for the coarse model (metrics.py:21-25,56-73), forward value AND the gradients it sends back:
//   beta_r = sum_j w_rj b_rj + beta_min
//   loss   = mean_{r,c} (rgb_rc - gt_rc)^2 / (2 beta_r^2)  +  (3 + mean_r log beta_r) / 2
// One wave per ray, 4 rays per block.  Block b writes its share of the loss to loss_parts[b] (no atomics, no memset; the
// value is sum(loss_parts) and only logging reads it); g_* are the gradients w.r.t. rgb (N,3), weights (N,S), beta (N,S).
__global__ void __launch_bounds__(256) satnerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                          const float* __restrict__ beta, const float* __restrict__ target, long n_rays, int S,
                                                          float beta_min, float grad_scale, const float* __restrict__ sched,
                                                          float* __restrict__ loss_parts, float* __restrict__ g_rgb,
                                                          float* __restrict__ g_weights, float* __restrict__ g_beta) {
  __shared__ float part[4];
  const bool warm = sched != nullptr && sched[kSchedWarm] != 0.f;  // SNerfLoss epochs (main.py:128-131): plain MSE, no beta
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wv;
  if (lane == 0) part[wv] = 0.f;
  const bool on = r < n_rays;
  float acc = 0.f;
  if (on)
    for (int j = lane; j < S; j += 64) acc += weights[r * S + j] * beta[r * S + j];
  const float b = wave_sum_f(acc) + beta_min;
  const float inv_n = 1.0f / (float)n_rays;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (on) d0 = rgb[r * 3] - target[r * 3], d1 = rgb[r * 3 + 1] - target[r * 3 + 1], d2 = rgb[r * 3 + 2] - target[r * 3 + 2];
  const float sq = d0 * d0 + d1 * d1 + d2 * d2;
  const float ib2 = warm ? 2.0f : 1.0f / (b * b);  // MSE = the same expression with beta^2 = 1/2 and no log term
  if (lane == 0 && on) {
    float contrib = sq * ib2 * (0.5f / 3.0f) * inv_n + (warm ? 0.f : 0.5f * logf(b) * inv_n);
    if (r == 0 && !warm) contrib += 1.5f;
    part[wv] = contrib;
    const float k = ib2 * (1.0f / 3.0f) * inv_n * grad_scale;
    g_rgb[r * 3] = d0 * k, g_rgb[r * 3 + 1] = d1 * k, g_rgb[r * 3 + 2] = d2 * k;
  }
  __syncthreads();
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
  if (!on) return;
  // d loss / d beta_r = -sq / (3 N beta^3) + 1 / (2 N beta)
  const float db = warm ? 0.f : (-sq * ib2 / b * (1.0f / 3.0f) + 0.5f / b) * inv_n * grad_scale;
  for (int j = lane; j < S; j += 64) {
    g_weights[r * S + j] = db * beta[r * S + j];
    g_beta[r * S + j] = db * weights[r * S + j];
  }
}

// ---- metrics.DepthLoss (metrics.py:75-92), coarse model: lambda_ds/3 * mean(w * (depth - target)^2) and its gradient ------------
// depths (N, stride) = [target, weight, ...] (datasets/satellite_depth.py); use_weights = 0 is ds_noweights (main.py:137).
// Block b writes its share of the value to loss_parts[b].
__global__ void __launch_bounds__(256) depth_loss_kernel(const float* __restrict__ depth, const float* __restrict__ depths, int stride,
                                                        int use_weights, long n, float lam, float* __restrict__ loss_parts,
                                                        float* __restrict__ g_depth) {
  __shared__ float part[4];
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (r < n) {
    const float diff = depth[r] - depths[r * stride];
    const float w = use_weights ? depths[r * stride + 1] : 1.f;
    const float inv_n = 1.0f / (float)n;
    contrib = lam * w * diff * diff * inv_n;
    g_depth[r] = 2.0f * lam * w * diff * inv_n;
  }
  contrib = wave_sum_f(contrib);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// ---- fused per-ray training kernel: compositing forward -> SatNerf loss -> compositing backward ---------------------------
// One wave per ray (lane = sample, S <= 64).  Replaces models/satnerf.py:52-70 + metrics.py:21-25,56-73 + their autograd in ONE
// launch: the wave already holds alpha, T, w for the whole ray, so the loss gradient (which needs the ray sums beta_r, rgb_r) and
// the closed-form compositing backward run back to back in registers.  Writes what the MLP backward consumes
// (d_sigma, d_albedo, d_sun_v, g_beta per sample; d_sky per ray), the loss partial sums and the rendered colour (for logging).
__global__ void __launch_bounds__(256) render_loss_kernel(const float* __restrict__ z, const float* __restrict__ sigma,
                                                         const float* __restrict__ noise, float noise_std, const float* __restrict__ albedo,
                                                         const float* __restrict__ sun_v, const float* __restrict__ beta,
                                                         const float* __restrict__ sky, const float* __restrict__ target, long n_rays, int S,
                                                         float beta_min, const float* __restrict__ sched, float* __restrict__ loss_parts,
                                                         float* __restrict__ rgb_out, float* __restrict__ d_sigma, float* __restrict__ d_albedo,
                                                         float* __restrict__ d_sun, float* __restrict__ g_beta, float* __restrict__ d_sky) {
  __shared__ float part[4];
  const bool warm = sched != nullptr && sched[kSchedWarm] != 0.f;  // SNerfLoss epochs (main.py:128-131): plain MSE, no beta
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wv;
  float contrib = 0.f;
  if (r < n_rays) {
    const long i = r * S;
    // (the same per-ray function as the epilogue of the fused training forward, csrc/mlp_fwd_io.inc: bit-identical results)
    contrib = render_loss_ray(z + i, sigma + i, noise ? noise + i : nullptr, noise_std, albedo + i * 3, sun_v + i, beta + i, sky[r * 3], sky[r * 3 + 1],
                              sky[r * 3 + 2], target + r * 3, r, n_rays, S, lane, beta_min, warm, rgb_out ? rgb_out + r * 3 : nullptr, d_sigma + i,
                              d_albedo + i * 3, d_sun + i, g_beta + i, d_sky + r * 3);
  }
  if (lane == 0) part[wv] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// ---- batch gather from the GPU-resident ray bank (replaces DataLoader collate + H2D copy, main.py:96-110) -----------------
// out_rays[b] = rays[idx[b]] (11 floats), out_rgbs[b] = rgbs[idx[b]] (3), out_ts[b] = ts[idx[b]]; one thread per (ray, column)
__global__ void __launch_bounds__(256) gather_batch_kernel(const float* __restrict__ rays, const float* __restrict__ rgbs,
                                                          const long long* __restrict__ ts, const long long* __restrict__ idx, long n,
                                                          float* __restrict__ out_rays, float* __restrict__ out_rgbs,
                                                          long long* __restrict__ out_ts, float* __restrict__ cursor, unsigned batches) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long b = t >> 4;
  const int c = (int)(t & 15);
  long first = 0;
  if (cursor) {  // idx holds a whole epoch: this launch takes batch cursor[0] of it and moves the cursor on (captured steps)
    const uint32_t k = (uint32_t)cursor[0];
    tick_when_all_read(cursor, k, batches);
    first = (long)k * n;
  }
  if (b >= n) return;
  const long src = idx[first + b];
  if (c < 11) out_rays[b * 11 + c] = rays[src * 11 + c];
  else if (c < 14) out_rgbs[b * 3 + (c - 11)] = rgbs[src * 3 + (c - 11)];
  else if (c == 14) out_ts[b] = ts[src];
}

// gather + ray set-up in one launch, one wave per ray (captured steps that sample for themselves): the batch row goes to the static
// inputs (lanes 0..14), the stratified depths (jitter drawn in the kernel) and the sky colour are computed from the row just read --
// the same functions as sr_ray_setup_rng, so z and sky are bit-identical to gather followed by ray set-up.
__global__ void __launch_bounds__(256) gather_setup_kernel(const float* __restrict__ rays, const float* __restrict__ rgbs,
                                                          const long long* __restrict__ ts, const long long* __restrict__ idx, long n,
                                                          float* __restrict__ out_rays, float* __restrict__ out_rgbs,
                                                          long long* __restrict__ out_ts, float* __restrict__ cursor, unsigned batches, int S,
                                                          int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ z_out, float* __restrict__ sky, unsigned long long seed,
                                                          const float* __restrict__ step_counter, int step_offset) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  long first = 0;
  if (cursor) {
    const uint32_t k = (uint32_t)cursor[0];
    tick_when_all_read(cursor, k, batches);
    first = (long)k * n;
  }
  const uint32_t rng_step = (step_counter ? (uint32_t)step_counter[0] : 0u) + (uint32_t)step_offset;
  if (b >= n) return;
  const long src = idx[first + b];
  const float* ray = rays + src * 11;
  if (lane < 11) out_rays[b * 11 + lane] = ray[lane];
  else if (lane < 14) out_rgbs[b * 3 + (lane - 11)] = rgbs[src * 3 + (lane - 11)];
  else if (lane == 14) out_ts[b] = ts[src];
  const float near = ray[6], far = ray[7];
  for (int j = lane; j < S; j += 64) z_out[b * S + j] = stratified_z(near, far, j, S, philox_uniform(seed, b, j, rng_step));
  float k0, k1, k2;
  sky_ray(ray[8], ray[9], ray[10], hidden, w1, b1, w2, b2, lane, k0, k1, k2);
  if (lane == 0) sky[b * 3 + 0] = k0, sky[b * 3 + 1] = k1, sky[b * 3 + 2] = k2;
}

// torch.optim.Adam (main.py:84: lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight decay), one launch over the flat buffer: four elements per
// thread (16-byte loads / stores; the launcher checks the alignment), the last block's first n % 4 threads take the scalar tail.
// (adam_one: adam_device.h)
template <bool VEC>
__device__ __forceinline__ void adam_body(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                          float lr, float b1, float b2, float eps, float grad_scale, float bc1, float bc2, int zero_grad) {
  const float step_size = lr / bc1, sqrt_bc2 = sqrtf(bc2);  // once per thread; the arithmetic per element is torch.optim.Adam's
  if constexpr (VEC) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x, n4 = n >> 2;
    if (i4 < n4) {
      float4 pp = reinterpret_cast<float4*>(p)[i4], gg = reinterpret_cast<float4*>(g)[i4], mm = reinterpret_cast<float4*>(m)[i4],
             vv = reinterpret_cast<float4*>(v)[i4];
      adam_one(pp.x, gg.x, mm.x, vv.x, step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
      adam_one(pp.y, gg.y, mm.y, vv.y, step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
      adam_one(pp.z, gg.z, mm.z, vv.z, step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
      adam_one(pp.w, gg.w, mm.w, vv.w, step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
      reinterpret_cast<float4*>(p)[i4] = pp, reinterpret_cast<float4*>(m)[i4] = mm, reinterpret_cast<float4*>(v)[i4] = vv;
      if (zero_grad) reinterpret_cast<float4*>(g)[i4] = gg;
    }
    if (blockIdx.x == gridDim.x - 1 && (long)threadIdx.x < (n & 3)) {
      const long i = (n4 << 2) + threadIdx.x;
      adam_one(p[i], g[i], m[i], v[i], step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
    }
  } else {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) adam_one(p[i], g[i], m[i], v[i], step_size, b1, b2, eps, grad_scale, sqrt_bc2, zero_grad);
  }
}
// The 1-based step count arrives by value (the update is launched eagerly after the gradient all-reduce, outside the
// captured forward/backward graph); grad is optionally zeroed for the next step.
template <bool VEC>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float b1, float b2, float eps, float grad_scale, float bc1,
                                                  float bc2, int zero_grad) {
  adam_body<VEC>(p, g, m, v, n, lr, b1, b2, eps, grad_scale, bc1, bc2, zero_grad);
}

// Graph-capturable Adam: the 1-based step count is read from state[0]; it is advanced earlier in the same graph by the step's
// first kernel (sr_pack_all's `tick`), so no kernel both reads and writes it.  Bias corrections once per block.
template <bool VEC>
__global__ void __launch_bounds__(256) adam_graph_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                        float grad_scale, const float* __restrict__ state, int zero_grad) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const float t = state[kSchedStep];
    bc[0] = 1.0f - powf(b1, t), bc[1] = 1.0f - powf(b2, t);
  }
  __syncthreads();
  if (lr < 0.f) lr = state[kSchedLr];  // the scheduler's current rate (StepLR per epoch, main.py:86-94) under graph replay
  adam_body<VEC>(p, g, m, v, n, lr, b1, b2, eps, grad_scale, bc[0], bc[1], zero_grad);
}

}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_loss(const float* rgb, const float* weights, const float* beta, const float* target, int64_t n_rays, int n_samples,
                               float beta_min, float grad_scale, const float* sched, float* loss_parts, float* g_rgb, float* g_weights,
                               float* g_beta, void* stream) {
  SR_REQUIRE(rgb && weights && beta && target && loss_parts && g_rgb && g_weights && g_beta, "sr_satnerf_loss: null pointer");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(satnerf_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rgb, weights, beta, target,
                     (long)n_rays, n_samples, beta_min, grad_scale, sched, loss_parts, g_rgb, g_weights, g_beta);
  return check_launch("satnerf_loss_kernel");
}

// ---- solar correction (metrics.py:27-34; rendering.py:102-108): the second pass renders the same depths along the SUN direction;
// term2 = sum_j (T_j - sun_j)^2, term3 = 1 - sum_j w_j sun_j with T, w of that pass DETACHED -- only sun_sc carries a gradient.
// One wave per ray: compositing forward of the pass (alpha, T, w) -> the two terms -> d loss / d sun_j, nothing else written.
__global__ void __launch_bounds__(256) sc_loss_kernel(const float* __restrict__ z, const float* __restrict__ sigma,
                                                     const float* __restrict__ noise, float noise_std, const float* __restrict__ sun_v,
                                                     long n_rays, int S, float lam, float* __restrict__ loss_parts,
                                                     float* __restrict__ d_sun) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wv;
  const bool ray_on = r < n_rays;
  const float inv_n = 1.0f / (float)n_rays;
  float carry = 1.f, t2 = 0.f, t3 = 0.f;
  for (int j0 = 0; j0 < S && ray_on; j0 += 64) {
    const int j = j0 + lane;
    const bool on = j < S;
    const long i = r * S + (on ? j : S - 1);
    float alpha = 0.f, sv = 0.f;
    if (on) {
#pragma clang fp contract(off)
      const float zj = z[i];
      const float delta = j < S - 1 ? z[i + 1] - zj : 1e10f;
      float sg = sigma[i];
      if (noise) sg = sg + noise[i] * noise_std;
      alpha = 1.0f - expf(-delta * (sg > 0.f ? sg : 0.f));
      sv = sun_v[i];
    }
    float f;
    {
#pragma clang fp contract(off)
      f = on ? (1.0f - alpha) + 1e-10f : 1.f;
    }
    const float incl = wave_scan_mul(f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    const float T = carry * excl;
    carry = carry * __shfl(incl, 63, 64);
    if (on) {
      const float w = alpha * T, e = T - sv;
      t2 += e * e, t3 += w * sv;
      d_sun[i] = lam * inv_n * (-2.0f * e - w);
    }
  }
  t2 = wave_sum_f(t2), t3 = wave_sum_f(t3);
  if (lane == 0) part[wv] = ray_on ? lam * inv_n * (t2 + 1.0f - t3) : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int sr_sc_loss(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* sun_v, int64_t n_rays,
                          int n_samples, float lambda_sc, float* loss_parts, float* d_sun_v, void* stream) {
  if (n_rays <= 0) return 0;
  SR_REQUIRE(z_vals && sigma && sun_v && loss_parts && d_sun_v, "sr_sc_loss: null pointer");
  SR_REQUIRE(n_samples >= 1, "sr_sc_loss: n_samples must be >= 1");
  hipLaunchKernelGGL(sc_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z_vals, sigma, noise, noise_std, sun_v,
                     (long)n_rays, n_samples, lambda_sc / 3.0f, loss_parts, d_sun_v);
  return check_launch("sc_loss_kernel");
}

extern "C" int sr_depth_loss(const float* depth, const float* depths, int depths_stride, int use_weights, int64_t n_rays, float lambda_ds,
                             float* loss_parts, float* g_depth, void* stream) {
  if (n_rays <= 0) return 0;
  SR_REQUIRE(depth && depths && loss_parts && g_depth, "sr_depth_loss: null pointer");
  SR_REQUIRE(depths_stride >= (use_weights ? 2 : 1), "sr_depth_loss: depths_stride=%d too small", depths_stride);
  hipLaunchKernelGGL(depth_loss_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, depths, depths_stride,
                     use_weights, (long)n_rays, lambda_ds / 3.0f, loss_parts, g_depth);
  return check_launch("depth_loss_kernel");
}

extern "C" int sr_render_loss(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                              const float* sun_v, const float* beta, const float* sky, const float* target, int64_t n_rays, int n_samples,
                              float beta_min, const float* sched, float* loss_parts, float* rgb, float* d_sigma, float* d_albedo,
                              float* d_sun_v, float* g_beta, float* d_sky, void* stream) {
  SR_REQUIRE(z_vals && sigma && albedo && sun_v && beta && sky && target && loss_parts && d_sigma && d_albedo && d_sun_v && g_beta && d_sky,
             "sr_render_loss: null pointer");
  SR_REQUIRE(n_samples >= 1 && n_samples <= 64, "sr_render_loss: n_samples=%d unsupported (1..64); use the separate kernels", n_samples);
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(render_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z_vals, sigma, noise, noise_std,
                     albedo, sun_v, beta, sky, target, (long)n_rays, n_samples, beta_min, sched, loss_parts, rgb, d_sigma, d_albedo, d_sun_v,
                     g_beta, d_sky);
  return check_launch("render_loss_kernel");
}

extern "C" int sr_gather_batch(const float* rays, const float* rgbs, const int64_t* ts, const int64_t* idx, int64_t n, float* out_rays,
                               float* out_rgbs, int64_t* out_ts, float* cursor, int64_t batches, void* stream) {
  SR_REQUIRE(rays && rgbs && ts && idx && out_rays && out_rgbs && out_ts, "sr_gather_batch: null pointer");
  SR_REQUIRE(cursor == nullptr || (batches >= 1 && batches < (1 << 24)), "sr_gather_batch: a cursor needs 1 <= batches < 2^24");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, rgbs,
                     (const long long*)ts, (const long long*)idx, (long)n, out_rays, out_rgbs, (long long*)out_ts, cursor, (unsigned)batches);
  return check_launch("gather_batch_kernel");
}

extern "C" int sr_gather_setup(const float* rays, const float* rgbs, const int64_t* ts, const int64_t* idx, int64_t n, float* out_rays,
                               float* out_rgbs, int64_t* out_ts, float* cursor, int64_t batches, int n_samples, int hidden, const float* w1,
                               const float* b1, const float* w2, const float* b2, float* z_vals, float* sky, uint64_t seed,
                               const float* step_counter, int step_offset, void* stream) {
  SR_REQUIRE(rays && rgbs && ts && idx && out_rays && out_rgbs && out_ts && w1 && b1 && w2 && b2 && z_vals && sky, "sr_gather_setup: null pointer");
  SR_REQUIRE(cursor == nullptr || (batches >= 1 && batches < (1 << 24)), "sr_gather_setup: a cursor needs 1 <= batches < 2^24");
  SR_REQUIRE(n_samples >= 2, "sr_gather_setup: n_samples >= 2 required");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_setup_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rays, rgbs, (const long long*)ts,
                     (const long long*)idx, (long)n, out_rays, out_rgbs, (long long*)out_ts, cursor, (unsigned)batches, n_samples, hidden, w1, b1,
                     w2, b2, z_vals, sky, (unsigned long long)seed, step_counter, step_offset);
  return check_launch("gather_setup_kernel");
}

static bool adam_vectorisable(const float* a, const float* b, const float* c, const float* d) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15u) == 0;
}
static unsigned adam_vec_blocks(int64_t n) {  // one thread per four elements; at least one block (it takes the scalar tail)
  const int64_t n4 = n >> 2;
  return (unsigned)(n4 > 0 ? (n4 + 255) / 256 : 1);
}

extern "C" int sr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                            float eps, float grad_scale, int64_t step, int zero_grad, void* stream) {
  SR_REQUIRE(params && grads && exp_avg && exp_avg_sq, "sr_adam_step: null pointer");
  SR_REQUIRE(step >= 1, "sr_adam_step: step is 1-based");
  if (n <= 0) return 0;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  if (adam_vectorisable(params, grads, exp_avg, exp_avg_sq))
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_vec_blocks(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (long)n, lr,
                       beta1, beta2, eps, grad_scale, bc1, bc2, zero_grad);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (long)n, lr, beta1, beta2, eps, grad_scale, bc1, bc2, zero_grad);
  return check_launch("adam_kernel");
}

extern "C" int sr_adam_step_graph(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                  float eps, float grad_scale, float* state, int zero_grad, void* stream) {
  SR_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, "sr_adam_step_graph: null pointer");
  if (n <= 0) return 0;
  if (adam_vectorisable(params, grads, exp_avg, exp_avg_sq))
    hipLaunchKernelGGL(adam_graph_kernel<true>, dim3(adam_vec_blocks(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (long)n, lr, beta1, beta2, eps, grad_scale, state, zero_grad);
  else
    hipLaunchKernelGGL(adam_graph_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, (long)n, lr, beta1, beta2, eps, grad_scale, state, zero_grad);
  return check_launch("adam_graph_kernel");
}
