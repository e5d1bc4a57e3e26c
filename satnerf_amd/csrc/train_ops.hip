// Training-step kernels around the renderer: the Sat-NeRF loss with its gradient in one launch,
// and Adam over the flat parameter buffer in one launch.  Both are HBM-bound streaming kernels over <= 3 MB.
#include "common.h"

namespace sr {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// metrics.SatNerfLoss for the coarse model (metrics.py:21-25,56-73), forward value AND the gradients it sends back:
//   beta_r = sum_j w_rj b_rj + beta_min
//   loss   = mean_{r,c} (rgb_rc - gt_rc)^2 / (2 beta_r^2)  +  (3 + mean_r log beta_r) / 2
// One wave per ray.  loss_out[0] must be zero on entry; g_* are the gradients w.r.t. rgb (N,3), weights (N,S), beta (N,S).
__global__ void __launch_bounds__(256) satnerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                          const float* __restrict__ beta, const float* __restrict__ target, long n_rays, int S,
                                                          float beta_min, float grad_scale, float* __restrict__ loss_out,
                                                          float* __restrict__ g_rgb, float* __restrict__ g_weights, float* __restrict__ g_beta) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  float acc = 0.f;
  for (int j = lane; j < S; j += 64) acc += weights[r * S + j] * beta[r * S + j];
  const float b = wave_sum_f(acc) + beta_min;
  const float inv_n = 1.0f / (float)n_rays;
  const float d0 = rgb[r * 3] - target[r * 3], d1 = rgb[r * 3 + 1] - target[r * 3 + 1], d2 = rgb[r * 3 + 2] - target[r * 3 + 2];
  const float sq = d0 * d0 + d1 * d1 + d2 * d2;
  const float ib2 = 1.0f / (b * b);
  if (lane == 0) {
    float contrib = sq * ib2 * (0.5f / 3.0f) * inv_n + 0.5f * logf(b) * inv_n;
    if (r == 0) contrib += 1.5f;
    atomicAdd(loss_out, contrib);
    const float k = ib2 * (1.0f / 3.0f) * inv_n * grad_scale;
    g_rgb[r * 3] = d0 * k, g_rgb[r * 3 + 1] = d1 * k, g_rgb[r * 3 + 2] = d2 * k;
  }
  // d loss / d beta_r = -sq / (3 N beta^3) + 1 / (2 N beta)
  const float db = (-sq * ib2 / b * (1.0f / 3.0f) + 0.5f / b) * inv_n * grad_scale;
  for (int j = lane; j < S; j += 64) {
    g_weights[r * S + j] = db * beta[r * S + j];
    g_beta[r * S + j] = db * weights[r * S + j];
  }
}

// torch.optim.Adam (main.py:84: lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight decay), one launch over the flat buffer.
// step_count lives on the device so the launch can be replayed from a hipGraph; grad is optionally zeroed for the next step.
__global__ void adam_tick_kernel(float* step_count) { step_count[0] += 1.0f; }

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float b1, float b2, float eps, float grad_scale,
                                                  const float* __restrict__ step_count, int zero_grad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float t = step_count[0];
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float gi = g[i] * grad_scale;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi, v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
  if (zero_grad) g[i] = 0.f;
}

}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_loss(const float* rgb, const float* weights, const float* beta, const float* target, int64_t n_rays, int n_samples,
                               float beta_min, float grad_scale, float* loss_out, float* g_rgb, float* g_weights, float* g_beta,
                               void* stream) {
  SR_REQUIRE(rgb && weights && beta && target && loss_out && g_rgb && g_weights && g_beta, "sr_satnerf_loss: null pointer");
  if (n_rays <= 0) return 0;
  if (hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) {
    set_error("sr_satnerf_loss: hipMemsetAsync failed");
    return 1;
  }
  hipLaunchKernelGGL(satnerf_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rgb, weights, beta, target,
                     (long)n_rays, n_samples, beta_min, grad_scale, loss_out, g_rgb, g_weights, g_beta);
  return check_launch("satnerf_loss_kernel");
}

extern "C" int sr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                            float eps, float grad_scale, float* step_count, int zero_grad, void* stream) {
  SR_REQUIRE(params && grads && exp_avg && exp_avg_sq && step_count, "sr_adam_step: null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_count);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     (long)n, lr, beta1, beta2, eps, grad_scale, step_count, zero_grad);
  return check_launch("adam_kernel");
}
