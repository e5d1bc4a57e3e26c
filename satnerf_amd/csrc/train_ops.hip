// Training-step kernels around the renderer: the Sat-NeRF loss with its gradient in one launch,
// and Adam over the flat parameter buffer in one launch.  Both are HBM-bound streaming kernels over <= 3 MB.
#include <math.h>

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
// One wave per ray, 4 rays per block.  Block b writes its share of the loss to loss_parts[b] (no atomics, no memset; the
// value is sum(loss_parts) and only logging reads it); g_* are the gradients w.r.t. rgb (N,3), weights (N,S), beta (N,S).
__global__ void __launch_bounds__(256) satnerf_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                          const float* __restrict__ beta, const float* __restrict__ target, long n_rays, int S,
                                                          float beta_min, float grad_scale, float* __restrict__ loss_parts,
                                                          float* __restrict__ g_rgb, float* __restrict__ g_weights, float* __restrict__ g_beta) {
  __shared__ float part[4];
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
  const float ib2 = 1.0f / (b * b);
  if (lane == 0 && on) {
    float contrib = sq * ib2 * (0.5f / 3.0f) * inv_n + 0.5f * logf(b) * inv_n;
    if (r == 0) contrib += 1.5f;
    part[wv] = contrib;
    const float k = ib2 * (1.0f / 3.0f) * inv_n * grad_scale;
    g_rgb[r * 3] = d0 * k, g_rgb[r * 3 + 1] = d1 * k, g_rgb[r * 3 + 2] = d2 * k;
  }
  __syncthreads();
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
  if (!on) return;
  // d loss / d beta_r = -sq / (3 N beta^3) + 1 / (2 N beta)
  const float db = (-sq * ib2 / b * (1.0f / 3.0f) + 0.5f / b) * inv_n * grad_scale;
  for (int j = lane; j < S; j += 64) {
    g_weights[r * S + j] = db * beta[r * S + j];
    g_beta[r * S + j] = db * weights[r * S + j];
  }
}

// torch.optim.Adam (main.py:84: lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight decay), one launch over the flat buffer.
// The 1-based step count arrives by value (the update is launched eagerly after the gradient all-reduce, outside the
// captured forward/backward graph); grad is optionally zeroed for the next step.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float b1, float b2, float eps, float grad_scale, float bc1,
                                                  float bc2, int zero_grad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
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
                               float beta_min, float grad_scale, float* loss_parts, float* g_rgb, float* g_weights, float* g_beta,
                               void* stream) {
  SR_REQUIRE(rgb && weights && beta && target && loss_parts && g_rgb && g_weights && g_beta, "sr_satnerf_loss: null pointer");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(satnerf_loss_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rgb, weights, beta, target,
                     (long)n_rays, n_samples, beta_min, grad_scale, loss_parts, g_rgb, g_weights, g_beta);
  return check_launch("satnerf_loss_kernel");
}

extern "C" int sr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                            float eps, float grad_scale, int64_t step, int zero_grad, void* stream) {
  SR_REQUIRE(params && grads && exp_avg && exp_avg_sq, "sr_adam_step: null pointer");
  SR_REQUIRE(step >= 1, "sr_adam_step: step is 1-based");
  if (n <= 0) return 0;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     (long)n, lr, beta1, beta2, eps, grad_scale, bc1, bc2, zero_grad);
  return check_launch("adam_kernel");
}
