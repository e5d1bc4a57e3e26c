// Per-ray device functions shared by the stand-alone ray kernels (ray_ops.hip) and the fused render kernel (mlp_fwd.inc):
// stratified sampling, the in-kernel jitter RNG, the sky head and sigma->alpha compositing, one wavefront per ray (lane = sample).
// The fused kernel calls exactly these functions on exactly the same fp32 values, so fused and unfused renders are bit-identical.
#pragma once
#include "common.h"

namespace sr {

// ---- wave primitives (64 lanes) ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// inclusive scans across the wave
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v *= o;
  }
  return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_rscan_add(float v, int lane) {  // inclusive suffix sum
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(v, d, 64);
    if (lane + d < 64) v += o;
  }
  return v;
}

// ---- stratified sampling: rendering.py:62-78 ---------------------------------------------------------------
// torch.linspace(0,1,S) in fp32: step = 1/(S-1); i < S/2 ? i*step : 1 - step*(S-1-i), the latter FUSED (one
// rounding) -- ATen RangeFactories as compiled; checked bit-for-bit against torch.linspace for S in {2,7,50,64,128}.
__device__ __forceinline__ float linspace01(int i, int n, float step) {
  return i < n / 2 ? (float)i * step : __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}
__device__ __forceinline__ float lerp_near_far(float near, float far, float s) {
#pragma clang fp contract(off)
  const float a = near * (1.0f - s), b = far * s;  // rendering.py:67, this exact form
  return a + b;
}
// depth of sample j of S: z_j = lower + (upper - lower) * u, lower / upper = midpoints to the neighbours (rendering.py:70-78)
__device__ __forceinline__ float stratified_z(float near, float far, int j, int S, float u) {
#pragma clang fp contract(off)
  const float step = 1.0f / (float)(S - 1);
  const float zj = lerp_near_far(near, far, linspace01(j, S, step));
  float lower = zj, upper = zj;
  if (j > 0) lower = 0.5f * (lerp_near_far(near, far, linspace01(j - 1, S, step)) + zj);
  if (j < S - 1) upper = 0.5f * (zj + lerp_near_far(near, far, linspace01(j + 1, S, step)));
  const float span = upper - lower;
  const float jit = span * u;
  return lower + jit;
}

// Philox-4x32-10 (Salmon et al., SC'11): counter-based, so a captured launch draws fresh jitter on every replay from a
// device-side step counter -- no RNG kernel, no generator-state bookkeeping in the graph.
__device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0, c[1] = (uint32_t)p1, c[2] = n2, c[3] = (uint32_t)p0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
}
// uniform in [0,1) with 24 random bits for sample j of ray r at step `step`: counter = (ray, sample / 4, step), key = seed
__device__ __forceinline__ float philox_uniform(unsigned long long seed, long r, int j, uint32_t step) {
  uint32_t c[4] = {static_cast<uint32_t>(r), static_cast<uint32_t>(static_cast<unsigned long long>(r) >> 32), static_cast<uint32_t>(j >> 2), step};
  philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c);
  return (float)(c[j & 3] >> 8) * 5.9604644775390625e-8f;
}

// A launch that draws from the device-side step counter can also advance it (forward graphs have no sr_pack_all to do so): every
// workgroup reads counter[0] first, then checks in at the arrival counter kept in counter[3] (uint32 bits); the LAST one to
// check in -- every other workgroup has read by then -- stores step + 1 and resets the arrival counter for the next launch.
// Contains a workgroup barrier: call it from uniform control flow.
// modulo > 0: the counter wraps (a cursor over the batches of an epoch).
// The counter is a float (the block also carries the learning rate): integers are exact up to 2^24 only, where += 1 would stop
// advancing (jitter, bank chunk and Adam step would freeze after ~25 minutes of 90-us replays).  next_step() wraps into
// [2^23, 2^24) instead, keeping the value's residue modulo `keep_mod` (the bank's chunk count): chunk order is unbroken, the
// jitter stream repeats after 2^23 steps, and Adam's bias corrections are 1 to fp32 precision from ~10^4 steps on.
__device__ __forceinline__ uint32_t next_step(uint32_t step, uint32_t keep_mod = 1u) {
  uint32_t next = step + 1u;
  if (next >= (1u << 24)) next = (1u << 23) + (next - (1u << 23)) % (keep_mod ? keep_mod : 1u);
  return next;
}
__device__ __forceinline__ void tick_when_all_read(float* counter, uint32_t step_read, uint32_t modulo = 0u, uint32_t keep_mod = 1u) {
  asm volatile("" ::"v"(step_read));  // the value has arrived
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* arrive = reinterpret_cast<unsigned*>(counter + 3);
    __threadfence();
    if (atomicAdd(arrive, 1u) == gridDim.x - 1) {
      counter[0] = (float)(modulo != 0u ? (step_read + 1u >= modulo ? 0u : step_read + 1u) : next_step(step_read, keep_mod));
      atomicExch(arrive, 0u);
    }
  }
}
// The same at the END of a launch (training launches, r05): ONE thread per workgroup calls it after every wave of the workgroup has long used
// the value read at the start -- no workgroup barrier and no fence (nothing but the counter is published, and it is read by the NEXT
// launch).  At the start of the launch the returning atomic held wave 0 -- and with it the workgroup's next barrier -- for a memory round
// trip per counter (the forward launch that also samples its batch ticks two: +3 us on a 92-us kernel).
__device__ __forceinline__ void tick_at_exit(float* counter, uint32_t step_read, uint32_t modulo = 0u, uint32_t keep_mod = 1u) {
  unsigned* arrive = reinterpret_cast<unsigned*>(counter + 3);
  if (atomicAdd(arrive, 1u) == gridDim.x - 1) {
    counter[0] = (float)(modulo != 0u ? (step_read + 1u >= modulo ? 0u : step_read + 1u) : next_step(step_read, keep_mod));
    atomicExch(arrive, 0u);
  }
}

// ---- sky colour head of one ray, computed by one wave: models/satnerf.py:138-143,201; every lane returns the colour -------
__device__ __forceinline__ void sky_ray(float sx, float sy, float sz, int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                        const float* __restrict__ w2, const float* __restrict__ b2, int lane, float& k0, float& k1, float& k2) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int k = lane; k < hidden; k += 64) {
    float hk = __builtin_fmaf(w1[k * 3 + 2], sz, __builtin_fmaf(w1[k * 3 + 1], sy, __builtin_fmaf(w1[k * 3], sx, b1[k])));
    hk = hk > 0.f ? hk : 0.f;
    a0 = __builtin_fmaf(w2[k], hk, a0);
    a1 = __builtin_fmaf(w2[hidden + k], hk, a1);
    a2 = __builtin_fmaf(w2[2 * hidden + k], hk, a2);
  }
  a0 = wave_sum(a0), a1 = wave_sum(a1), a2 = wave_sum(a2);
  k0 = sigmoid_f(a0 + b2[0]), k1 = sigmoid_f(a1 + b2[1]), k2 = sigmoid_f(a2 + b2[2]);
}

// ---- compositing: models/satnerf.py:52-70 --------------------------------------------------------------------
// One wave per ray; samples are processed in segments of 64 (lane = sample) with a running transmittance carry.
// (z, sigma, noise: the ray's own rows.)
__device__ __forceinline__ void alpha_at(const float* z, const float* sigma, const float* noise, float noise_std, int j, int S, float& delta,
                                         float& dens, float& alpha) {
#pragma clang fp contract(off)
  delta = (j < S - 1) ? (z[j + 1] - z[j]) : 1e10f;
  float s = sigma[j];
  if (noise) s = s + noise[j] * noise_std;
  dens = s;
  const float rl = s > 0.f ? s : 0.f;
  alpha = 1.0f - expf(-delta * rl);
}
__device__ __forceinline__ void alpha_of(const float* z, const float* sigma, const float* noise, float noise_std, long base, int j, int S,
                                         float& delta, float& dens, float& alpha) {
  alpha_at(z + base, sigma + base, noise ? noise + base : nullptr, noise_std, j, S, delta, dens, alpha);
}

// weights / transparency / depth / rgb of ONE ray by one wave.  Inputs are the ray's own rows (global or LDS); albedo may be
// NULL (sigma-only rendering), sun_v NULL = irradiance 1; outputs weights / transp are the ray's rows, depth_r / rgb_r may be NULL.
__device__ __forceinline__ void composite_ray(const float* z, const float* sigma, const float* noise, float noise_std, const float* albedo,
                                              const float* sun_v, float k0, float k1, float k2, int S, int lane, int clamp_rgb,
                                              float* __restrict__ weights, float* __restrict__ transp, float* __restrict__ depth_r,
                                              float* __restrict__ rgb_r) {
  float carry = 1.f, dsum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const bool on = j < S;
    float delta, dens, alpha = 0.f;
    if (on) alpha_at(z, sigma, noise, noise_std, j, S, delta, dens, alpha);
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
      const float w = alpha * T;
      weights[j] = w;
      transp[j] = T;
      dsum += w * z[j];
      if (albedo) {
        const float* a = albedo + j * 3;
        float i0 = 1.f, i1 = 1.f, i2 = 1.f;
        if (sun_v) {
          const float sv = sun_v[j];
          i0 = sv + (1.f - sv) * k0, i1 = sv + (1.f - sv) * k1, i2 = sv + (1.f - sv) * k2;  // :68
        }
        c0 += w * a[0] * i0, c1 += w * a[1] * i1, c2 += w * a[2] * i2;
      }
    }
  }
  dsum = wave_sum(dsum), c0 = wave_sum(c0), c1 = wave_sum(c1), c2 = wave_sum(c2);
  if (lane == 0) {
    if (depth_r) *depth_r = dsum;
    if (rgb_r) {
      if (clamp_rgb) c0 = fminf(fmaxf(c0, 0.f), 1.f), c1 = fminf(fmaxf(c1, 0.f), 1.f), c2 = fminf(fmaxf(c2, 0.f), 1.f);
      rgb_r[0] = c0, rgb_r[1] = c1, rgb_r[2] = c2;
    }
  }
}

// ---- compositing forward -> SatNerf / SNerf colour loss -> compositing backward of ONE ray by one wave (lane = sample, S <= 64) ----------
// models/satnerf.py:52-70 + metrics.py:21-25,36-44,56-73 + their autograd (closed form).  Inputs are the ray's own rows
// (global memory or LDS); writes what the MLP backward consumes (d_sigma, d_albedo (S,3), d_sun, g_beta per sample; d_sky (3)), optionally the
// rendered colour, and returns (every lane) the ray's share of the batch-mean loss.  `warm` = the SNerfLoss epochs (main.py:128-131: plain
// MSE, no uncertainty term).  Shared by sr_render_loss and the epilogue of the fused training forward: the two are bit-identical.
__device__ __forceinline__ float render_loss_ray(const float* z, const float* sigma, const float* noise, float noise_std, const float* albedo,
                                                 const float* sun_v, const float* beta, float k0, float k1, float k2, const float* target, long r,
                                                 long n_rays, int S, int lane, float beta_min, bool warm, float* __restrict__ rgb_out,
                                                 float* __restrict__ d_sigma, float* __restrict__ d_albedo, float* __restrict__ d_sun,
                                                 float* __restrict__ g_beta, float* __restrict__ d_sky) {
  const bool on = lane < S;
  const int i = lane < S ? lane : S - 1;
  // forward: alpha, transmittance, weights (models/satnerf.py:52-63)
  float delta = 0.f, dens = 0.f, alpha = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, sv = 0.f, bj = 0.f;
  if (on) {
#pragma clang fp contract(off)
    const float zj = z[i];
    delta = lane < S - 1 ? z[i + 1] - zj : 1e10f;
    float s = sigma[i];
    if (noise) s = s + noise[i] * noise_std;
    dens = s;
    alpha = 1.0f - expf(-delta * (s > 0.f ? s : 0.f));
    a0 = albedo[i * 3], a1 = albedo[i * 3 + 1], a2 = albedo[i * 3 + 2];
    sv = sun_v[i], bj = beta[i];
  }
  float f;
  {
#pragma clang fp contract(off)
    f = on ? (1.0f - alpha) + 1e-10f : 1.f;
  }
  const float incl = wave_scan_mul(f, lane);
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  const float w = on ? alpha * T : 0.f;
  const float i0 = sv + (1.f - sv) * k0, i1 = sv + (1.f - sv) * k1, i2 = sv + (1.f - sv) * k2;  // irradiance, :68
  const float c0 = wave_sum(w * a0 * i0), c1 = wave_sum(w * a1 * i1), c2 = wave_sum(w * a2 * i2);
  const float b = wave_sum(w * bj) + beta_min;
  const float r0 = fminf(fmaxf(c0, 0.f), 1.f), r1 = fminf(fmaxf(c1, 0.f), 1.f), r2 = fminf(fmaxf(c2, 0.f), 1.f);
  // loss (metrics.py:21-25) and its gradient w.r.t. rgb and beta_r
  const float inv_n = 1.0f / (float)n_rays;
  const float e0 = r0 - target[0], e1 = r1 - target[1], e2 = r2 - target[2];
  const float sq = e0 * e0 + e1 * e1 + e2 * e2;
  const float ib2 = warm ? 2.0f : 1.0f / (b * b);  // metrics.SNerfLoss colour term = the same expression with beta^2 = 1/2, no log term
  float contrib = sq * ib2 * (0.5f / 3.0f) * inv_n + (warm ? 0.f : 0.5f * logf(b) * inv_n);
  if (r == 0 && !warm) contrib += 1.5f;
  if (lane == 0 && rgb_out) rgb_out[0] = r0, rgb_out[1] = r1, rgb_out[2] = r2;
  const float kk = ib2 * (1.0f / 3.0f) * inv_n;
  const float gr0 = (c0 >= 0.f && c0 <= 1.f) ? e0 * kk : 0.f;  // torch.clamp passes the gradient where min <= x <= max
  const float gr1 = (c1 >= 0.f && c1 <= 1.f) ? e1 * kk : 0.f;
  const float gr2 = (c2 >= 0.f && c2 <= 1.f) ? e2 * kk : 0.f;
  const float db = warm ? 0.f : (-sq * ib2 / b * (1.0f / 3.0f) + 0.5f / b) * inv_n;  // d loss / d beta_r
  // backward through compositing: G_j = dL/dw_j
  const float G = db * bj + gr0 * a0 * i0 + gr1 * a1 * i1 + gr2 * a2 * i2;
  const float tail = on ? G * w : 0.f;
  const float suf = wave_rscan_add(tail, lane);
  const float after = suf - tail;
  const float q0 = w * gr0, q1 = w * gr1, q2 = w * gr2;
  const float di0 = q0 * a0, di1 = q1 * a1, di2 = q2 * a2;
  const float s0 = wave_sum(di0 * (1.f - sv)), s1 = wave_sum(di1 * (1.f - sv)), s2 = wave_sum(di2 * (1.f - sv));
  if (lane == 0) d_sky[0] = s0, d_sky[1] = s1, d_sky[2] = s2;
  if (on) {
    const float dalpha = G * T - after / f;
    d_sigma[i] = dens > 0.f ? dalpha * delta * expf(-delta * dens) : 0.f;
    d_albedo[i * 3] = q0 * i0, d_albedo[i * 3 + 1] = q1 * i1, d_albedo[i * 3 + 2] = q2 * i2;
    d_sun[i] = di0 * (1.f - k0) + di1 * (1.f - k1) + di2 * (1.f - k2);
    g_beta[i] = db * w;
  }
  return contrib;
}

}  // namespace sr
