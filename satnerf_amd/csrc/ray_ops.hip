// Per-ray kernels of the Sat-NeRF rendering path for gfx950: stratified sampling, the per-ray sky head,
// sigma->alpha compositing (forward and closed-form backward), importance resampling + merge, and the
// weight-stream pack/unpack helpers.  All are HBM-bound and tiny next to the fused MLP; they are written
// one wavefront (64 lanes) per ray with lane = sample, wave scans/reductions through DPP shuffles, and
// coalesced row accesses.
#include <stdarg.h>

#include "common.h"
#include "mlp_layout.h"
#include <mutex>

#include "adam_device.h"
#include "ray_device.h"

namespace sr {

// ---- error plumbing -------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// one hipFuncSetAttribute per (kernel, device) of the process: several devices in one process (tests, notebooks) each need their own
// call, which a function-local `static bool` would skip.  The table is tiny and only ever grows; a mutex keeps it sane.
bool ensure_dynamic_lds(const void* kernel, size_t bytes) {
  struct Ent { const void* k; int dev; size_t bytes; };
  static Ent done[256];
  static int n_done = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n_done; ++i)
    if (done[i].k == kernel && done[i].dev == dev && done[i].bytes >= bytes) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed on device %d", bytes, dev);
    return false;
  }
  if (n_done < 256) done[n_done++] = Ent{kernel, dev, bytes};
  return true;
}

constexpr int kRaysPerBlock = 4;  // 4 waves = 256 threads

// ---- stratified sampling: rendering.py:62-78 (ray_device.h stratified_z) ---------------------------------
__global__ void __launch_bounds__(256) ray_sample_kernel(const float* __restrict__ rays, int ray_stride, const float* __restrict__ u,
                                                        long n_rays, int S, float* __restrict__ z_out) {
#pragma clang fp contract(off)
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_rays * S) return;
  const long r = idx / S;
  const int j = (int)(idx - r * S);
  z_out[idx] = stratified_z(rays[r * ray_stride + 6], rays[r * ray_stride + 7], j, S, u[idx]);
}

// ---- sky colour head, one wave per ray: models/satnerf.py:138-143,201 --------------------------------------
__global__ void __launch_bounds__(256) sky_kernel(const float* __restrict__ sun, int sun_stride, long n, int hidden,
                                                 const float* __restrict__ w1, const float* __restrict__ b1,
                                                 const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ sky) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= n) return;
  float k0, k1, k2;
  sky_ray(sun[r * sun_stride], sun[r * sun_stride + 1], sun[r * sun_stride + 2], hidden, w1, b1, w2, b2, lane, k0, k1, k2);
  if (lane == 0) sky[r * 3 + 0] = k0, sky[r * 3 + 1] = k1, sky[r * 3 + 2] = k2;
}

// ---- sampling + sky head in one launch (training fast path), one wave per ray ---------------------------------------------------
__global__ void __launch_bounds__(256) ray_setup_kernel(const float* __restrict__ rays, int ray_stride, const float* __restrict__ u, long n_rays,
                                                       int S, int hidden, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ z_out,
                                                       float* __restrict__ sky, unsigned long long seed, float* __restrict__ step_counter, int tick) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  const uint32_t rng_step = step_counter ? (uint32_t)step_counter[0] : 0u;
  if (tick) tick_when_all_read(step_counter, rng_step);
  if (r >= n_rays) return;
  const float* ray = rays + r * ray_stride;
  const float near = ray[6], far = ray[7];
  for (int j = lane; j < S; j += 64)
    z_out[r * S + j] = stratified_z(near, far, j, S, u ? u[r * S + j] : philox_uniform(seed, r, j, rng_step));
  float k0, k1, k2;
  sky_ray(ray[8], ray[9], ray[10], hidden, w1, b1, w2, b2, lane, k0, k1, k2);
  if (lane == 0) sky[r * 3 + 0] = k0, sky[r * 3 + 1] = k1, sky[r * 3 + 2] = k2;
}

// ---- compositing: models/satnerf.py:52-70 --------------------------------------------------------------------
__global__ void __launch_bounds__(256) composite_fwd_kernel(const float* __restrict__ z, const float* __restrict__ sigma,
                                                           const float* __restrict__ noise, float noise_std,
                                                           const float* __restrict__ albedo, const float* __restrict__ sun_v,
                                                           const float* __restrict__ sky, long n_rays, int S, int clamp_rgb,
                                                           float* __restrict__ weights, float* __restrict__ transp,
                                                           float* __restrict__ depth, float* __restrict__ rgb) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const long base = r * S;
  float k0 = 1.f, k1 = 1.f, k2 = 1.f;
  if (sky) k0 = sky[r * 3], k1 = sky[r * 3 + 1], k2 = sky[r * 3 + 2];
  composite_ray(z + base, sigma + base, noise ? noise + base : nullptr, noise_std, albedo ? albedo + base * 3 : nullptr,
                sun_v ? sun_v + base : nullptr, k0, k1, k2, S, lane, clamp_rgb, weights + base, transp + base, depth ? depth + r : nullptr,
                rgb ? rgb + r * 3 : nullptr);
}

// Whole-image evaluation: the same compositing, keeping per ray only what
// eval_satnerf.save_nerf_output_to_images writes per pixel (eval_satnerf.py:106-146) instead of 2,576 B of per-sample outputs:
// image[r] = { rgb(3) clamped, depth, acc = sum w, sum w*sun, sum w*albedo (3), sum w*beta, sum w*sky (3) }.
constexpr int kImageFloats = 13;
__global__ void __launch_bounds__(256) composite_image_kernel(const float* __restrict__ z, const float* __restrict__ sigma,
                                                             const float* __restrict__ noise, float noise_std,
                                                             const float* __restrict__ albedo, const float* __restrict__ sun_v,
                                                             const float* __restrict__ beta, const float* __restrict__ sky, long n_rays,
                                                             int S, float* __restrict__ image) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const long base = r * S;
  const float k0 = sky[r * 3], k1 = sky[r * 3 + 1], k2 = sky[r * 3 + 2];
  float carry = 1.f, v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // rgb(3) depth acc sun albedo(3); beta apart
  float bsum = 0.f;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const bool on = j < S;
    float delta, dens, alpha = 0.f;
    if (on) alpha_of(z, sigma, noise, noise_std, base, j, S, delta, dens, alpha);
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
      const float* a = albedo + (base + j) * 3;
      const float sv = sun_v[base + j];
      v[0] += w * a[0] * (sv + (1.f - sv) * k0), v[1] += w * a[1] * (sv + (1.f - sv) * k1), v[2] += w * a[2] * (sv + (1.f - sv) * k2);
      v[3] += w * z[base + j], v[4] += w, v[5] += w * sv;
      v[6] += w * a[0], v[7] += w * a[1], v[8] += w * a[2];
      bsum += w * beta[base + j];
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = wave_sum(v[i]);
  bsum = wave_sum(bsum);
  if (lane == 0) {
    float* o = image + r * kImageFloats;
    o[0] = fminf(fmaxf(v[0], 0.f), 1.f), o[1] = fminf(fmaxf(v[1], 0.f), 1.f), o[2] = fminf(fmaxf(v[2], 0.f), 1.f);
    o[3] = v[3], o[4] = v[4], o[5] = v[5], o[6] = v[6], o[7] = v[7], o[8] = v[8], o[9] = bsum;
    o[10] = v[4] * k0, o[11] = v[4] * k1, o[12] = v[4] * k2;  // sky is constant along the ray
  }
}

// depth -> scene point -> ECEF -> geodetic, in fp64 (datasets/satellite.py:246-275 + sat_utils.py:76-95): one thread per ray.
__global__ void __launch_bounds__(256) latlonalt_kernel(const float* __restrict__ rays, int ray_stride, const float* __restrict__ depth,
                                                       long n, double cx, double cy, double cz, double range, double* __restrict__ lat,
                                                       double* __restrict__ lon, double* __restrict__ alt) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + i * ray_stride;
  const double d = (double)depth[i];
  const double x = ((double)r[0] + (double)r[3] * d) * range + cx;
  const double y = ((double)r[1] + (double)r[4] * d) * range + cy;
  const double zz = ((double)r[2] + (double)r[5] * d) * range + cz;
  const double a = 6378137.0, e = 8.1819190842622e-2;
  const double asq = a * a, esq = e * e;
  const double b = sqrt(asq * (1 - esq));
  const double bsq = b * b;
  const double ep = sqrt((asq - bsq) / bsq);
  const double p = sqrt(x * x + y * y);
  const double th = atan2(a * zz, b * p);
  const double lo = atan2(y, x);
  const double sth = sin(th), cth = cos(th);
  const double la = atan2(zz + (ep * ep) * b * (sth * sth * sth), p - esq * a * (cth * cth * cth));
  const double sla = sin(la);
  const double N = a / sqrt(1 - esq * (sla * sla));
  alt[i] = p / cos(la) - N;
  lon[i] = lo * 180 / 3.141592653589793;
  lat[i] = la * 180 / 3.141592653589793;
}

// Closed-form backward (extended with a transparency gradient):
//   G_j   = g_w_j + g_depth z_j + sum_c ghat_c albedo_jc irr_jc
//   dalpha_j = G_j T_j - ( sum_{k>j} (G_k w_k + gT_k T_k) ) / (1 - alpha_j + 1e-10)
//   dsigma_j = dalpha_j * delta_j * exp(-delta_j relu(s_j)) * [s_j > 0]
__global__ void __launch_bounds__(256) composite_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ sigma, const float* __restrict__ noise, float noise_std,
    const float* __restrict__ albedo, const float* __restrict__ sun_v, const float* __restrict__ sky,
    const float* __restrict__ weights, const float* __restrict__ transp, long n_rays, int S, int clamp_rgb,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_w, const float* __restrict__ g_T,
    float* __restrict__ d_sigma, float* __restrict__ d_albedo, float* __restrict__ d_sun, float* __restrict__ d_sky) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const long base = r * S;
  float k0 = 1.f, k1 = 1.f, k2 = 1.f;
  if (sky) k0 = sky[r * 3], k1 = sky[r * 3 + 1], k2 = sky[r * 3 + 2];
  float gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
  if (g_rgb && albedo) {
    gr0 = g_rgb[r * 3], gr1 = g_rgb[r * 3 + 1], gr2 = g_rgb[r * 3 + 2];
    if (clamp_rgb) {  // torch.clamp passes the gradient where min <= x <= max: recompute the unclamped colour
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
      for (int j = lane; j < S; j += 64) {
        const float w = weights[base + j];
        const float* a = albedo + (base + j) * 3;
        float i0 = 1.f, i1 = 1.f, i2 = 1.f;
        if (sun_v) {
          const float sv = sun_v[base + j];
          i0 = sv + (1.f - sv) * k0, i1 = sv + (1.f - sv) * k1, i2 = sv + (1.f - sv) * k2;
        }
        c0 += w * a[0] * i0, c1 += w * a[1] * i1, c2 += w * a[2] * i2;
      }
      c0 = wave_sum(c0), c1 = wave_sum(c1), c2 = wave_sum(c2);
      if (!(c0 >= 0.f && c0 <= 1.f)) gr0 = 0.f;
      if (!(c1 >= 0.f && c1 <= 1.f)) gr1 = 0.f;
      if (!(c2 >= 0.f && c2 <= 1.f)) gr2 = 0.f;
    }
  }
  const float gd = g_depth ? g_depth[r] : 0.f;
  float ks0 = 0.f, ks1 = 0.f, ks2 = 0.f;  // d_sky accumulators
  float carry = 0.f;                       // suffix sum from later segments
  const int nseg = (S + 63) / 64;
  for (int seg = nseg - 1; seg >= 0; --seg) {
    const int j = seg * 64 + lane;
    const bool on = j < S;
    float delta = 0.f, dens = 0.f, alpha = 0.f, w = 0.f, T = 0.f, G = 0.f, tail = 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, i0 = 1.f, i1 = 1.f, i2 = 1.f, sv = 1.f;
    if (on) {
      alpha_of(z, sigma, noise, noise_std, base, j, S, delta, dens, alpha);
      w = weights[base + j], T = transp[base + j];
      G = (g_w ? g_w[base + j] : 0.f) + gd * z[base + j];
      if (albedo) {
        const float* a = albedo + (base + j) * 3;
        a0 = a[0], a1 = a[1], a2 = a[2];
        if (sun_v) {
          sv = sun_v[base + j];
          i0 = sv + (1.f - sv) * k0, i1 = sv + (1.f - sv) * k1, i2 = sv + (1.f - sv) * k2;
        }
        G += gr0 * a0 * i0 + gr1 * a1 * i1 + gr2 * a2 * i2;
      }
      tail = G * w + (g_T ? g_T[base + j] * T : 0.f);
    }
    const float incl = wave_rscan_add(tail, lane);  // sum_{k>=j} within the segment
    const float after = incl - tail + carry;        // sum_{k>j} over the whole ray
    carry += __shfl(incl, 0, 64);
    if (on) {
      const float f = (1.0f - alpha) + 1e-10f;
      const float dalpha = G * T - after / f;
      if (d_sigma) {
        const float rl = dens > 0.f ? dens : 0.f;
        d_sigma[base + j] = dens > 0.f ? dalpha * delta * expf(-delta * rl) : 0.f;
      }
      if (albedo) {
        const float q0 = w * gr0, q1 = w * gr1, q2 = w * gr2;
        if (d_albedo) {
          float* da = d_albedo + (base + j) * 3;
          da[0] = q0 * i0, da[1] = q1 * i1, da[2] = q2 * i2;
        }
        const float di0 = q0 * a0, di1 = q1 * a1, di2 = q2 * a2;  // d irradiance
        if (d_sun) d_sun[base + j] = sun_v ? (di0 * (1.f - k0) + di1 * (1.f - k1) + di2 * (1.f - k2)) : 0.f;
        ks0 += di0 * (1.f - sv), ks1 += di1 * (1.f - sv), ks2 += di2 * (1.f - sv);
      }
    }
  }
  if (d_sky) {
    ks0 = wave_sum(ks0), ks1 = wave_sum(ks1), ks2 = wave_sum(ks2);
    if (lane == 0) {
      const bool has = sun_v != nullptr && sky != nullptr;
      d_sky[r * 3] = has ? ks0 : 0.f, d_sky[r * 3 + 1] = has ? ks1 : 0.f, d_sky[r * 3 + 2] = has ? ks2 : 0.f;
    }
  }
}

// ---- importance resampling + merge: rendering.py:10-49,121-125 ----------------------------------------------
constexpr int kMaxMerge = 1024;  // S + I rounded up to a power of two must fit

// cdf[0] = 0, cdf[k] = sum_{j<k} (w_j + eps) / total for nw weights (rendering.py:22-27); one wave, cdf in LDS
__device__ __forceinline__ void build_cdf(const float* w, int nw, float eps, float* cdf, int lane) {
  float tot = 0.f;
  for (int j = lane; j < nw; j += 64) tot += w[j] + eps;
  tot = wave_sum(tot);
  float carry = 0.f;
  for (int j0 = 0; j0 < nw; j0 += 64) {
    const int j = j0 + lane;
    const float p = j < nw ? (w[j] + eps) / tot : 0.f;
    const float incl = wave_scan_add(p, lane);
    if (j < nw) cdf[j + 1] = carry + incl;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) cdf[0] = 0.f;
}

// inverse-cdf sample (rendering.py:36-48): searchsorted(right=True), clamped neighbours, denom < eps -> 1
__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int nw, float ui, float eps) {
#pragma clang fp contract(off)
  int lo = 0, hi = nw + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= ui) lo = mid + 1;
    else hi = mid;
  }
  const int below = lo - 1 > 0 ? lo - 1 : 0;
  const int above = lo < nw ? lo : nw;
  const float cb = cdf[below], ca = cdf[above], bb = bins[below], ba = bins[above];
  float denom = ca - cb;
  if (denom < eps) denom = 1.f;
  const float t = (ui - cb) / denom;
  const float step = t * (ba - bb);
  return bb + step;
}

__global__ void __launch_bounds__(256) sample_pdf_merge_kernel(const float* __restrict__ zc, const float* __restrict__ wc,
                                                              const float* __restrict__ u, long n_rays, int S, int I, float eps,
                                                              int npow2, float* __restrict__ z_fine) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * kRaysPerBlock + wv;
  if (r >= n_rays) return;
  const int nb = S - 1, nw = S - 2;  // bins = interval mid points, weights = w[1:-1]
  float* cdf = reinterpret_cast<float*>(smem) + (size_t)wv * (2 * (size_t)S + npow2);
  float* bins = cdf + S;
  float* buf = bins + S;
  const float* zr = zc + r * S;
  for (int j = lane; j < nb; j += 64) bins[j] = 0.5f * (zr[j] + zr[j + 1]);
  build_cdf(wc + r * S + 1, nw, eps, cdf, lane);
  // merge buffer: coarse depths, then the new samples, padded with +inf
  for (int j = lane; j < S; j += 64) buf[j] = zr[j];
  for (int j = S + I + lane; j < npow2; j += 64) buf[j] = __builtin_inff();
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  for (int i = lane; i < I; i += 64) buf[S + i] = invert_cdf(cdf, bins, nw, u[r * I + i], eps);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // bitonic sort of npow2 values by one wave
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = lane; t < npow2 / 2; t += 64) {
        const int lo_i = ((t / jj) * jj * 2) + (t % jj);
        const int hi_i = lo_i + jj;
        const bool up = (lo_i & k) == 0;
        const float a = buf[lo_i], b = buf[hi_i];
        if ((a > b) == up) buf[lo_i] = b, buf[hi_i] = a;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (int j = lane; j < S + I; j += 64) z_fine[r * (S + I) + j] = buf[j];
}

// stand-alone rendering.sample_pdf(bins (N,nb), weights (N,nb-1), u (N,I)) -> samples (N,I), no merge
__global__ void __launch_bounds__(256) sample_pdf_kernel(const float* __restrict__ bins_g, const float* __restrict__ w, const float* __restrict__ u,
                                                        long n_rays, int nb, int I, float eps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * kRaysPerBlock + wv;
  if (r >= n_rays) return;
  const int nw = nb - 1;
  float* cdf = reinterpret_cast<float*>(smem) + (size_t)wv * 2 * nb;
  float* bins = cdf + nb;
  for (int j = lane; j < nb; j += 64) bins[j] = bins_g[r * nb + j];
  build_cdf(w + r * nw, nw, eps, cdf, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < I; i += 64) out[r * I + i] = invert_cdf(cdf, bins, nw, u[r * I + i], eps);
}

// ---- weight stream pack / gradient unpack ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_stream_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                         const float* __restrict__ scale, long n, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, long n_f16) {
#pragma clang fp contract(off)
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= n) return;  // n is even (pieces are 512 elements)
  const int i0 = idx[i], i1 = idx[i + 1];
  const float v0 = i0 >= 0 ? src[i0] * scale[i] : 0.f;
  const float v1 = i1 >= 0 ? src[i1] * scale[i + 1] : 0.f;
  uint32_t h, l;
  split_bf16x2(v0, v1, h, l);
  if (i < n_f16) h = pack_f16x2_sat(v0, v1);  // the fp16 forward stream (SR_MODE_F16), saturating
  reinterpret_cast<uint32_t*>(hi)[i >> 1] = h;
  if (lo) reinterpret_cast<uint32_t*>(lo)[i >> 1] = l;
}

__global__ void __launch_bounds__(256) pack_all_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ scale,
                                                      long n, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, const int* __restrict__ fidx,
                                                      const float* __restrict__ fscale, long nf, float* __restrict__ fout, float* tick, long n_f16) {
#pragma clang fp contract(off)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (tick != nullptr && t == 0) tick[0] = (float)next_step((uint32_t)tick[0]);  // optimizer step counter of the captured training step (exact past 2^24: ray_device.h)
  const long i = t * 2;
  if (i < n) {
    const int i0 = idx[i], i1 = idx[i + 1];
    const float v0 = i0 >= 0 ? src[i0] * scale[i] : 0.f;
    const float v1 = i1 >= 0 ? src[i1] * scale[i + 1] : 0.f;
    uint32_t h, l;
    split_bf16x2(v0, v1, h, l);
    if (i < n_f16) h = pack_f16x2_sat(v0, v1);
    reinterpret_cast<uint32_t*>(hi)[t] = h;
    if (lo) reinterpret_cast<uint32_t*>(lo)[t] = l;
  } else {
    const long k = t - n / 2;
    if (k < nf) {
      const int j = fidx[k];
      fout[k] = j >= 0 ? src[j] * fscale[k] : 0.f;
    }
  }
}

__global__ void __launch_bounds__(256) gather_scale_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          const float* __restrict__ scale, long n, float* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = idx[i];
  out[i] = k >= 0 ? src[k] * scale[i] : 0.f;
}

// ---- backward of the per-ray sky head (models/satnerf.py:138-143): parameter gradients only ----------------------------
// Block = 32 rays; thread (k = hidden unit, half) reduces 16 rays in registers, the two halves meet in LDS, then one atomicAdd
// per parameter per block (1024 rays -> 32 blocks x 899 atomics).
constexpr int kSkyRays = 32;
__device__ __forceinline__ void sky_bwd_body(int block, const float* __restrict__ sun, int sun_stride, long n, int hidden,
                                             const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                             const float* __restrict__ sky, const float* __restrict__ d_sky, float* __restrict__ g_w1,
                                             float* __restrict__ g_b1, float* __restrict__ g_w2, float* __restrict__ g_b2) {
  __shared__ float ray[kSkyRays][8];   // sun xyz, dz0..2 per ray
  __shared__ float comb[128][7];
  const long r0 = (long)block * kSkyRays;
  const int nr = (int)((r0 + kSkyRays < n ? r0 + kSkyRays : n) - r0);
  if (threadIdx.x < nr) {
    const long r = r0 + threadIdx.x;
    const float s0 = sky[r * 3], s1 = sky[r * 3 + 1], s2 = sky[r * 3 + 2];
    ray[threadIdx.x][0] = sun[r * sun_stride], ray[threadIdx.x][1] = sun[r * sun_stride + 1], ray[threadIdx.x][2] = sun[r * sun_stride + 2];
    ray[threadIdx.x][3] = d_sky[r * 3] * s0 * (1.f - s0);
    ray[threadIdx.x][4] = d_sky[r * 3 + 1] * s1 * (1.f - s1);
    ray[threadIdx.x][5] = d_sky[r * 3 + 2] * s2 * (1.f - s2);
  }
  __syncthreads();
  const int half = threadIdx.x >> 7, kk = threadIdx.x & 127;
  for (int k0 = 0; k0 < hidden; k0 += 128) {
    const int k = k0 + kk;
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < hidden) {
      const float wx = w1[k * 3], wy = w1[k * 3 + 1], wz = w1[k * 3 + 2], bb = b1[k];
      const float v0 = w2[k], v1 = w2[hidden + k], v2 = w2[2 * hidden + k];
      for (int i = half * (kSkyRays / 2); i < (half + 1) * (kSkyRays / 2) && i < nr; ++i) {
        const float sx = ray[i][0], sy = ray[i][1], sz = ray[i][2], z0 = ray[i][3], z1 = ray[i][4], z2 = ray[i][5];
        const float hk = __builtin_fmaf(wz, sz, __builtin_fmaf(wy, sy, __builtin_fmaf(wx, sx, bb)));
        if (hk > 0.f) {
          a[4] += z0 * hk, a[5] += z1 * hk, a[6] += z2 * hk;
          const float dh = z0 * v0 + z1 * v1 + z2 * v2;
          a[3] += dh, a[0] += dh * sx, a[1] += dh * sy, a[2] += dh * sz;
        }
      }
    }
    if (half == 1) {
#pragma unroll
      for (int c = 0; c < 7; ++c) comb[kk][c] = a[c];
    }
    __syncthreads();
    if (half == 0 && k < hidden) {
#pragma unroll
      for (int c = 0; c < 7; ++c) a[c] += comb[kk][c];
      atomicAdd(&g_w1[k * 3], a[0]), atomicAdd(&g_w1[k * 3 + 1], a[1]), atomicAdd(&g_w1[k * 3 + 2], a[2]), atomicAdd(&g_b1[k], a[3]);
      atomicAdd(&g_w2[k], a[4]), atomicAdd(&g_w2[hidden + k], a[5]), atomicAdd(&g_w2[2 * hidden + k], a[6]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    float acc = 0.f;
    for (int i = 0; i < nr; ++i) acc += ray[i][3 + threadIdx.x];
    atomicAdd(&g_b2[threadIdx.x], acc);
  }
}

__global__ void __launch_bounds__(256) sky_bwd_kernel(const float* __restrict__ sun, int sun_stride, long n, int hidden,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ sky,
                                                     const float* __restrict__ d_sky, float* __restrict__ g_w1, float* __restrict__ g_b1,
                                                     float* __restrict__ g_w2, float* __restrict__ g_b2) {
  sky_bwd_body(blockIdx.x, sun, sun_stride, n, hidden, w1, b1, w2, sky, d_sky, g_w1, g_b1, g_w2, g_b2);
}

// ---- embedding gradient (nn.Embedding backward, rendering.py:100): d_emb[ts[r]] += sum_j d_t[r, j, :] ----------------------
__device__ __forceinline__ void embedding_bwd_body(int block, const float* __restrict__ d_t, const long long* __restrict__ ts, long n_rays,
                                                   int S, int tau, float* __restrict__ g_emb) {
  const int lane = threadIdx.x & 63;
  const long r = (long)block * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const long row = ts[r];
  const float* src = d_t + r * (long)S * tau;  // the ray's S x tau block is contiguous
  if (64 % tau == 0) {  // lane k always meets component k % tau: coalesced sweep, then reduce lanes of equal k % tau
    float a = 0.f;
    for (int k = lane; k < S * tau; k += 64) a += src[k];
    for (int m = 32; m >= tau; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane < tau) atomicAdd(&g_emb[row * tau + lane], a);
  } else {
    for (int i = 0; i < tau; ++i) {
      float a = 0.f;
      for (int j = lane; j < S; j += 64) a += src[j * tau + i];
      a = wave_sum(a);
      if (lane == 0) atomicAdd(&g_emb[row * tau + i], a);
    }
  }
}

__global__ void __launch_bounds__(256) embedding_bwd_kernel(const float* __restrict__ d_t, const long long* __restrict__ ts, long n_rays,
                                                           int S, int tau, float* __restrict__ g_emb) {
  embedding_bwd_body(blockIdx.x, d_t, ts, n_rays, S, tau, g_emb);
}

// gradient tail of the training fast path: [split-K reduction + scatter of the MLP weight gradients | sky-head gradients |
// embedding gradients] as three block ranges of ONE launch (each sub-kernel is 8-15 us at a ~5 us launch floor)
struct GradTailParams {
  const float* partial; const int* gidx; const float* gscale; long n_params; const int* blocks; float* grad; int accumulate;
  const float* sun; int sun_stride; long n_rays; int hidden; const float* w1; const float* b1; const float* w2; const float* sky;
  const float* d_sky; float* g_w1; float* g_b1; float* g_w2; float* g_b2;
  const float* d_t; const long long* ts; int S; int tau; float* g_emb;
  int blocks_unpack, blocks_sky, blocks_emb;
  int n_blocks;  // rows of the job table `blocks`
};
// (n_slices, first_slice) of the job blocks in LDS: the split-K sums then need ONE dependent memory round trip (gidx -> the slices)
// instead of two (gidx -> table row -> slices).  Tables beyond kTailTab rows are read from memory.
constexpr int kTailTab = 256;
__device__ __forceinline__ void tail_table_to_lds(int2* tab, const int* __restrict__ blocks, int n_blocks) {
  if ((int)threadIdx.x < n_blocks && threadIdx.x < kTailTab)
    tab[threadIdx.x] = int2{blocks[kWgTableInts * threadIdx.x + kWgSlices], blocks[kWgTableInts * threadIdx.x + kWgFirstSlice]};
}
__device__ __forceinline__ float tail_sum(const float* __restrict__ partial, const int2* tab, const int* __restrict__ blocks, int k) {
  const int b = k / kWgBlockFloats;
  const int2 t = b < kTailTab ? tab[b] : int2{blocks[kWgTableInts * b + kWgSlices], blocks[kWgTableInts * b + kWgFirstSlice]};
  return wg_sum_slices(partial, t.x, t.y, k - b * kWgBlockFloats);
}
__global__ void __launch_bounds__(256) grad_tail_kernel(const GradTailParams q) {
  // the two small latency-bound ranges (atomics, one wave per ray) come FIRST in dispatch order so that they run under the cover
  // of the bandwidth-bound reduction instead of forming the kernel's tail
  int b = blockIdx.x;
  if (b < q.blocks_sky) {
    sky_bwd_body(b, q.sun, q.sun_stride, q.n_rays, q.hidden, q.w1, q.b1, q.w2, q.sky, q.d_sky, q.g_w1, q.g_b1, q.g_w2, q.g_b2);
    return;
  }
  b -= q.blocks_sky;
  if (b < q.blocks_emb) {
    embedding_bwd_body(b, q.d_t, q.ts, q.n_rays, q.S, q.tau, q.g_emb);
    return;
  }
  b -= q.blocks_emb;
  __shared__ int2 tab[kTailTab];
  tail_table_to_lds(tab, q.blocks, q.n_blocks);
  const long i = (long)b * 256 + threadIdx.x;
  const bool in = i < q.n_params;
  // everything that does not depend on the sum is requested before it
  const int k = in ? q.gidx[i] : -1;
  const float gs = in ? q.gscale[i] : 0.f;
  const float g0 = in && q.accumulate ? q.grad[i] : 0.f;
  __syncthreads();
  if (k < 0) return;
  q.grad[i] = g0 + tail_sum(q.partial, tab, q.blocks, k) * gs;
}

// ---- gradient tail + Adam in ONE launch (single-GPU captured step) -------------------------------------------------------------------
// grad_tail_kernel followed by adam_graph_kernel streamed the same flat buffers back to back (grad written, read, zeroed: 8 MB and a
// launch).  Here the thread that reduces a parameter's split-K slices applies torch.optim.Adam to that parameter on the spot and leaves
// the gradient slot zero for the next step.  The parameters whose gradients arrive by float atomics from OTHER blocks -- the sky head
// (models/satnerf.py:138-143) and the embedding rows -- are listed in `late`: the last of the atomics' blocks to finish (arrival counter
// behind a __threadfence, reset for the next launch) updates them.  Same adam_one as the stand-alone launches; the bias corrections come
// from the device-side step counter state[0] (ticked by sr_pack_all earlier in the captured step), lr < 0 reads state[1].
struct TailAdamParams {
  GradTailParams q;
  float* p; float* m; float* v;   // flat parameter / moment buffers, aligned with q.grad (element i of all four = parameter i)
  const int* late; int n_late;    // flat indices (relative to q.grad) updated by the last atomics block
  const float* state; unsigned* arrive;
  float lr, b1, b2, eps, grad_scale;
  sr_pack_scatter pack;           // pack.map != NULL: the updated parameter is also written into the weight streams (r05: no sr_pack_all launch)
};

// The weight streams' copy of ONE freshly updated parameter (what sr_pack_all gathers for the whole stream at the start of the next step):
// `map` lists the parameter's (at most two) places -- the forward stream, the transposed stream of the dX kernel, or the fp32 fc_net.0
// table -- as  position | scale index << 26 | (1 << 28 for the fp32 table),  -1 = none.  The arithmetic per element is sr_pack_all's
// (src * scale, unfused; bf16 hi = RNE(v), lo = RNE(v - hi); the fp16 part of the stream saturates), so the streams hold the same bits.
__device__ __forceinline__ void pack_scatter_words(const sr_pack_scatter& k, int2 w, float p) {  // w = the parameter's two map words
#pragma clang fp contract(off)
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int c = o == 0 ? w.x : w.y;
    if (c < 0) continue;
    const int pos = c & 0x3ffffff, si = (c >> 26) & 3;
    const float v = p * (si == 0 ? k.scales[0] : si == 1 ? k.scales[1] : si == 2 ? k.scales[2] : k.scales[3]);
    if (c & (1 << 28)) {
      k.l0[pos] = v;
      continue;
    }
    uint32_t h, l;
    split_bf16x2(v, 0.f, h, l);
    if (pos < k.n_f16) h = pack_f16x2_sat(v, 0.f);
    k.hi[pos] = (uint16_t)h;
    if (k.lo) k.lo[pos] = (uint16_t)l;
  }
}
__global__ void __launch_bounds__(256) grad_tail_adam_kernel(const TailAdamParams a) {
  const GradTailParams& q = a.q;
  __shared__ float bc[2];
  __shared__ int last;
  __shared__ int2 tab[kTailTab];
  if (threadIdx.x == 0) {
    const float t = a.state[0];
    bc[0] = 1.0f - powf(a.b1, t), bc[1] = 1.0f - powf(a.b2, t);
  }
  tail_table_to_lds(tab, q.blocks, q.n_blocks);
  __syncthreads();
  const float lr = a.lr < 0.f ? a.state[1] : a.lr;
  const float step_size = lr / bc[0], sqrt_bc2 = sqrtf(bc[1]);
  int b = blockIdx.x;
  const int n_atomic = q.blocks_sky + q.blocks_emb;
  if (b < n_atomic) {
    if (b < q.blocks_sky) sky_bwd_body(b, q.sun, q.sun_stride, q.n_rays, q.hidden, q.w1, q.b1, q.w2, q.sky, q.d_sky, q.g_w1, q.g_b1, q.g_w2, q.g_b2);
    else embedding_bwd_body(b - q.blocks_sky, q.d_t, q.ts, q.n_rays, q.S, q.tau, q.g_emb);
    // Publishing this block's float atomics to the block that arrives last.  Two builds:
    //  SR_TAIL_RELEASE = 1: what the HIP / LLVM memory model asks for -- after the workgroup barrier ONE thread arrives with an agent-scope
    //    acq_rel fetch_add (release: the barrier makes the block's atomics happen-before it; acquire: the last arriver then reads everything
    //    the earlier ones released), and the barrier behind it carries that to the block's other threads.
    //  SR_TAIL_RELEASE = 0 (the r05 code): relaxed arrive behind `s_waitcnt vmcnt(0)`.  Rests on gfx950 behaviour, not on the model: agent-scope
    //    float atomics are performed at the memory side (not in an XCD's L2) and vmcnt acknowledges them once performed, so a counter
    //    increment issued afterwards cannot overtake them.  A per-THREAD agent-scope release (__threadfence) writes the XCD's whole L2
    //    back (buffer_wbl2) 73 k times while the other blocks stream 18 MB of moments: 64 us instead of 28 (r05).
    //  Measured (r06, tools/ab_tail.py, same box, interleaved): release build 34.1-34.3 us per launch, relaxed 24.5-25.0 (r05 tree: 25.0-25.6):
    //  even ONE agent-scope release per block (288 L2 write-backs while 18 MB of moments stream through) costs 9.5 us of a 25-us launch.
    //  The product build is therefore the relaxed one, compiled for gfx950 only (the #error below) and held by a stress test
    //  (tests/test_hip_step_fusion.py: 400 launches beside unrelated traffic, late parameters against the unfused pair); -DSR_TAIL_RELEASE=1
    //  builds the model-conforming variant for any other part.
#ifndef SR_TAIL_RELEASE
#define SR_TAIL_RELEASE 0
#endif
#if SR_TAIL_RELEASE
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_atomic - 1);
    __syncthreads();
#else
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "the relaxed arrive of grad_tail_adam_kernel relies on gfx950's memory-side atomics"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_atomic - 1);
    __syncthreads();
#endif
    if (!last) return;
    for (int k = threadIdx.x; k < a.n_late; k += 256) {
      const int i = a.late[k];
      float g = __hip_atomic_load(q.grad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the atomics live in L2)
      adam_one(a.p[i], g, a.m[i], a.v[i], step_size, a.b1, a.b2, a.eps, a.grad_scale, sqrt_bc2, 1);
      q.grad[i] = 0.f;
    }
    if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  b -= n_atomic;
  const long i = (long)b * 256 + threadIdx.x;
  const bool in = i < q.n_params;
  // everything that does not depend on the split-K sum is requested before it: the launch used to be a chain of five dependent memory
  // round trips per thread (gidx -> table row -> 10 slices -> 8 slices -> optimizer state), 74 % of its wave cycles parked (r05 PMC)
  const int k = in ? q.gidx[i] : -1;  // (< 0: no weight-gradient GEMM produces it: on the `late` list)
  const bool live = k >= 0;
  const float gs = in ? q.gscale[i] : 0.f;
  const float g0 = in && q.accumulate ? q.grad[i] : 0.f;  // what the solar-correction / depth-supervision passes of this step left
  float p = in ? a.p[i] : 0.f, m = in ? a.m[i] : 0.f, v = in ? a.v[i] : 0.f;
  int2 w = int2{-1, -1};
  if (a.pack.map != nullptr && in) w = reinterpret_cast<const int2*>(a.pack.map)[i];
  if (!live) return;
  float g = g0 + tail_sum(q.partial, tab, q.blocks, k) * gs;
  adam_one(p, g, m, v, step_size, a.b1, a.b2, a.eps, a.grad_scale, sqrt_bc2, 1);
  a.p[i] = p, a.m[i] = m, a.v[i] = v;
  q.grad[i] = 0.f;
  if (a.pack.map != nullptr) pack_scatter_words(a.pack, w, p);
}

// ---- Adam + re-pack in ONE launch (data-parallel captured step, r06) ---------------------------------------------------------------------
// With more than one rank the split-K reduction (sr_grad_tail) and the update are separated by the gradient all-reduce; the update launch
// then does what the single-GPU tail does after its sums: torch.optim.Adam on every element of the flat buffers (adam_one), the gradient
// slot zeroed, and each of the first n_packed parameters (the coarse model's) written into its places of the weight streams
// (pack_scatter_words) -- the N > 1 step is the N = 1 step plus one collective, with no sr_pack_all and no separate Adam launch.
struct AdamPackParams {
  float* p; float* g; float* m; float* v; long n, n_packed;
  const float* state;
  float lr, b1, b2, eps, grad_scale;
  int zero_grad;
  sr_pack_scatter pack;
};
__global__ void __launch_bounds__(256) adam_pack_kernel(const AdamPackParams a) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const float t = a.state[0];
    bc[0] = 1.0f - powf(a.b1, t), bc[1] = 1.0f - powf(a.b2, t);
  }
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool in = i < a.n;
  float p = in ? a.p[i] : 0.f, g = in ? a.g[i] : 0.f, m = in ? a.m[i] : 0.f, v = in ? a.v[i] : 0.f;
  int2 w = int2{-1, -1};
  if (in && i < a.n_packed) w = reinterpret_cast<const int2*>(a.pack.map)[i];
  __syncthreads();
  if (!in) return;
  const float lr = a.lr < 0.f ? a.state[1] : a.lr;
  adam_one(p, g, m, v, lr / bc[0], a.b1, a.b2, a.eps, a.grad_scale, sqrtf(bc[1]), a.zero_grad);
  a.p[i] = p, a.m[i] = m, a.v[i] = v;
  if (a.zero_grad) a.g[i] = 0.f;
  if (i < a.n_packed) pack_scatter_words(a.pack, w, p);
}

}  // namespace sr

using namespace sr;

extern "C" int sr_version(void) { return SR_VERSION; }
extern "C" const char* sr_last_error(void) { return sr::g_err; }

extern "C" int sr_ray_sample_fwd(const float* rays, int ray_stride, const float* u, int64_t n_rays, int n_samples, float* z_vals,
                                 void* stream) {
  SR_REQUIRE(rays && u && z_vals, "sr_ray_sample_fwd: null pointer");
  SR_REQUIRE(ray_stride >= 8 && n_samples >= 2, "sr_ray_sample_fwd: ray_stride>=8 and n_samples>=2 required (got %d, %d)", ray_stride, n_samples);
  if (n_rays <= 0) return 0;
  const long tot = (long)n_rays * n_samples;
  hipLaunchKernelGGL(ray_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, u,
                     (long)n_rays, n_samples, z_vals);
  return check_launch("ray_sample_kernel");
}

extern "C" int sr_ray_setup(const float* rays, int ray_stride, const float* u, int64_t n_rays, int n_samples, int hidden, const float* w1,
                            const float* b1, const float* w2, const float* b2, float* z_vals, float* sky, void* stream) {
  SR_REQUIRE(rays && u && w1 && b1 && w2 && b2 && z_vals && sky, "sr_ray_setup: null pointer");
  SR_REQUIRE(ray_stride >= 11 && n_samples >= 2, "sr_ray_setup: ray_stride>=11 and n_samples>=2 required");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0, (hipStream_t)stream, rays,
                     ray_stride, u, (long)n_rays, n_samples, hidden, w1, b1, w2, b2, z_vals, sky, 0ull, (float*)nullptr, 0);
  return check_launch("ray_setup_kernel");
}

extern "C" int sr_ray_setup_rng(const float* rays, int ray_stride, uint64_t seed, float* step_counter, int tick, int64_t n_rays, int n_samples,
                                int hidden, const float* w1, const float* b1, const float* w2, const float* b2, float* z_vals, float* sky,
                                void* stream) {
  SR_REQUIRE(!tick || step_counter, "sr_ray_setup_rng: tick needs the 4-float step counter block");
  SR_REQUIRE(rays && w1 && b1 && w2 && b2 && z_vals && sky, "sr_ray_setup_rng: null pointer");
  SR_REQUIRE(ray_stride >= 11 && n_samples >= 2, "sr_ray_setup_rng: ray_stride>=11 and n_samples>=2 required");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0, (hipStream_t)stream, rays,
                     ray_stride, (const float*)nullptr, (long)n_rays, n_samples, hidden, w1, b1, w2, b2, z_vals, sky, (unsigned long long)seed,
                     step_counter, tick);
  return check_launch("ray_setup_kernel");
}

extern "C" int sr_sky_fwd(const float* sun, int sun_stride, int64_t n, int hidden, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* sky, void* stream) {
  SR_REQUIRE(sun && w1 && b1 && w2 && b2 && sky, "sr_sky_fwd: null pointer");
  SR_REQUIRE(sun_stride >= 3 && hidden >= 1, "sr_sky_fwd: bad sun_stride/hidden");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sky_kernel, dim3((unsigned)((n + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0, (hipStream_t)stream, sun,
                     sun_stride, (long)n, hidden, w1, b1, w2, b2, sky);
  return check_launch("sky_kernel");
}

extern "C" int sr_composite_fwd(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                                const float* sun_v, const float* sky, int64_t n_rays, int n_samples, int clamp_rgb, float* weights,
                                float* transparency, float* depth, float* rgb, void* stream) {
  SR_REQUIRE(z_vals && sigma && weights && transparency, "sr_composite_fwd: null pointer");
  SR_REQUIRE(n_samples >= 1, "sr_composite_fwd: n_samples must be >= 1");
  SR_REQUIRE((sun_v == nullptr) == (sky == nullptr), "sr_composite_fwd: sun_v and sky must be given together");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0,
                     (hipStream_t)stream, z_vals, sigma, noise, noise_std, albedo, sun_v, sky, (long)n_rays, n_samples, clamp_rgb,
                     weights, transparency, depth, rgb);
  return check_launch("composite_fwd_kernel");
}

extern "C" int sr_composite_bwd(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                                const float* sun_v, const float* sky, const float* weights, const float* transparency,
                                const float* rgb_unclamped_or_null, int64_t n_rays, int n_samples, int clamp_rgb, const float* g_rgb,
                                const float* g_depth, const float* g_weights, const float* g_transparency, float* d_sigma,
                                float* d_albedo, float* d_sun_v, float* d_sky, void* stream) {
  (void)rgb_unclamped_or_null;  // recomputed in-kernel
  SR_REQUIRE(z_vals && sigma && weights && transparency, "sr_composite_bwd: null pointer");
  SR_REQUIRE((sun_v == nullptr) == (sky == nullptr), "sr_composite_bwd: sun_v and sky must be given together");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0,
                     (hipStream_t)stream, z_vals, sigma, noise, noise_std, albedo, sun_v, sky, weights, transparency, (long)n_rays,
                     n_samples, clamp_rgb, g_rgb, g_depth, g_weights, g_transparency, d_sigma, d_albedo, d_sun_v, d_sky);
  return check_launch("composite_bwd_kernel");
}

extern "C" int sr_composite_image(const float* z_vals, const float* sigma, const float* noise, float noise_std, const float* albedo,
                                  const float* sun_v, const float* beta, const float* sky, int64_t n_rays, int n_samples, float* image,
                                  void* stream) {
  if (n_rays <= 0) return 0;  // empty batch: nothing to do (empty tensors carry null pointers)
  SR_REQUIRE(z_vals && sigma && albedo && sun_v && beta && sky && image, "sr_composite_image: null pointer");
  SR_REQUIRE(n_samples >= 1, "sr_composite_image: n_samples must be >= 1");
  hipLaunchKernelGGL(composite_image_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0, (hipStream_t)stream,
                     z_vals, sigma, noise, noise_std, albedo, sun_v, beta, sky, (long)n_rays, n_samples, image);
  return check_launch("composite_image_kernel");
}

extern "C" int sr_latlonalt_from_depth(const float* rays, int ray_stride, const float* depth, int64_t n_rays, const double* center, double range,
                                       double* lat, double* lon, double* alt, void* stream) {
  if (n_rays <= 0) return 0;
  SR_REQUIRE(rays && depth && center && lat && lon && alt, "sr_latlonalt_from_depth: null pointer");
  SR_REQUIRE(ray_stride >= 6, "sr_latlonalt_from_depth: ray_stride must be >= 6 (got %d)", ray_stride);
  hipLaunchKernelGGL(latlonalt_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, depth,
                     (long)n_rays, center[0], center[1], center[2], range, lat, lon, alt);
  return check_launch("latlonalt_kernel");
}

extern "C" int sr_sample_pdf_merge(const float* z_coarse, const float* weights_coarse, const float* u, int64_t n_rays, int n_samples,
                                   int n_importance, float eps, float* z_fine, void* stream) {
  SR_REQUIRE(z_coarse && weights_coarse && u && z_fine, "sr_sample_pdf_merge: null pointer");
  SR_REQUIRE(n_samples >= 3 && n_importance >= 1, "sr_sample_pdf_merge: need n_samples>=3, n_importance>=1");
  int npow2 = 64;
  while (npow2 < n_samples + n_importance) npow2 <<= 1;
  SR_REQUIRE(npow2 <= kMaxMerge, "sr_sample_pdf_merge: n_samples+n_importance=%d too large (max %d)", n_samples + n_importance, kMaxMerge);
  if (n_rays <= 0) return 0;
  const size_t lds = (size_t)kRaysPerBlock * (2 * (size_t)n_samples + npow2) * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), lds,
                     (hipStream_t)stream, z_coarse, weights_coarse, u, (long)n_rays, n_samples, n_importance, eps, npow2, z_fine);
  return check_launch("sample_pdf_merge_kernel");
}

extern "C" int sr_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins, int n_importance, float eps,
                             float* samples, void* stream) {
  SR_REQUIRE(bins && weights && u && samples, "sr_sample_pdf: null pointer");
  SR_REQUIRE(n_bins >= 2 && n_importance >= 1, "sr_sample_pdf: need n_bins>=2, n_importance>=1");
  if (n_rays <= 0) return 0;
  const size_t lds = (size_t)kRaysPerBlock * 2 * n_bins * sizeof(float);
  SR_REQUIRE(lds <= 64 * 1024, "sr_sample_pdf: n_bins=%d too large", n_bins);
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), lds, (hipStream_t)stream, bins,
                     weights, u, (long)n_rays, n_bins, n_importance, eps, samples);
  return check_launch("sample_pdf_kernel");
}

extern "C" int sr_pack_stream(const float* src, const int32_t* idx, const float* scale, int64_t n, uint16_t* out_hi, uint16_t* out_lo,
                              int64_t n_f16, void* stream) {
  SR_REQUIRE(src && idx && scale && out_hi, "sr_pack_stream: null pointer");
  SR_REQUIRE(n % 2 == 0 && n_f16 % 2 == 0 && n_f16 >= 0 && n_f16 <= n, "sr_pack_stream: n and n_f16 must be even, n_f16 <= n");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pack_stream_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx, scale,
                     (long)n, out_hi, out_lo, (long)n_f16);
  return check_launch("pack_stream_kernel");
}

extern "C" int sr_pack_all(const float* src, const int32_t* idx, const float* scale, int64_t n, uint16_t* out_hi, uint16_t* out_lo,
                           const int32_t* f32_idx, const float* f32_scale, int64_t n_f32, float* out_f32, float* tick, int64_t n_f16,
                           void* stream) {
  SR_REQUIRE(src && idx && scale && out_hi && f32_idx && f32_scale && out_f32, "sr_pack_all: null pointer");
  SR_REQUIRE(n % 2 == 0 && n_f16 % 2 == 0 && n_f16 >= 0 && n_f16 <= n, "sr_pack_all: n and n_f16 must be even, n_f16 <= n");
  const long threads = n / 2 + n_f32;
  if (threads <= 0) return 0;
  hipLaunchKernelGGL(pack_all_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx, scale, (long)n, out_hi,
                     out_lo, f32_idx, f32_scale, (long)n_f32, out_f32, tick, (long)n_f16);
  return check_launch("pack_all_kernel");
}

extern "C" int sr_gather_scale_f32(const float* src, const int32_t* idx, const float* scale, int64_t n, float* out, void* stream) {
  SR_REQUIRE(src && idx && scale && out, "sr_gather_scale_f32: null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gather_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, idx, scale, (long)n, out);
  return check_launch("gather_scale_kernel");
}

extern "C" int sr_sky_bwd(const float* sun, int sun_stride, int64_t n, int hidden, const float* w1, const float* b1, const float* w2,
                          const float* sky, const float* d_sky, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* stream) {
  SR_REQUIRE(sun && w1 && b1 && w2 && sky && d_sky && g_w1 && g_b1 && g_w2 && g_b2, "sr_sky_bwd: null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sky_bwd_kernel, dim3((unsigned)((n + kSkyRays - 1) / kSkyRays)), dim3(256), 0, (hipStream_t)stream, sun, sun_stride, (long)n, hidden, w1,
                     b1, w2, sky, d_sky, g_w1, g_b1, g_w2, g_b2);
  return check_launch("sky_bwd_kernel");
}

extern "C" int sr_grad_tail(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks, int n_blocks,
                            float* grad, int accumulate, const float* sun, int sun_stride, int64_t n_rays, int hidden, const float* w1,
                            const float* b1, const float* w2, const float* sky, const float* d_sky, float* g_w1, float* g_b1, float* g_w2,
                            float* g_b2, const float* d_t, const int64_t* ts, int n_samples, int tau, float* g_emb, void* stream) {
  SR_REQUIRE(partial && gidx && gscale && blocks && grad && sun && w1 && b1 && w2 && sky && d_sky && g_w1 && g_b1 && g_w2 && g_b2 && d_t && ts && g_emb,
             "sr_grad_tail: null pointer");
  SR_REQUIRE(n_blocks >= 1, "sr_grad_tail: n_blocks = %d", n_blocks);
  if (n_rays <= 0) return 0;
  GradTailParams q;
  q.partial = partial, q.gidx = gidx, q.gscale = gscale, q.n_params = n_params, q.blocks = blocks, q.grad = grad;
  q.accumulate = accumulate, q.sun = sun, q.sun_stride = sun_stride, q.n_rays = n_rays, q.hidden = hidden, q.w1 = w1, q.b1 = b1, q.w2 = w2;
  q.sky = sky, q.d_sky = d_sky, q.g_w1 = g_w1, q.g_b1 = g_b1, q.g_w2 = g_w2, q.g_b2 = g_b2, q.d_t = d_t, q.ts = (const long long*)ts;
  q.S = n_samples, q.tau = tau, q.g_emb = g_emb, q.n_blocks = n_blocks;
  q.blocks_unpack = (int)((n_params + 255) / 256);
  q.blocks_sky = (int)((n_rays + kSkyRays - 1) / kSkyRays);
  q.blocks_emb = (int)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  hipLaunchKernelGGL(grad_tail_kernel, dim3(q.blocks_unpack + q.blocks_sky + q.blocks_emb), dim3(256), 0, (hipStream_t)stream, q);
  return check_launch("grad_tail_kernel");
}

extern "C" int sr_grad_tail_adam(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, const int32_t* blocks, int n_blocks,
                                 float* grad, int accumulate, const float* sun, int sun_stride, int64_t n_rays, int hidden, const float* w1,
                                 const float* b1, const float* w2, const float* sky, const float* d_sky, float* g_w1, float* g_b1, float* g_w2,
                                 float* g_b2, const float* d_t, const int64_t* ts, int n_samples, int tau, float* g_emb, float* params,
                                 float* exp_avg, float* exp_avg_sq, const int32_t* late_idx, int n_late, float* state, float lr, float beta1,
                                 float beta2, float eps, float grad_scale, const sr_pack_scatter* pack, void* stream) {
  SR_REQUIRE(partial && gidx && gscale && blocks && grad && sun && w1 && b1 && w2 && sky && d_sky && g_w1 && g_b1 && g_w2 && g_b2 && d_t && ts && g_emb,
             "sr_grad_tail_adam: null pointer");
  SR_REQUIRE(n_blocks >= 1, "sr_grad_tail_adam: n_blocks = %d", n_blocks);
  SR_REQUIRE(params && exp_avg && exp_avg_sq && state && (late_idx || n_late == 0) && n_late >= 0, "sr_grad_tail_adam: null optimizer pointer");
  if (n_rays <= 0) return 0;
  TailAdamParams a;
  GradTailParams& q = a.q;
  q.partial = partial, q.gidx = gidx, q.gscale = gscale, q.n_params = n_params, q.blocks = blocks, q.grad = grad;
  q.accumulate = accumulate, q.sun = sun, q.sun_stride = sun_stride, q.n_rays = n_rays, q.hidden = hidden, q.w1 = w1, q.b1 = b1, q.w2 = w2;
  q.sky = sky, q.d_sky = d_sky, q.g_w1 = g_w1, q.g_b1 = g_b1, q.g_w2 = g_w2, q.g_b2 = g_b2, q.d_t = d_t, q.ts = (const long long*)ts;
  q.S = n_samples, q.tau = tau, q.g_emb = g_emb, q.n_blocks = n_blocks;
  q.blocks_unpack = (int)((n_params + 255) / 256);
  q.blocks_sky = (int)((n_rays + kSkyRays - 1) / kSkyRays);
  q.blocks_emb = (int)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  a.p = params, a.m = exp_avg, a.v = exp_avg_sq, a.late = late_idx, a.n_late = n_late, a.state = state;
  a.arrive = reinterpret_cast<unsigned*>(state + 3);  // the schedule block's arrival counter (ray_device.h tick_when_all_read: unused by training steps)
  a.lr = lr, a.b1 = beta1, a.b2 = beta2, a.eps = eps, a.grad_scale = grad_scale;
  a.pack = sr_pack_scatter{};
  if (pack != nullptr && pack->map != nullptr) {
    SR_REQUIRE(pack->hi && pack->l0 && pack->n_f16 >= 0, "sr_grad_tail_adam: pack needs the stream and the fc_net.0 table");
    a.pack = *pack;
  }
  hipLaunchKernelGGL(grad_tail_adam_kernel, dim3(q.blocks_unpack + q.blocks_sky + q.blocks_emb), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("grad_tail_adam_kernel");
}

extern "C" int sr_adam_step_pack(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float grad_scale, float* state, int zero_grad, int64_t n_packed, const sr_pack_scatter* pack, void* stream) {
  SR_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, "sr_adam_step_pack: null pointer");
  SR_REQUIRE(n_packed >= 0 && n_packed <= n, "sr_adam_step_pack: n_packed = %lld of %lld", (long long)n_packed, (long long)n);
  if (n <= 0) return 0;
  AdamPackParams a;
  a.p = params, a.g = grads, a.m = exp_avg, a.v = exp_avg_sq, a.n = n, a.n_packed = 0, a.state = state;
  a.lr = lr, a.b1 = beta1, a.b2 = beta2, a.eps = eps, a.grad_scale = grad_scale, a.zero_grad = zero_grad;
  a.pack = sr_pack_scatter{};
  if (pack != nullptr && pack->map != nullptr && n_packed > 0) {
    SR_REQUIRE(pack->hi && pack->l0 && pack->n_f16 >= 0, "sr_adam_step_pack: pack needs the stream and the fc_net.0 table");
    a.pack = *pack, a.n_packed = n_packed;
  }
  hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("adam_pack_kernel");
}

extern "C" int sr_embedding_bwd(const float* d_t, const int64_t* ts, int64_t n_rays, int n_samples, int tau, float* g_emb, void* stream) {
  SR_REQUIRE(d_t && ts && g_emb, "sr_embedding_bwd: null pointer");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)((n_rays + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(256), 0, (hipStream_t)stream,
                     d_t, (const long long*)ts, (long)n_rays, n_samples, tau, g_emb);
  return check_launch("embedding_bwd_kernel");
}
