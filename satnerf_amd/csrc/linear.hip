// Layer-by-layer path for Sat-NeRF widths the fused kernel does not cover (fc_units != 256; opt.py's default is 512).
//
// One tiled MFMA GEMM kernel, three uses -- the three matrix products of an nn.Linear whose input is the concatenation of up
// to two sources with the previous layer's activation applied ON LOAD (so the tensors between layers are pre-activations
// and nothing but GEMM outputs is ever written):
//   FWD  y[p][n]     = out_act( sum_f acat[p][f] * W[n][f] + b[n] )                    models/satnerf.py:156-208, one Linear
//   DX   d_src[p][k] = ( sum_n G[p][n] * W[n][col0 + k] ) * act'(src[p][k])            autograd: grad_input (+ Siren / ReLU)
//   DW   dW[n][f]   += sum_p G[p][n] * acat[p][f] ,  db[n] += sum_p G[p][n]            autograd: grad_weight, grad_bias
// with acat[p][f] = act_s(src_s[p / row_div_s][f - off_s]) and G[p][n] = gy[p][n] * out_act'(y[p][n]).
//
// Arithmetic: fp32 operands are split into bf16 hi + lo while being staged into LDS and every k-step issues
// hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) -- the same 3-pass scheme as the fused kernel's parity
// mode, ~1e-6 relative.  Workgroup = 4 waves on a 128 x 128 output tile (wave = 64 x 64 = 2 x 2 MFMA tiles), k in steps of 32.
// This path favours generality (any sizes, two concatenated sources, per-ray rows) over speed: scalar staging loads, one LDS
// buffer; the width-256 hot path never uses it.
#include "common.h"

namespace sr {

struct LinSrc {
  const float* x;
  int ld, k, act;
  float w0;
  int row_div;
};

struct LinParams {
  LinSrc s[2];
  int n_src, ktot;  // acat columns: [0, s[0].k) from s[0], then s[1]; column ktot reads 1.0 (bias gradient) in DW
  const float* w;
  int ldw;
  const float* bias;
  const float *gy, *y;
  int ldg, ldy, out_act;
  int col0;     // DX: first weight column of the target source
  LinSrc tgt;   // DX: the source whose act' multiplies the product
  float* out;
  int ldo;
  float* dbias;
  long M, N, K;  // GEMM extents of this launch: out is M x N, contraction length K
  long kchunk;   // contraction range per blockIdx.z
};

enum { kFwd = 0, kDx = 1, kDw = 2 };

__device__ __forceinline__ float apply_act(float v, int act, float w0) {
#pragma clang fp contract(off)
  if (act == SR_ACT_SIN) return sin_rev_precise((w0 * v) * 0.15915494309189533577f);
  if (act == SR_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float act_grad(float v, int act, float w0) {
#pragma clang fp contract(off)
  if (act == SR_ACT_SIN) return w0 * sin_rev_precise((w0 * v) * 0.15915494309189533577f + 0.25f);
  if (act == SR_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  return 1.f;
}
__device__ __forceinline__ float out_act_fwd(float v, int oa) {
#pragma clang fp contract(off)
  if (oa == SR_OUT_SOFTPLUS) return softplus_f(v);
  if (oa == SR_OUT_SIGMOID) return sigmoid_f(v);
  if (oa == SR_OUT_SIGMOID_RGB) return sigmoid_f(v) * 1.002f - 0.001f;  // models/satnerf.py:195-196, rgb_padding = 0.001
  return v;
}
__device__ __forceinline__ float out_act_grad(float y, int oa) {  // derivative w.r.t. the pre-activation, from the OUTPUT y
#pragma clang fp contract(off)
  if (oa == SR_OUT_SOFTPLUS) return 1.f - expf(-y);
  if (oa == SR_OUT_SIGMOID) return y * (1.f - y);
  if (oa == SR_OUT_SIGMOID_RGB) {
    const float s = (y + 0.001f) / 1.002f;
    return 1.002f * s * (1.f - s);
  }
  return 1.f;
}

__device__ __forceinline__ float fetch_acat(const LinParams& q, long p, int f, bool ones_col) {
  if (f < q.s[0].k) return apply_act(q.s[0].x[(p / q.s[0].row_div) * q.s[0].ld + f], q.s[0].act, q.s[0].w0);
  f -= q.s[0].k;
  if (q.n_src > 1 && f < q.s[1].k) return apply_act(q.s[1].x[(p / q.s[1].row_div) * q.s[1].ld + f], q.s[1].act, q.s[1].w0);
  return (ones_col && f == (q.n_src > 1 ? q.s[1].k : 0)) ? 1.f : 0.f;
}
__device__ __forceinline__ float fetch_g(const LinParams& q, long p, int n) {
  const float g = q.gy[p * q.ldg + n];
  return q.out_act == SR_OUT_NONE ? g : g * out_act_grad(q.y[p * q.ldy + n], q.out_act);
}

constexpr int kBM = 128, kBN = 128, kBK = 32, kRow = 40;  // LDS row = 32 bf16 + 8 of padding (80 B: 16-byte aligned fragments)
constexpr int kPlane = kBM * kRow;                          // halfwords per plane

// Aop(i, k) / Bop(j, k) for the three products
template <int KIND>
__device__ __forceinline__ float fetch_a(const LinParams& q, long i, long k) {
  if (i >= q.M || k >= q.K) return 0.f;
  if (KIND == kFwd) return fetch_acat(q, i, (int)k, false);
  if (KIND == kDx) return fetch_g(q, i, (int)k);
  return fetch_g(q, k, (int)i);
}
template <int KIND>
__device__ __forceinline__ float fetch_b(const LinParams& q, long j, long k) {
  if (j >= q.N || k >= q.K) return 0.f;
  if (KIND == kFwd) return q.w[j * q.ldw + k];
  if (KIND == kDx) return q.w[k * q.ldw + q.col0 + j];
  return fetch_acat(q, k, (int)j, true);
}

template <int KIND>
__global__ void __launch_bounds__(256) linear_kernel(const LinParams q) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * kPlane];  // A hi, A lo, B hi, B lo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long i0 = (long)blockIdx.y * kBM, j0 = (long)blockIdx.x * kBN;
  const long k_begin = (long)blockIdx.z * q.kchunk;
  long k_end = k_begin + q.kchunk;
  if (k_end > q.K) k_end = q.K;
  // operand memory is contiguous along k (FWD both, DX A) or along the row index (DX B, DW both): consecutive threads follow it
  constexpr bool kARowMajor = KIND == kDw, kBRowMajor = KIND != kFwd;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;

  f32x16 acc[2][2] = {};
  for (long kt = k_begin; kt < k_end; kt += kBK) {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      int r, k;
      if (kARowMajor) r = tid & 127, k = (tid >> 7) + 2 * it;
      else k = tid & 31, r = (tid >> 5) + 8 * it;
      float v = 0.f;
      if (kt + k < k_end) v = fetch_a<KIND>(q, i0 + r, kt + k);
      const uint32_t hi = pack_bf16x2(v, 0.f) & 0xffffu;
      lds[r * kRow + k] = (uint16_t)hi;
      lds[kPlane + r * kRow + k] = (uint16_t)(pack_bf16x2(v - bf16_lo_to_f32(hi), 0.f) & 0xffffu);
    }
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      int r, k;
      if (kBRowMajor) r = tid & 127, k = (tid >> 7) + 2 * it;
      else k = tid & 31, r = (tid >> 5) + 8 * it;
      float v = 0.f;
      if (kt + k < k_end) v = fetch_b<KIND>(q, j0 + r, kt + k);
      const uint32_t hi = pack_bf16x2(v, 0.f) & 0xffffu;
      lds[2 * kPlane + r * kRow + k] = (uint16_t)hi;
      lds[3 * kPlane + r * kRow + k] = (uint16_t)(pack_bf16x2(v - bf16_lo_to_f32(hi), 0.f) & 0xffffu);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = (wm * 64 + t * 32 + l31) * kRow + ks * 16 + h * 8, rb = (wn * 64 + t * 32 + l31) * kRow + ks * 16 + h * 8;
        ah[t] = *reinterpret_cast<const uint4*>(lds + ra), al[t] = *reinterpret_cast<const uint4*>(lds + kPlane + ra);
        bh[t] = *reinterpret_cast<const uint4*>(lds + 2 * kPlane + rb), bl[t] = *reinterpret_cast<const uint4*>(lds + 3 * kPlane + rb);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x16 c = acc[rt][ct];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[rt]), __builtin_bit_cast(bf16x8, bh[ct]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[rt]), __builtin_bit_cast(bf16x8, bl[ct]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[rt]), __builtin_bit_cast(bf16x8, bh[ct]), c, 0, 0, 0);
          acc[rt][ct] = c;
        }
    }
    __syncthreads();
  }

#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const long i = i0 + wm * 64 + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        const long j = j0 + wn * 64 + ct * 32 + l31;
        if (i >= q.M || j >= q.N) continue;
        float v = acc[rt][ct][g];
        if (KIND == kFwd) {
          if (q.bias) v += q.bias[j];
          q.out[i * q.ldo + j] = out_act_fwd(v, q.out_act);
        } else if (KIND == kDx) {
          if (q.tgt.act != SR_ACT_NONE) v *= act_grad(q.tgt.x[i * q.tgt.ld + j], q.tgt.act, q.tgt.w0);
          q.out[i * q.ldo + j] = v;
        } else {
          if (j < q.ktot) unsafeAtomicAdd(q.out + i * q.ldo + j, v);
          else if (q.dbias) unsafeAtomicAdd(q.dbias + i, v);
        }
      }
}

__global__ void __launch_bounds__(256) points_along_kernel(const float* __restrict__ rays, int ray_stride, int dir_col,
                                                          const float* __restrict__ z, long n_points, int S, float* __restrict__ xyz) {
#pragma clang fp contract(off)
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_points) return;
  const float* r = rays + (p / S) * ray_stride;
  const float zz = z[p];
  xyz[p * 3] = r[0] + r[dir_col] * zz, xyz[p * 3 + 1] = r[1] + r[dir_col + 1] * zz, xyz[p * 3 + 2] = r[2] + r[dir_col + 2] * zz;
}

// Mapping.forward (models/nerf.py:53-69): out[r] = [sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...], x = the row's `dim`
// values, NO identity term; frequencies up to 2^9 on coordinates up to ~6 reach thousands of radians, hence the library
// sinf/cosf (full range reduction) instead of the revolution-based fast path.
__global__ void __launch_bounds__(256) positional_map_kernel(const float* __restrict__ x, int ld, int dim, long rows, int n_freqs,
                                                            float* __restrict__ out) {
#pragma clang fp contract(off)
  const int width = 2 * n_freqs * dim;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * width) return;
  const long r = e / width;
  const int c = (int)(e - r * width);
  const int f = c / (2 * dim), fn = (c / dim) & 1, ch = c % dim;
  const float arg = (float)(1 << f) * x[r * ld + ch];
  out[e] = fn ? cosf(arg) : sinf(arg);
}

static int copy_src(const sr_linear_src& in, LinSrc& o, const char* what) {
  SR_REQUIRE(in.x != nullptr && in.k >= 1 && in.ld >= in.k && in.row_div >= 1, "%s: bad source (k=%d ld=%d row_div=%d)", what, in.k, in.ld, in.row_div);
  SR_REQUIRE(in.act == SR_ACT_NONE || in.act == SR_ACT_SIN || in.act == SR_ACT_RELU, "%s: unknown activation %d", what, in.act);
  o.x = in.x, o.ld = in.ld, o.k = in.k, o.act = in.act, o.w0 = in.w0, o.row_div = in.row_div;
  return 0;
}
static bool out_act_ok(int oa) { return oa == SR_OUT_NONE || oa == SR_OUT_SOFTPLUS || oa == SR_OUT_SIGMOID || oa == SR_OUT_SIGMOID_RGB; }

}  // namespace sr

using namespace sr;

extern "C" int sr_points_along(const float* rays, int ray_stride, int dir_col, const float* z_vals, int64_t n_rays, int n_samples, float* xyz,
                               void* stream) {
  SR_REQUIRE(rays && z_vals && xyz, "sr_points_along: null pointer");
  SR_REQUIRE(dir_col >= 3 && ray_stride >= dir_col + 3 && n_samples >= 1, "sr_points_along: bad layout (stride %d, dir_col %d)", ray_stride, dir_col);
  const long n = (long)n_rays * n_samples;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(points_along_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, dir_col, z_vals, n,
                     n_samples, xyz);
  return check_launch("points_along_kernel");
}

extern "C" int sr_positional_map(const float* x, int ld, int dim, int64_t rows, int n_freqs, float* out, void* stream) {
  SR_REQUIRE(x && out, "sr_positional_map: null pointer");
  SR_REQUIRE(dim >= 1 && ld >= dim && n_freqs >= 1 && n_freqs <= 24, "sr_positional_map: bad sizes (dim %d, ld %d, %d frequencies)", dim, ld, n_freqs);
  const long n = (long)rows * 2 * n_freqs * dim;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(positional_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, dim, (long)rows, n_freqs, out);
  return check_launch("positional_map_kernel");
}

extern "C" int sr_linear_fwd(const sr_linear_src* src, int n_src, const float* weight, const float* bias, int64_t n_points, int n_out, int out_act,
                             float* y, int ldy, void* stream) {
  SR_REQUIRE(src && weight && y, "sr_linear_fwd: null pointer");
  SR_REQUIRE((n_src == 1 || n_src == 2) && n_out >= 1 && ldy >= n_out && out_act_ok(out_act), "sr_linear_fwd: bad arguments");
  if (n_points <= 0) return 0;
  LinParams q = {};
  for (int s = 0; s < n_src; ++s)
    if (copy_src(src[s], q.s[s], "sr_linear_fwd")) return 1;
  q.n_src = n_src, q.ktot = q.s[0].k + (n_src > 1 ? q.s[1].k : 0);
  q.w = weight, q.ldw = q.ktot, q.bias = bias, q.out_act = out_act, q.out = y, q.ldo = ldy;
  q.M = n_points, q.N = n_out, q.K = q.ktot, q.kchunk = q.K;
  hipLaunchKernelGGL(linear_kernel<kFwd>, dim3((unsigned)((q.N + kBN - 1) / kBN), (unsigned)((q.M + kBM - 1) / kBM), 1), dim3(256), 0,
                     (hipStream_t)stream, q);
  return check_launch("linear_kernel<fwd>");
}

extern "C" int sr_linear_bwd_input(const float* gy, int ldg, const float* y, int ldy, int out_act, const float* weight, int k_total, int col0,
                                   const sr_linear_src* target, int64_t n_points, int n_out, float* d_src, int ldd, void* stream) {
  SR_REQUIRE(gy && weight && target && d_src, "sr_linear_bwd_input: null pointer");
  SR_REQUIRE(out_act_ok(out_act) && (out_act == SR_OUT_NONE || y != nullptr), "sr_linear_bwd_input: output activation %d needs y", out_act);
  if (n_points <= 0) return 0;
  LinParams q = {};
  if (copy_src(*target, q.tgt, "sr_linear_bwd_input")) return 1;
  SR_REQUIRE(q.tgt.act == SR_ACT_NONE || q.tgt.row_div == 1, "sr_linear_bwd_input: an activated source must be per point (row_div 1)");
  SR_REQUIRE(col0 >= 0 && col0 + q.tgt.k <= k_total && ldd >= q.tgt.k && ldg >= n_out, "sr_linear_bwd_input: bad column range");
  q.gy = gy, q.ldg = ldg, q.y = y, q.ldy = ldy, q.out_act = out_act, q.w = weight, q.ldw = k_total, q.col0 = col0;
  q.out = d_src, q.ldo = ldd;
  q.M = n_points, q.N = q.tgt.k, q.K = n_out, q.kchunk = q.K;
  hipLaunchKernelGGL(linear_kernel<kDx>, dim3((unsigned)((q.N + kBN - 1) / kBN), (unsigned)((q.M + kBM - 1) / kBM), 1), dim3(256), 0,
                     (hipStream_t)stream, q);
  return check_launch("linear_kernel<dx>");
}

extern "C" int sr_linear_bwd_weight(const float* gy, int ldg, const float* y, int ldy, int out_act, const sr_linear_src* src, int n_src,
                                    int64_t n_points, int n_out, float* d_weight, float* d_bias, void* stream) {
  SR_REQUIRE(gy && src && d_weight, "sr_linear_bwd_weight: null pointer");
  SR_REQUIRE((n_src == 1 || n_src == 2) && out_act_ok(out_act) && (out_act == SR_OUT_NONE || y != nullptr) && ldg >= n_out,
             "sr_linear_bwd_weight: bad arguments");
  if (n_points <= 0) return 0;
  LinParams q = {};
  for (int s = 0; s < n_src; ++s)
    if (copy_src(src[s], q.s[s], "sr_linear_bwd_weight")) return 1;
  q.n_src = n_src, q.ktot = q.s[0].k + (n_src > 1 ? q.s[1].k : 0);
  q.gy = gy, q.ldg = ldg, q.y = y, q.ldy = ldy, q.out_act = out_act, q.out = d_weight, q.ldo = q.ktot, q.dbias = d_bias;
  q.M = n_out, q.N = q.ktot + 1, q.K = n_points;  // column ktot = the bias gradient
  q.kchunk = 2048;                                 // points per workgroup: split-K with fp32 atomics
  const unsigned splits = (unsigned)((q.K + q.kchunk - 1) / q.kchunk);
  hipLaunchKernelGGL(linear_kernel<kDw>, dim3((unsigned)((q.N + kBN - 1) / kBN), (unsigned)((q.M + kBM - 1) / kBM), splits), dim3(256), 0,
                     (hipStream_t)stream, q);
  return check_launch("linear_kernel<dw>");
}
