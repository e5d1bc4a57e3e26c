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
  // the hardware sine of the exact fraction, as the fused parity kernels (mlp_fwd.inc: 1.6e-6 of the reference end to end); SR_LINEAR_POLY_SIN:
  // the degree-11 polynomial it replaces (r05: ~22 VALU instructions per staged element made the staging, not the MFMAs, this kernel's bound)
#ifdef SR_LINEAR_POLY_SIN
  if (act == SR_ACT_SIN) return sin_rev_precise((w0 * v) * 0.15915494309189533577f);
#else
  if (act == SR_ACT_SIN) return sin_rev_fast(__builtin_amdgcn_fractf((w0 * v) * 0.15915494309189533577f));
#endif
  if (act == SR_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float act_grad(float v, int act, float w0) {
#pragma clang fp contract(off)
#ifdef SR_LINEAR_POLY_SIN
  if (act == SR_ACT_SIN) return w0 * sin_rev_precise((w0 * v) * 0.15915494309189533577f + 0.25f);
#else
  if (act == SR_ACT_SIN) return w0 * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf((w0 * v) * 0.15915494309189533577f));
#endif
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

struct Vec4 {
  float v[4];
};
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__device__ __forceinline__ Vec4 load4(const float* p) {  // 4 consecutive floats, one 16-byte load when aligned
  Vec4 r;
  if (aligned16(p)) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x, r.v[1] = t.y, r.v[2] = t.z, r.v[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = p[e];
  }
  return r;
}
__device__ __forceinline__ long row_of(long p, int div) { return div == 1 ? p : (long)((unsigned long)p / (unsigned)div); }

// acat[p][f0 .. f0+3]: r0 / r1 = row p of the two sources (already divided by row_div)
__device__ __forceinline__ Vec4 acat4(const LinParams& q, const float* r0, const float* r1, int f0, bool ones_col) {
  Vec4 r;
  const int k0 = q.s[0].k;
  if (f0 + 4 <= k0) {
    r = load4(r0 + f0);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = apply_act(r.v[e], q.s[0].act, q.s[0].w0);
  } else if (q.n_src > 1 && f0 >= k0 && f0 - k0 + 4 <= q.s[1].k) {
    r = load4(r1 + (f0 - k0));
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = apply_act(r.v[e], q.s[1].act, q.s[1].w0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = f0 + e;
      float v = 0.f;
      if (f < k0) v = apply_act(r0[f], q.s[0].act, q.s[0].w0);
      else if (q.n_src > 1 && f - k0 < q.s[1].k) v = apply_act(r1[f - k0], q.s[1].act, q.s[1].w0);
      else if (ones_col && f == q.ktot) v = 1.f;
      r.v[e] = v;
    }
  }
  return r;
}
// G[p][n0 .. n0+3] = gy * out_act'(y), zero past n_lim
__device__ __forceinline__ Vec4 g4(const LinParams& q, const float* gyrow, const float* yrow, int n0, int n_lim) {
  Vec4 r;
  if (n0 + 4 <= n_lim) {
    r = load4(gyrow + n0);
    if (q.out_act != SR_OUT_NONE) {
      const Vec4 y = load4(yrow + n0);
#pragma unroll
      for (int e = 0; e < 4; ++e) r.v[e] *= out_act_grad(y.v[e], q.out_act);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
      if (n0 + e < n_lim) v = q.out_act == SR_OUT_NONE ? gyrow[n0 + e] : gyrow[n0 + e] * out_act_grad(yrow[n0 + e], q.out_act);
      r.v[e] = v;
    }
  }
  return r;
}
__device__ __forceinline__ Vec4 w4(const float* p, int i0, int lim) {  // p[i0 .. i0+3], zero past lim
  Vec4 r;
  if (i0 + 4 <= lim) r = load4(p + i0);
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = i0 + e < lim ? p[i0 + e] : 0.f;
  }
  return r;
}

constexpr int kBM = 128, kBN = 128, kBK = 32, kRow = 40;  // LDS row = 32 bf16 + 8 of padding (80 B: 16-byte aligned fragments)
constexpr int kPlane = kBM * kRow;                          // halfwords per plane

__device__ __forceinline__ void split4(const Vec4& x, uint2& hi, uint2& lo) {
  uint32_t h0, l0, h1, l1;
  split_bf16x2(x.v[0], x.v[1], h0, l0);
  split_bf16x2(x.v[2], x.v[3], h1, l1);
  hi = make_uint2(h0, h1), lo = make_uint2(l0, l1);
}
// 4 consecutive k of one tile row -> one 8-byte store per plane
__device__ __forceinline__ void put_k4(uint16_t* plane_hi, int row, int k, const Vec4& x) {
  uint2 hi, lo;
  split4(x, hi, lo);
  *reinterpret_cast<uint2*>(plane_hi + row * kRow + k) = hi;
  *reinterpret_cast<uint2*>(plane_hi + kPlane + row * kRow + k) = lo;
}
// Row-major operands (memory runs along the tile's ROW index: DX B, DW A and B) are staged as "fragments", the layout the
// weight-gradient kernel uses (wgrad.hip): fragment f = rows 16f..16f+15 of the tile x 32 k, made of 16-byte units
// (hslot h, k) = rows 16f + 8h .. +7 at one k; [hslot 0: k = 0..31][hslot 1 rotated by 8 k][64 B gap].  A thread stores 8
// consecutive rows of one k with ONE 16-byte store per plane, and ds_read_b64_tr_b16 hands every lane the 8 consecutive k of
// its row -- the same operand the k-major image gives through ds_read_b128.
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kFragBytes = 1088, kFragPlane = 8 * kFragBytes;  // 8 fragments = 128 rows; bytes per plane (<= 2 * kPlane)
__device__ __forceinline__ void put_r8(char* plane_hi, int row8, int k, const Vec4& x0, const Vec4& x1) {
  uint2 h0, l0, h1, l1;
  split4(x0, h0, l0), split4(x1, h1, l1);
  const int f = row8 >> 1, pos = (row8 & 1) ? 32 + ((k + 8) & 31) : k;
  char* p = plane_hi + f * kFragBytes + pos * 16;
  *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(p + kFragPlane) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// MFMA operand (8 bf16 per lane: row lane&31 of the 32-row pair `frag_pair`, k = 16 ks + 8 (lane>>5) .. +7) from a fragment plane
__device__ __forceinline__ uint4 operand_tr(const char* plane, int frag_pair, int off0, int off1) {
  const char* p = plane + frag_pair * 2 * kFragBytes;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + off1));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}

// Staging.  Operands whose memory runs along the contraction index (FWD A and B, DX A) are moved as 4 consecutive k of one row
// per thread (thread -> k = 4*(t&7), rows (t>>3) + 32*it: the row pointers are computed once per kernel); operands whose memory
// runs along the row index (DX B, DW A and B) as 8 consecutive rows at one k (thread -> rows 8*(t&15), k = (t>>4) + 16*it).
// 4 waves per SIMD (<= 128 VGPRs, a handful of spilled values): the kernel is single-buffered and lives on occupancy to hide its
// global-load latency -- 512 x 512 forward 316 -> 244 us against 3 waves at 168 VGPRs; prefetching the next tile into registers
// instead (200 VGPRs, 2 waves) was slower, 365 us.
template <int KIND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) linear_kernel(const LinParams q) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * kPlane];  // A hi, A lo, B hi, B lo
  uint16_t* const lds_a = lds;
  uint16_t* const lds_b = lds + 2 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long i0 = (long)blockIdx.y * kBM, j0 = (long)blockIdx.x * kBN;
  const long k_begin = (long)blockIdx.z * q.kchunk;
  long k_end = k_begin + q.kchunk;
  if (k_end > q.K) k_end = q.K;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int kq = (tid & 7) * 4, rq = tid >> 3;        // k-major staging
  const int r8 = tid & 15, kr = tid >> 4;             // row-major staging: 8 rows 8*r8.. at k = kr + 16*it
  char* const frag_a = reinterpret_cast<char*>(lds_a);
  char* const frag_b = reinterpret_cast<char*>(lds_b);
  // transposed-read offsets of this lane (wgrad.hip): lane = (hh, rh, m, qq); k = 16 ks + 8 hh + 4 rd + m
  int tr_off[2][2];
  {
    const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int kk = 16 * ks + 8 * hh + 4 * rd + m;
        tr_off[ks][rd] = rh * kFragBytes + ((qq >> 1) ? 512 + ((kk + 8) & 31) * 16 : kk * 16) + (qq & 1) * 8;
      }
  }

  // row pointers of the k-major operands (fixed over the k loop)
  const float *pa0[4] = {}, *pa1[4] = {}, *pb[4] = {};
  bool va[4] = {}, vb[4] = {};
  if (KIND == kFwd || KIND == kDx) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long i = i0 + rq + 32 * it;
      va[it] = i < q.M;
      const long ic = va[it] ? i : 0;
      if (KIND == kFwd) {
        pa0[it] = q.s[0].x + row_of(ic, q.s[0].row_div) * q.s[0].ld;
        pa1[it] = q.n_src > 1 ? q.s[1].x + row_of(ic, q.s[1].row_div) * q.s[1].ld : nullptr;
      } else {
        pa0[it] = q.gy + ic * q.ldg;
        pa1[it] = q.out_act != SR_OUT_NONE ? q.y + ic * q.ldy : nullptr;
      }
    }
  }
  if (KIND == kFwd) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long j = j0 + rq + 32 * it;
      vb[it] = j < q.N;
      pb[it] = q.w + (vb[it] ? j : 0) * q.ldw;
    }
  }
  const Vec4 zero4 = {{0.f, 0.f, 0.f, 0.f}};

  f32x16 acc[2][2] = {};
  for (long kt = k_begin; kt < k_end; kt += kBK) {
    const int klim = (int)(k_end - kt < kBK ? k_end - kt : kBK);  // valid k in this tile
    // ---- A tile
    if (KIND == kFwd || KIND == kDx) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        Vec4 x = zero4;
        if (va[it] && kq < klim) {
          const int f0 = (int)kt + kq;
          x = KIND == kFwd ? acat4(q, pa0[it], pa1[it], f0, false) : g4(q, pa0[it], pa1[it], f0, (int)q.K);
          if (kq + 4 > klim) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kq + e >= klim) x.v[e] = 0.f;
          }
        }
        put_k4(lds_a, rq + 32 * it, kq, x);
      }
    } else {  // DW: Aop(i = n, k = p) = G[p][n], memory along n
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int k = kr + 16 * it;
        Vec4 x0 = zero4, x1 = zero4;
        if (k < klim && i0 + 8 * r8 < q.M) {
          const long p = kt + k;
          const float *gr = q.gy + p * q.ldg, *yr = q.out_act != SR_OUT_NONE ? q.y + p * q.ldy : nullptr;
          x0 = g4(q, gr, yr, (int)i0 + 8 * r8, (int)q.M), x1 = g4(q, gr, yr, (int)i0 + 8 * r8 + 4, (int)q.M);
        }
        put_r8(frag_a, r8, k, x0, x1);
      }
    }
    // ---- B tile
    if (KIND == kFwd) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        Vec4 x = zero4;
        if (vb[it] && kq < klim) x = w4(pb[it], (int)kt + kq, (int)k_end);
        put_k4(lds_b, rq + 32 * it, kq, x);
      }
    } else if (KIND == kDx) {  // Bop(j, k = n) = W[n][col0 + j], memory along j
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int k = kr + 16 * it;
        Vec4 x0 = zero4, x1 = zero4;
        if (k < klim && j0 + 8 * r8 < q.N) {
          const float* wr = q.w + (kt + k) * q.ldw + q.col0;
          x0 = w4(wr, (int)j0 + 8 * r8, (int)q.N), x1 = w4(wr, (int)j0 + 8 * r8 + 4, (int)q.N);
        }
        put_r8(frag_b, r8, k, x0, x1);
      }
    } else {  // DW: Bop(j = f, k = p) = acat[p][f], memory along f
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int k = kr + 16 * it;
        Vec4 x0 = zero4, x1 = zero4;
        if (k < klim && j0 + 8 * r8 < q.N) {
          const long p = kt + k;
          const float* s0 = q.s[0].x + row_of(p, q.s[0].row_div) * q.s[0].ld;
          const float* s1 = q.n_src > 1 ? q.s[1].x + row_of(p, q.s[1].row_div) * q.s[1].ld : nullptr;
          x0 = acat4(q, s0, s1, (int)j0 + 8 * r8, true), x1 = acat4(q, s0, s1, (int)j0 + 8 * r8 + 4, true);
        }
        put_r8(frag_b, r8, k, x0, x1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = (wm * 64 + t * 32 + l31) * kRow + ks * 16 + h * 8, rb = (wn * 64 + t * 32 + l31) * kRow + ks * 16 + h * 8;
        if (KIND == kDw) {
          ah[t] = operand_tr(frag_a, wm * 2 + t, tr_off[ks][0], tr_off[ks][1]);
          al[t] = operand_tr(frag_a + kFragPlane, wm * 2 + t, tr_off[ks][0], tr_off[ks][1]);
        } else {
          ah[t] = *reinterpret_cast<const uint4*>(lds_a + ra), al[t] = *reinterpret_cast<const uint4*>(lds_a + kPlane + ra);
        }
        if (KIND != kFwd) {
          bh[t] = operand_tr(frag_b, wn * 2 + t, tr_off[ks][0], tr_off[ks][1]);
          bl[t] = operand_tr(frag_b + kFragPlane, wn * 2 + t, tr_off[ks][0], tr_off[ks][1]);
        } else {
          bh[t] = *reinterpret_cast<const uint4*>(lds_b + rb), bl[t] = *reinterpret_cast<const uint4*>(lds_b + kPlane + rb);
        }
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

__device__ __forceinline__ float fetch_g(const LinParams& q, long p, int n) {
  const float g = q.gy[p * q.ldg + n];
  return q.out_act == SR_OUT_NONE ? g : g * out_act_grad(q.y[p * q.ldy + n], q.out_act);
}

// d_bias[n] += sum_p G[p][n]: block = 64 columns x a slab of rows, thread (column, row phase), LDS reduce, one atomic per column
constexpr int kColsumRows = 512;  // rows per block: enough blocks (and 4 loads in flight per thread) to run at memory speed
__global__ void __launch_bounds__(256) colsum_kernel(const LinParams q) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const long p0 = (long)blockIdx.y * kColsumRows;
  long p1 = p0 + kColsumRows;
  if (p1 > q.K) p1 = q.K;
  float s = 0.f;
  if (n < q.M) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long p = p0 + ph;
    for (; p + 12 < p1; p += 16) s += fetch_g(q, p, n), s1 += fetch_g(q, p + 4, n), s2 += fetch_g(q, p + 8, n), s3 += fetch_g(q, p + 12, n);
    for (; p < p1; p += 4) s += fetch_g(q, p, n);
    s += s1 + s2 + s3;
  }
  part[ph][c] = s;
  __syncthreads();
  if (ph == 0 && n < q.M) unsafeAtomicAdd(q.dbias + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
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
  SR_REQUIRE(dir_col >= 3 && ray_stride >= dir_col + 3 && n_samples >= 1, "sr_points_along: bad layout (stride %d, dir_col %d)", ray_stride, dir_col);
  const long n = (long)n_rays * n_samples;
  if (n <= 0) return 0;
  SR_REQUIRE(rays && z_vals && xyz, "sr_points_along: null pointer");
  hipLaunchKernelGGL(points_along_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, dir_col, z_vals, n,
                     n_samples, xyz);
  return check_launch("points_along_kernel");
}

extern "C" int sr_positional_map(const float* x, int ld, int dim, int64_t rows, int n_freqs, float* out, void* stream) {
  SR_REQUIRE(dim >= 1 && ld >= dim && n_freqs >= 1 && n_freqs <= 24, "sr_positional_map: bad sizes (dim %d, ld %d, %d frequencies)", dim, ld, n_freqs);
  const long n = (long)rows * 2 * n_freqs * dim;
  if (n <= 0) return 0;
  SR_REQUIRE(x && out, "sr_positional_map: null pointer");
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
  q.M = n_out, q.N = q.ktot, q.K = n_points;
  q.kchunk = 1024;  // points per workgroup: split-K with fp32 atomics
  const unsigned splits = (unsigned)((q.K + q.kchunk - 1) / q.kchunk);
  hipLaunchKernelGGL(linear_kernel<kDw>, dim3((unsigned)((q.N + kBN - 1) / kBN), (unsigned)((q.M + kBM - 1) / kBM), splits), dim3(256), 0,
                     (hipStream_t)stream, q);
  if (check_launch("linear_kernel<dw>")) return 1;
  if (d_bias) {
    const unsigned chunks = (unsigned)((n_points + kColsumRows - 1) / kColsumRows);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((n_out + 63) / 64), chunks), dim3(256), 0, (hipStream_t)stream, q);
    return check_launch("colsum_kernel");
  }
  return 0;
}
