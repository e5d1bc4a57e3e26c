// Weight gradients of the fused Sat-NeRF MLP for gfx950:  dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Replaces autograd's `grad_weight = grad_output^T @ input` / `grad_bias = sum(grad_output)` of every nn.Linear in
// SatNeRF (models/satnerf.py:104-153).  Both operands live in HBM as B fragments (point-major: one 16-byte unit holds 8
// consecutive slots of ONE point, see mlp_layout.h) because that is how the forward / dX kernels hold them in registers;
// this contraction runs over POINTS, so both MFMA operands need 8 consecutive points of one slot instead.  The transpose
// is done by the LDS: fragments are staged point-major and read back with ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-point x 16-slot block and each lane receives one slot's 4 points), two reads per 32x32x16 MFMA operand.
//
// Grid = (job blocks, split-K slices).  A workgroup (4 waves) owns one 128 x 128 block of one job (8 row fragments of
// dpre x 8 column fragments of the saved activations, packing.backward_maps lists them) over a contiguous slice of
// 32-point tiles; phase-coded activation fragments are decoded to bf16 sin() on the way into LDS.  fp32 partial blocks
// go to `partial[slice][block][128][128]`; sr_unpack_grads sums the slices and scatters into the flat gradient.
#include "common.h"
#include "mlp_layout.h"

namespace sr {

struct WgradParams {
  const uint4* dpre;
  const uint4* acts;
  const int* blocks;  // 8 ints per block: row_frag0, n_row, col_frag0, n_col, col_kind, -, -, -
  float* partial;
  long n_tiles;
  long tiles_per_split;
  long split_stride;  // floats between slices
  int ak;             // activation fragments per tile
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS image of one fragment: [hslot 0: 32 points x 16 B][gap][hslot 1: 32 points x 16 B][pad]; the strides put the 32
// lanes of a transposed read on 64 distinct banks (point*16 + half*8 covers 64 B, hslot adds 128 B, fragment parity 64 B).
constexpr int kHslotStride = 640;
constexpr int kFragStride = 1344;
constexpr int kBufBytes = 16 * kFragStride;

__device__ __forceinline__ uint32_t phase_pair_to_bf16(uint32_t w) {
  const float a = __builtin_amdgcn_sinf((float)(w & 0xffffu) * (1.0f / 65535.0f));
  const float b = __builtin_amdgcn_sinf((float)(w >> 16) * (1.0f / 65535.0f));
  return pack_bf16x2(a, b);
}

__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams prm) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kBufBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* d = prm.blocks + blockIdx.x * 8;
  const int rf0 = d[0], nr = d[1], cf0 = d[2], nc = d[3], kind = d[4];
  const long t_begin = (long)blockIdx.y * prm.tiles_per_split;
  long t_end = t_begin + prm.tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;

  // staging: this thread moves 4 units per tile: fragments wave, wave+4 (rows) and wave+8, wave+12 (columns)
  const int fr0 = rf0 + (wave < nr ? wave : nr - 1), fr1 = rf0 + (wave + 4 < nr ? wave + 4 : nr - 1);
  const int fc0 = cf0 + (wave < nc ? wave : nc - 1), fc1 = cf0 + (wave + 4 < nc ? wave + 4 : nc - 1);
  const int unit_off = (lane >> 5) * kHslotStride + (lane & 31) * 16;
  uint4 st[4];
  auto fetch = [&](long tile) {
    const uint4* dp = prm.dpre + tile * kDpFrags * 64 + lane;
    const uint4* ac = prm.acts + tile * prm.ak * 64 + lane;
    st[0] = dp[fr0 * 64], st[1] = dp[fr1 * 64], st[2] = ac[fc0 * 64], st[3] = ac[fc1 * 64];
  };
  auto stash = [&](int buf) {
    char* base = lds + buf * kBufBytes + unit_off;
    if (kind == 1) {  // phase-coded sin stage -> bf16 activation values
#pragma unroll
      for (int k = 2; k < 4; ++k)
        st[k] = make_uint4(phase_pair_to_bf16(st[k].x), phase_pair_to_bf16(st[k].y), phase_pair_to_bf16(st[k].z), phase_pair_to_bf16(st[k].w));
    }
    *reinterpret_cast<uint4*>(base + (wave)*kFragStride) = st[0];
    *reinterpret_cast<uint4*>(base + (wave + 4) * kFragStride) = st[1];
    *reinterpret_cast<uint4*>(base + (wave + 8) * kFragStride) = st[2];
    *reinterpret_cast<uint4*>(base + (wave + 12) * kFragStride) = st[3];
  };

  // transposed operand reads: lane = (hh, rh, m, q): MFMA row/col = 16*rh + 4*q + e, k = 8*hh + 4*rd + m
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  const int rd_off = rh * kFragStride + (q >> 1) * kHslotStride + (8 * hh + m) * 16 + (q & 1) * 8;
  const int wr = wave >> 1, wc = wave & 1;  // this wave's 64 x 64 quadrant
  auto operand = [&](const char* buf, int frag_pair, int ks) {
    const char* p = buf + frag_pair * 2 * kFragStride + ks * 256 + rd_off;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 64));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
  };

  f32x16 acc[2][2] = {};
  if (t_begin < t_end) fetch(t_begin);
  int buf = 0;
  for (long tile = t_begin; tile < t_end; ++tile) {
    stash(buf);
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);  // next tile's global loads fly during this tile's MFMAs
    const char* b = lds + buf * kBufBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 a_op[2], b_op[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a_op[i] = operand(b, 2 * wr + i, ks), b_op[i] = operand(b, 4 + 2 * wc + i, ks);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_op[rt]), __builtin_bit_cast(bf16x8, b_op[ct]),
                                                                 acc[rt][ct], 0, 0, 0);
    }
    buf ^= 1;
  }
  float* out = prm.partial + (long)blockIdx.y * prm.split_stride + (long)blockIdx.x * (128 * 128);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = 64 * wr + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh;
        const int col = 64 * wc + 32 * ct + (lane & 31);
        out[row * 128 + col] = acc[rt][ct][g];
      }
}

// grad[e] (+)= gscale[e] * sum_s partial[s * split_stride + gidx[e]]
__global__ void __launch_bounds__(256) unpack_grads_kernel(const float* __restrict__ partial, const int* __restrict__ gidx,
                                                          const float* __restrict__ gscale, long n, int n_split, long split_stride,
                                                          float* __restrict__ grad, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = gidx[i];
  if (k < 0) return;  // not produced by the fused MLP (sky head): left to its own kernel
  float s = 0.f;
  for (int sp = 0; sp < n_split; ++sp) s += partial[sp * split_stride + k];
  s *= gscale[i];
  grad[i] = accumulate ? grad[i] + s : s;
}

}  // namespace sr

using namespace sr;

extern "C" int sr_satnerf_wgrad(int feat, int tau, int64_t n_points, const uint16_t* dpre, const uint16_t* acts, const int32_t* blocks,
                                int n_blocks, int n_split, float* partial, void* stream) {
  SR_REQUIRE(feat == kFeat, "sr_satnerf_wgrad: feat=%d unsupported", feat);
  SR_REQUIRE(dpre && acts && blocks && partial, "sr_satnerf_wgrad: null pointer argument");
  SR_REQUIRE(n_blocks >= 1 && n_split >= 1 && n_split <= 65535, "sr_satnerf_wgrad: bad grid (%d blocks, %d slices)", n_blocks, n_split);
  WgradParams p;
  p.dpre = (const uint4*)dpre, p.acts = (const uint4*)acts, p.blocks = blocks, p.partial = partial;
  p.n_tiles = (n_points + 31) / 32;
  p.tiles_per_split = (p.n_tiles + n_split - 1) / n_split;
  p.split_stride = (long)n_blocks * 128 * 128;
  p.ak = act_ksteps(aux_steps(tau));
  hipLaunchKernelGGL(wgrad_kernel, dim3(n_blocks, n_split), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("wgrad_kernel");
}

extern "C" int sr_unpack_grads(const float* partial, const int32_t* gidx, const float* gscale, int64_t n_params, int n_split,
                               int64_t split_stride, float* grad, int accumulate, void* stream) {
  SR_REQUIRE(partial && gidx && gscale && grad, "sr_unpack_grads: null pointer");
  if (n_params <= 0) return 0;
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, gidx, gscale,
                     (long)n_params, n_split, (long)split_stride, grad, accumulate);
  return check_launch("unpack_grads_kernel");
}
