// Weight gradients of the fused Sat-NeRF MLP for gfx950:  dW[row][col] = sum over sample points of dpre[row] * act[col].
//
// Replaces autograd's `grad_weight = grad_output^T @ input` / `grad_bias = sum(grad_output)` of every nn.Linear in
// SatNeRF (models/satnerf.py:104-153).  Both operands live in HBM as B fragments (point-major: one 16-byte unit holds 8
// consecutive slots of ONE point, see mlp_layout.h) because that is how the forward / dX kernels hold them in registers;
// this contraction runs over POINTS, so both MFMA operands need 8 consecutive points of one slot instead.  The transpose
// is done by the LDS: fragments are staged point-major and read back with ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-point x 16-slot block and each lane receives one slot's 4 points), two reads per 32x32x16 MFMA operand.
//
// Grid = (job blocks, split-K slices).  A workgroup (8 waves) owns one job block of up to 256 x 256 (16 row fragments of
// dpre x 16 column fragments of the saved activations, packing.backward_maps lists them) PLUS the block's aux columns
// (biases, skip / sun / embedding columns) over a contiguous slice of 32-point tiles, so every operand fragment is read
// from HBM once per job.  Phase-coded activation fragments are decoded to bf16 sin() on the way into LDS.  fp32 partial
// blocks go to `partial[slice][block][256*256 + 256*32]`; sr_unpack_grads sums the slices and scatters into the flat
// gradient.
#include "common.h"
#include "mlp_layout.h"

namespace sr {

struct WgradParams {
  const uint4* dpre;
  const uint4* acts;
  const int* blocks;  // 8 ints per block: row_frag0, n_row (<=16), col_frag0, n_col (0..16), col_kind, -, -, -
  float* partial;
  long n_tiles;
  long tiles_per_split;
  long split_stride;  // floats between slices
  int ak;             // activation fragments per tile
  int auxs;           // aux fragments (1 or 2) at the head of each activation tile
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS image of one fragment: [hslot 0: 32 points x 16 B][gap][hslot 1: 32 points x 16 B][pad]; the strides put the 32
// lanes of a transposed read on 64 distinct banks (point*16 + half*8 covers 64 B, hslot adds 128 B, fragment parity 64 B).
constexpr int kHslotStride = 640;
constexpr int kFragStride = 1344;
constexpr int kWgFrags = 34;  // 16 row + 16 column + 2 aux fragments
constexpr int kBufBytes = kWgFrags * kFragStride;
constexpr int kBlockFloats = 256 * 256 + 256 * 32;  // main block + aux columns

__device__ __forceinline__ uint32_t phase_pair_to_bf16(uint32_t w) {
  const float a = __builtin_amdgcn_sinf((float)(w & 0xffffu) * (1.0f / 65535.0f));
  const float b = __builtin_amdgcn_sinf((float)(w >> 16) * (1.0f / 65535.0f));
  return pack_bf16x2(a, b);
}

// Workgroup = 8 waves in a 4 x 2 grid; wave (wr, wc) owns rows 64*wr.. and columns 128*wc.. of the 256 x 256 block
// (2 x 4 MFMA tiles, 128 accumulator registers) plus the aux columns of row tile 2*wr + wc.  Ablation on MI355X
// (profiles/r01_ab_variants.txt): the kernel is bound by LDS traffic and the per-tile rendezvous, not by MFMA or HBM, hence few
// fat waves (1.4 transposed reads per MFMA instead of 2.2) and global loads issued two point tiles ahead.
struct Stage {
  uint4 r0, r1, c0, c1, ax;
};

__global__ void __launch_bounds__(512) wgrad_kernel(const WgradParams prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* d = prm.blocks + blockIdx.x * 8;
  const int rf0 = d[0], nr = d[1], cf0 = d[2], nc = d[3], kind = d[4];
  const long t_begin = (long)blockIdx.y * prm.tiles_per_split;
  long t_end = t_begin + prm.tiles_per_split;
  if (t_end > prm.n_tiles) t_end = prm.n_tiles;

  // staging: wave w moves row fragments w, w+8 and column fragments w, w+8; waves 0,1 also move the aux fragments
  const int fr0 = rf0 + (wave < nr ? wave : nr - 1), fr1 = rf0 + (wave + 8 < nr ? wave + 8 : nr - 1);
  const int fc0 = nc > 0 ? cf0 + (wave < nc ? wave : nc - 1) : 0, fc1 = nc > 0 ? cf0 + (wave + 8 < nc ? wave + 8 : nc - 1) : 0;
  const int fa = wave < prm.auxs ? wave : prm.auxs - 1;  // aux fragments are the first fragments of the activation tile
  const int unit_off = (lane >> 5) * kHslotStride + (lane & 31) * 16;
  auto fetch = [&](long tile, Stage& st) {
    const uint4* dp = prm.dpre + tile * kDpFrags * 64 + lane;
    const uint4* ac = prm.acts + tile * prm.ak * 64 + lane;
    st.r0 = ws_load_cached(dp + fr0 * 64), st.r1 = ws_load_cached(dp + fr1 * 64);
    st.c0 = ws_load_cached(ac + fc0 * 64), st.c1 = ws_load_cached(ac + fc1 * 64);
    if (wave < 2) st.ax = ws_load_cached(ac + fa * 64);
  };
  auto decode = [](const uint4& v) {
    return make_uint4(phase_pair_to_bf16(v.x), phase_pair_to_bf16(v.y), phase_pair_to_bf16(v.z), phase_pair_to_bf16(v.w));
  };
  auto stash = [&](int buf, Stage& st) {
    char* base = lds + buf * kBufBytes + unit_off;
#ifndef SR_ABL_NODECODE
    if (kind == 1) st.c0 = decode(st.c0), st.c1 = decode(st.c1);  // phase-coded sin stage -> bf16 activation values
#endif
    *reinterpret_cast<uint4*>(base + wave * kFragStride) = st.r0;
    *reinterpret_cast<uint4*>(base + (8 + wave) * kFragStride) = st.r1;
    *reinterpret_cast<uint4*>(base + (16 + wave) * kFragStride) = st.c0;
    *reinterpret_cast<uint4*>(base + (24 + wave) * kFragStride) = st.c1;
    if (wave < 2) *reinterpret_cast<uint4*>(base + (32 + wave) * kFragStride) = st.ax;
  };

  // transposed operand reads: lane = (hh, rh, m, q): MFMA row/col = 16*rh + 4*q + e, k = 8*hh + 4*rd + m
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  const int rd_off = rh * kFragStride + (q >> 1) * kHslotStride + (8 * hh + m) * 16 + (q & 1) * 8;
  const int wr = wave >> 1, wc = wave & 1;
  auto operand = [&](const char* buf, int frag_pair, int ks) {
    const char* p = buf + frag_pair * 2 * kFragStride + ks * 256 + rd_off;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 64));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
  };
  auto mma = [](const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
  const bool main_on = nc > 0;

  f32x16 acc[2][4] = {}, acc_aux = {};
  auto compute = [&](const char* b) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 a0 = operand(b, 2 * wr, ks), a1 = operand(b, 2 * wr + 1, ks);
#ifndef SR_ABL_NOMFMA
      if (main_on) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const uint4 bc = operand(b, 8 + 4 * wc + ct, ks);
          acc[0][ct] = mma(a0, bc, acc[0][ct]);
          acc[1][ct] = mma(a1, bc, acc[1][ct]);
        }
      }
#endif
      const uint4 bx = operand(b, 16, ks);  // aux fragments 32, 33 = fragment pair 16
      if (wc) acc_aux = mma(a1, bx, acc_aux);  // wave-uniform
      else acc_aux = mma(a0, bx, acc_aux);
    }
  };

  Stage sa, sb;  // tiles t and t+1 in flight: the loop is unrolled by two so the sets keep static names
  if (t_begin < t_end) fetch(t_begin, sa);
  if (t_begin + 1 < t_end) fetch(t_begin + 1, sb);
  for (long tile = t_begin; tile < t_end; tile += 2) {
    stash(0, sa);
    __syncthreads();
#ifndef SR_ABL_NOLOAD
    if (tile + 2 < t_end) fetch(tile + 2, sa);
#endif
    compute(lds);
    if (tile + 1 < t_end) {
      stash(1, sb);
      __syncthreads();
#ifndef SR_ABL_NOLOAD
      if (tile + 3 < t_end) fetch(tile + 3, sb);
#endif
      compute(lds + kBufBytes);
    }
  }

  float* out = prm.partial + (long)blockIdx.y * prm.split_stride + (long)blockIdx.x * kBlockFloats;
  const int n_rows = 16 * nr, n_cols = 16 * nc;
  if (main_on) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = 64 * wr + 32 * rt + (g & 3) + 8 * (g >> 2) + 4 * hh;
          const int col = 128 * wc + 32 * ct + (lane & 31);
          if (row < n_rows && col < n_cols) out[row * 256 + col] = acc[rt][ct][g];
        }
  }
  float* oa = out + 256 * 256;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int row = 64 * wr + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
    if (row < n_rows) oa[row * 32 + (lane & 31)] = acc_aux[g];
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
  p.split_stride = (long)n_blocks * kBlockFloats;
  p.auxs = aux_steps(tau);
  p.ak = act_ksteps(p.auxs);
  const size_t lds = 2 * (size_t)kBufBytes;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3(n_blocks, n_split), dim3(512), lds, (hipStream_t)stream, p);
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
