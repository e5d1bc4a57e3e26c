// Probe: can the split-K partial blocks of the weight-gradient kernel be reduced by fp32 atomics in L2 instead of a 74-MB round trip
// through HBM?  252 workgroups x 1024 threads each add a 295-KB block (73,728 floats) into one of 15 accumulators (1.1 MB each, 4.4 MB in
// all: L2-resident), all at once -- the end-of-kernel burst of sr_satnerf_wgrad8.  Compared with writing the blocks out plainly.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probe_atomic_reduce.hip -o build_variants/probe_atomic_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kBlockFloats = 256 * 256 + 256 * 32;
__global__ void __launch_bounds__(1024) reduce_atomic(float* acc, const int* block_of, float v) {
  float* dst = acc + (long)block_of[blockIdx.x] * kBlockFloats;
  for (int i = threadIdx.x; i < kBlockFloats; i += 1024) unsafeAtomicAdd(dst + i, v + i);
}
// the same with the slices of a block all on ONE XCD (workgroup i runs on XCD i % 8: blocks 2x, 2x + 1 belong to XCD x) and atomics of
// workgroup scope, which the XCD's own L2 performs
__global__ void __launch_bounds__(1024) reduce_atomic_xcd(float* acc, float v) {
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  float* dst = acc + (long)(2 * xcd + (k & 1)) * kBlockFloats;
  for (int i = threadIdx.x; i < kBlockFloats; i += 1024) __hip_atomic_fetch_add(dst + i, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void __launch_bounds__(1024) write_plain(float* out, float v) {
  float* dst = out + (long)blockIdx.x * kBlockFloats;
  for (int i = threadIdx.x; i < kBlockFloats; i += 1024) __builtin_nontemporal_store(v + i, dst + i);
}
int main() {
  const int n_wg = 256, n_blocks = 16;
  float *acc, *out; int* bo; int h[n_wg];
  for (int i = 0; i < n_wg; ++i) h[i] = i * n_blocks / n_wg;
  (void)hipMalloc(&acc, (size_t)n_blocks * kBlockFloats * 4); (void)hipMalloc(&out, (size_t)n_wg * kBlockFloats * 4); (void)hipMalloc(&bo, sizeof(h));
  (void)hipMemcpy(bo, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMemset(acc, 0, (size_t)n_blocks * kBlockFloats * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) {
        if (variant == 0) hipLaunchKernelGGL(reduce_atomic, dim3(n_wg), dim3(1024), 0, 0, acc, bo, 1.0f);
        else if (variant == 2) hipLaunchKernelGGL(reduce_atomic_xcd, dim3(n_wg), dim3(1024), 0, 0, acc, 1.0f);
        else hipLaunchKernelGGL(write_plain, dim3(n_wg), dim3(1024), 0, 0, out, 1.0f);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.1f us per launch (%d workgroups x %d KB)\n", variant == 0 ? "fp32 agent-scope atomics into 16 blocks" : variant == 2 ? "workgroup-scope atomics, a block's slices on one XCD" : "plain non-temporal stores of 256 blocks", ms / 20 * 1e3, n_wg, kBlockFloats * 4 / 1024);
    }
  }
  return 0;
}
