// Probe (r06): does the LAYOUT of the training workspaces limit the bandwidth their traffic gets?  The workspaces are tile-major (a 32-point tile's U 1-KiB units
// contiguous); the forward / dX kernels run one wave per tile, all 2,048 tiles at once, each wave writing (reading) its tile's units one after the other -- isolated
// 1-KiB accesses 97 KiB apart -- and the weight-gradient kernel's workgroup streams ~103 consecutive tiles for a fixed set of 17 units.  The alternative, unit-major
// planes ([unit][tile]), makes the first pattern 2-MB sequential bursts and the second 17 sequential streams per workgroup.  Pure traffic, no arithmetic: what each
// address map sustains at the same bytes in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_layout.hip -o build_variants/probe_layout
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int T = 2048, U = 96;          // tiles, units per tile (1 KiB each): 192 MiB
__device__ __forceinline__ size_t addr(bool unit_major, int t, int u, int lane) {
  return (unit_major ? (size_t)u * T + t : (size_t)t * U + u) * 64 + lane;     // in uint4
}
// one wave per tile, units in order (the forward's stores / the dX kernel's loads); `depth` accesses in flight per wave
template <bool WRITE>
__global__ void __launch_bounds__(512) per_tile(uint4* ws, int unit_major, uint4* sink) {
  const int lane = threadIdx.x & 63, t = blockIdx.x * 8 + (threadIdx.x >> 6);
  uint4 acc = make_uint4(lane, t, 0, 0);
  for (int u = 0; u < U; ++u) {
    uint4* p = ws + addr(unit_major, t, u, lane);
    if (WRITE) __builtin_nontemporal_store(acc.x + u, &p->x), __builtin_nontemporal_store(acc.y, &p->y), __builtin_nontemporal_store(acc.z, &p->z), __builtin_nontemporal_store(acc.w, &p->w);
    else { const uint4 v = *p; acc.x ^= v.x, acc.y ^= v.y, acc.z += v.z, acc.w += v.w; }
  }
  if (!WRITE && acc.x == 0x12345678u) sink[0] = acc;
}
// the weight-gradient pattern: 256 workgroups x 4 waves; workgroup i = block i % 14 (a fixed set of 17 units), slice i / 14 of the tiles; every wave loads 4 or 5 units per tile
__global__ void __launch_bounds__(256) per_slice(const uint4* ws, int unit_major, uint4* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x % 14, sl = blockIdx.x / 14, n_sl = (gridDim.x + 13) / 14;
  const int per = (T + n_sl - 1) / n_sl, t0 = sl * per, t1 = min(T, t0 + per);
  const int u0 = (blk * 6) % (U - 17);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = u0 + (k < 2 ? 4 * k + wave : 8 + 4 * (k - 2) + wave);   // two "row" units, two "column" units per wave: 16 units per workgroup and tile
      const uint4 v = ws[addr(unit_major, t, u, lane)];
      acc.x ^= v.x, acc.y ^= v.y, acc.z += v.z, acc.w += v.w;
    }
    if (wave == 1) { const uint4 v = ws[addr(unit_major, t, u0 + 16, lane)]; acc.x ^= v.x; }
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  uint4 *ws, *sink;
  const size_t bytes = (size_t)T * U * 1024;
  (void)hipMalloc(&ws, bytes); (void)hipMalloc(&sink, 64); (void)hipMemset(ws, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](auto launch) { for (int i = 0; i < 3; ++i) launch(); (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                                 float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3; };
  for (int um = 0; um < 2; ++um) {
    const float w = time([&] { hipLaunchKernelGGL(per_tile<true>, dim3(T / 8), dim3(512), 0, 0, ws, um, sink); });
    const float r = time([&] { hipLaunchKernelGGL(per_tile<false>, dim3(T / 8), dim3(512), 0, 0, ws, um, sink); });
    const float g = time([&] { hipLaunchKernelGGL(per_slice, dim3(256), dim3(256), 0, 0, ws, um, sink); });
    double gb = 0;   // bytes the per_slice launch reads: workgroup i covers slice i / 14 of block i % 14 (the last level of slices is partly empty)
    for (int i = 0; i < 256; ++i) { const int sl = i / 14, per = (T + 18) / 19, t0 = sl * per, t1 = t0 + per < T ? t0 + per : T; if (t1 > t0) gb += (t1 - t0) * 17.0 * 1024 / 1e9; }
    printf("%-10s one wave per tile: write %6.1f us = %5.2f TB/s, read %6.1f us = %5.2f TB/s;  weight-gradient pattern (17 units x %d-tile slices): %6.1f us = %5.2f TB/s\n",
           um ? "unit-major" : "tile-major", w, bytes / w / 1e6, r, bytes / r / 1e6, (T + 18) / 19, g, gb * 1e3 / g);
  }
  return 0;
}
