// RPC ray generation on the GPU: the (H*W, 11) ray block of one satellite image from its RPC00B camera.
//
// Replaces datasets/satellite.py:18-65 (get_rays: rpcm.RPCModel.localization of every pixel at max_alt and min_alt ->
// sat_utils.latlon_to_ecef_custom (sat_utils.py:59-74) -> origin / unit direction / near = 0 / far), :218-227 (normalize_rays) and
// :229-244 (per-image sun direction) -- numpy on the host in the reference, cached per image as a torch-saved (H*W, 8) fp32 tensor.
// One thread per pixel, everything up to the fp32 cast in fp64: the 20-term RPC00B cubics (term order of rpcm's apply_poly),
// Newton iteration on the normalised projection with a finite-difference Jacobian until the squared pixel error is < 1e-18
// (rpcm's tolerance), then the same fp32 arithmetic as the reference's in-place tensor ops for the normalisation.
#include <math.h>
#include <string.h>

#include "common.h"

namespace sr {

struct RpcModel {  // host-filled, passed by value (90 doubles)
  double row_num[20], row_den[20], col_num[20], col_den[20];
  double row_offset, col_offset, lat_offset, lon_offset, alt_offset, row_scale, col_scale, lat_scale, lon_scale, alt_scale;
};

__device__ __forceinline__ double rpc_poly(const double* c, double x, double y, double z) {  // x = lat, y = lon, z = alt (normalised)
#pragma clang fp contract(off)
  return c[0] + c[1] * y + c[2] * x + c[3] * z + c[4] * y * x + c[5] * y * z + c[6] * x * z + c[7] * y * y + c[8] * x * x + c[9] * z * z +
         c[10] * x * y * z + c[11] * y * y * y + c[12] * y * x * x + c[13] * y * z * z + c[14] * y * y * x + c[15] * x * x * x +
         c[16] * x * z * z + c[17] * y * y * z + c[18] * x * x * z + c[19] * z * z * z;
}

// image (normalised col, row) at normalised altitude z -> normalised (lat x, lon y)
__device__ void rpc_localize(const RpcModel& m, double nc, double nr, double z, double& x, double& y) {
#pragma clang fp contract(off)
  x = 0.0, y = 0.0;
  const double eps = 1e-6;
  for (int it = 0; it < 100; ++it) {
    const double c0 = rpc_poly(m.col_num, x, y, z) / rpc_poly(m.col_den, x, y, z);
    const double r0 = rpc_poly(m.row_num, x, y, z) / rpc_poly(m.row_den, x, y, z);
    const double ec = nc - c0, er = nr - r0;
    if (ec * ec + er * er < 1e-18) break;
    const double cx = rpc_poly(m.col_num, x + eps, y, z) / rpc_poly(m.col_den, x + eps, y, z);
    const double rx = rpc_poly(m.row_num, x + eps, y, z) / rpc_poly(m.row_den, x + eps, y, z);
    const double cy = rpc_poly(m.col_num, x, y + eps, z) / rpc_poly(m.col_den, x, y + eps, z);
    const double ry = rpc_poly(m.row_num, x, y + eps, z) / rpc_poly(m.row_den, x, y + eps, z);
    const double j11 = (cx - c0) / eps, j12 = (cy - c0) / eps, j21 = (rx - r0) / eps, j22 = (ry - r0) / eps;
    const double det = j11 * j22 - j12 * j21;
    x = x + (ec * j22 - er * j12) / det;
    y = y + (er * j11 - ec * j21) / det;
  }
}

__device__ __forceinline__ void geodetic_to_ecef(double lat, double lon, double alt, double& X, double& Y, double& Z) {
#pragma clang fp contract(off)
  const double rad_lat = lat * (3.141592653589793 / 180.0), rad_lon = lon * (3.141592653589793 / 180.0);
  const double a = 6378137.0, f = 1 / 298.257223563;
  const double e2 = 1 - (1 - f) * (1 - f);
  const double v = a / sqrt(1 - e2 * sin(rad_lat) * sin(rad_lat));
  X = (v + alt) * cos(rad_lat) * cos(rad_lon);
  Y = (v + alt) * cos(rad_lat) * sin(rad_lon);
  Z = (v * (1 - e2) + alt) * sin(rad_lat);
}

__global__ void __launch_bounds__(256) rpc_rays_kernel(const RpcModel m, int width, long n, double min_alt, double max_alt, float cx, float cy,
                                                      float cz, float range, float sx, float sy, float sz, float* __restrict__ rays11,
                                                      float* __restrict__ rays8) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double col = (double)(i % width), row = (double)(i / width);  // np.meshgrid(arange(w), arange(h)) flattened, :193-194
  const double nc = (col - m.col_offset) / m.col_scale, nr = (row - m.row_offset) / m.row_scale;
  double P[2][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // k = 0: max_alt (closest to the camera = origin), k = 1: min_alt
    const double alt = k == 0 ? max_alt : min_alt;
    double x, y;
    rpc_localize(m, nc, nr, (alt - m.alt_offset) / m.alt_scale, x, y);
    geodetic_to_ecef(x * m.lat_scale + m.lat_offset, y * m.lon_scale + m.lon_offset, alt, P[k][0], P[k][1], P[k][2]);
  }
  const double d0 = P[1][0] - P[0][0], d1 = P[1][1] - P[0][1], d2 = P[1][2] - P[0][2];
  const double far = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  const float r8[8] = {(float)P[0][0], (float)P[0][1], (float)P[0][2], (float)(d0 / far), (float)(d1 / far), (float)(d2 / far), 0.f, (float)far};
  if (rays8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) rays8[i * 8 + c] = r8[c];  // the reference's <cache_dir>/<img_id>.data content
  }
  if (rays11) {  // normalize_rays in the tensor's own fp32 (datasets/satellite.py:218-227) + sun direction (:199-211)
    float* o = rays11 + i * 11;
    o[0] = (r8[0] - cx) / range, o[1] = (r8[1] - cy) / range, o[2] = (r8[2] - cz) / range;
    o[3] = r8[3], o[4] = r8[4], o[5] = r8[5];
    o[6] = r8[6] / range, o[7] = r8[7] / range;
    o[8] = sx, o[9] = sy, o[10] = sz;
  }
}

}  // namespace sr

using namespace sr;

extern "C" int sr_rpc_rays(const double* rpc, int width, int height, double min_alt, double max_alt, const double* center, double range,
                           double sun_elevation_deg, double sun_azimuth_deg, float* rays11, float* rays8, void* stream) {
  SR_REQUIRE(rpc && center, "sr_rpc_rays: null pointer");
  SR_REQUIRE(rays11 || rays8, "sr_rpc_rays: no output requested");
  SR_REQUIRE(width >= 1 && height >= 1, "sr_rpc_rays: bad image size %d x %d", width, height);
  SR_REQUIRE(range > 0, "sr_rpc_rays: scene range must be positive");
  RpcModel m;
  static_assert(sizeof(RpcModel) == 90 * sizeof(double), "RpcModel layout = the 90 host doubles");
  memcpy(&m, rpc, sizeof(m));
  SR_REQUIRE(m.row_scale != 0 && m.col_scale != 0 && m.lat_scale != 0 && m.lon_scale != 0 && m.alt_scale != 0, "sr_rpc_rays: zero RPC scale");
  const double el = sun_elevation_deg * (3.141592653589793 / 180.0), az = sun_azimuth_deg * (3.141592653589793 / 180.0);
  const long n = (long)width * height;
  hipLaunchKernelGGL(rpc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, width, n, min_alt, max_alt,
                     (float)center[0], (float)center[1], (float)center[2], (float)range, (float)(sin(az) * cos(el)), (float)(cos(az) * cos(el)),
                     (float)sin(el), rays11, rays8);
  return check_launch("rpc_rays_kernel");
}
