// Kernel parameter blocks shared by the API translation unit and the kernel instantiations (independent of the trunk width:
// they stay outside the per-width inline namespace of mlp_layout.h).
#pragma once
#include "common.h"

namespace sr {

struct FwdParams {
  sr_mlp_inputs in;
  const char* stream_hi;
  const char* stream_lo;
  const float4* l0;
  float* albedo;
  float* sigma;
  float* sun_v;
  float* beta;
  uint4* acts;
  int tau;
};

struct BwdParams {
  const float* g_albedo;  // upstream gradients of the per-point outputs (any may be null = 0)
  const float* g_sigma;
  const float* g_sun;
  const float* g_beta;
  const float* albedo;    // forward outputs (activation derivatives of the heads)
  const float* sigma;
  const float* sun_v;
  const float* beta;
  const uint4* acts;      // saved activations, act_ksteps(auxs) fragments per 32-point tile
  uint4* dpre;            // out: pre-activation gradients, kDpFrags fragments per tile
  float* d_t;             // out: (P, tau) gradient of the embedding vector per point (may be null)
  const char* stream;     // transposed weight stream (bf16)
  long n_points;
  int tau;
  int auxs;
};

}  // namespace sr
