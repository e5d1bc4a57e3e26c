// Kernel parameter blocks shared by the API translation unit and the kernel instantiations.
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

}  // namespace sr
