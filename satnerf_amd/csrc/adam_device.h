// One element of torch.optim.Adam (main.py:84: lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight decay), shared by the stand-alone Adam
// launches (train_ops.hip) and the fused gradient-tail + Adam launch (ray_ops.hip): the arithmetic per element is torch's.
#pragma once
#include "common.h"

namespace sr {

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float step_size, float b1, float b2, float eps, float grad_scale,
                                         float sqrt_bc2, int zero_grad) {
  const float gi = g * grad_scale;
  const float mi = b1 * m + (1.0f - b1) * gi;
  const float vi = b2 * v + (1.0f - b2) * gi * gi;
  m = mi, v = vi;
  p -= step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));  // torch: denom = sqrt(v) / sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr / bc1)
  if (zero_grad) g = 0.f;
}

}  // namespace sr
