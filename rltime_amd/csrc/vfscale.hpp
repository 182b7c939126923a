// vfscale.hpp — value-function rescaling h / h^-1 and the n-step target tail,
// shared by the target kernels (qmath.hip) and the acting-time priority
// initialisation (replay.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mirl {

// torch_trainer.py:46-52
__device__ __forceinline__ float vf_scale(float x, float eps) {
  float s = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
  return s * (sqrtf(fabsf(x) + 1.f) - 1.f) + eps * x;
}
// torch_trainer.py:54-78 (float64 inside, float32 out)
__device__ __forceinline__ float vf_unscale(float y, double eps) {
  double a = fabs((double)y);
  double x = a / eps - (1.0 / (2.0 * (eps * eps))) * sqrt(4.0 * eps * a + (2.0 * eps + 1.0) * (2.0 * eps + 1.0)) +
             (2.0 * eps + 1.0) / (2.0 * (eps * eps));
  double s = y > 0.f ? 1.0 : (y < 0.f ? -1.0 : 0.0);
  return (float)(x * s);
}
// torch_trainer.py:144-147: h(ret + gamma**n * h^-1(v) * mask)
__device__ __forceinline__ float finish_target(float v, float ret, float disc, float mask, double vf_eps) {
  if (vf_eps > 0.0) v = vf_unscale(v, vf_eps);
  float y = ret + disc * v * mask;
  if (vf_eps > 0.0) y = vf_scale(y, (float)vf_eps);
  return y;
}

}  // namespace mirl
